#!/usr/bin/env python3
"""Ceilings of stitchw_kernel's data movement (vcf2prot_amd/csrc/bench/wave_copy_bench.hip): TB/s written per source pattern,
extra gathers, descriptor stream, workgroup shape.

    python tools/wave_copy_bench.py [--gb 8] [--rounds 5]
"""
import argparse
import ctypes
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=8.0)
    ap.add_argument("--window-mb", type=float, default=8.0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--json", default="")
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--chain-only", action="store_true")
    ap.add_argument("--phased-only", action="store_true")
    a = ap.parse_args()
    from vcf2prot_amd import build as B
    path = B.build_bench()
    if a.build_only:
        return
    import torch                                   # (first: the library then binds to the HIP runtime torch has loaded)
    lib = ctypes.CDLL(path)
    lib.v2p_bench_wave_copy.restype = ctypes.c_int
    lib.v2p_bench_wave_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                        ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int,
                                        ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    dev = torch.device("cuda", 0)
    nbytes = int(a.gb * (1 << 30)) // 8192 * 8192
    window = int(a.window_mb * (1 << 20))
    src = torch.randint(0, 255, (window + (1 << 17),), dtype=torch.uint8, device=dev)
    out = torch.zeros(nbytes + 4096, dtype=torch.uint8, device=dev)
    dsc = torch.zeros(nbytes // 8192 * 64 + 64, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream()
    rows = []
    # (wpg, pattern, label, shift, extra gathers, descriptor lanes, descriptor table wrap, store policy, prefetch distance)
    C2, C3, ONE = 0, 1, 2
    cases = [(1, C2, "contiguous (C2)", 5, 0, 0, 0, 2, 0), (1, C2, "C2 + 512 B desc", 5, 0, 64, 0, 2, 0),
             (1, C2, "C2 + 512 B desc, table of 65536 chunks (32 MB)", 5, 0, 64, 65536, 2, 0),
             (1, C2, "C2 + 512 B desc, table of 131072 chunks (64 MB)", 5, 0, 64, 131072, 2, 0),
             (1, C2, "C2 + 512 B desc, table of 262144 chunks (128 MB)", 5, 0, 64, 262144, 2, 0),
             (1, C2, "C2 + 512 B desc, table of 393216 chunks (192 MB)", 5, 0, 64, 393216, 2, 0),
             (1, C2, "C2 + 512 B desc, table of 524288 chunks (256 MB)", 5, 0, 64, 524288, 2, 0),
             (1, C3, "C3 + 448 B desc", 5, 0, 56, 0, 2, 0)]
    if a.chain_only or a.phased_only:
        cases = cases[:2]
    for wpg, pattern, label, shift, n_p, dl, dmod, aux, pf in cases:
        ms = []
        for r in range(a.rounds + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = lib.v2p_bench_wave_copy(ctypes.c_void_p(st.cuda_stream), src.data_ptr(), window, out.data_ptr(), nbytes, dsc.data_ptr() if dl else None,
                                         pattern, shift, n_p, 26, 0, wpg, dl, dmod, aux, pf)
            assert rc == 0, rc
            e1.record(st)
            torch.cuda.synchronize()
            if r:
                ms.append(e0.elapsed_time(e1))
        med = statistics.median(ms)
        rows.append({"wpg": wpg, "what": label, "extra_gathers": n_p, "descriptor_bytes": dl * 8, "descriptor_table_chunks": dmod, "store_aux": aux, "prefetch": pf, "ms": med, "TBps_written": nbytes / med / 1e9})
        print(f"{label:58s}: {med:7.3f} ms  {nbytes / med / 1e9:6.2f} TB/s")
    lib.v2p_bench_wave_chain.restype = ctypes.c_int
    lib.v2p_bench_wave_chain.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    n_chunks = nbytes // 8192
    rec = torch.zeros(2 * n_chunks + 16, dtype=torch.int64, device=dev)
    rec[0:2 * n_chunks:2] = torch.arange(n_chunks, dtype=torch.int64, device=dev)
    for pattern, dl, label in (() if a.phased_only else ((C2, 64, "chain C2 + 512 B desc"), (C3, 56, "chain C3 + 448 B desc"), (C2, 30, "chain C2 + 240 B desc"))):
        for remap, persist in ((0, 0), (1, 0), (0, 8192), (0, 4096), (0, 6144), (0, 16384)):
            ms = []
            for r in range(a.rounds + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                rc = lib.v2p_bench_wave_chain(ctypes.c_void_p(st.cuda_stream), src.data_ptr(), window, out.data_ptr(), nbytes, dsc.data_ptr(), rec.data_ptr(),
                                              pattern, dl, remap, persist)
                assert rc == 0, rc
                e1.record(st)
                torch.cuda.synchronize()
                if r:
                    ms.append(e0.elapsed_time(e1))
            med = statistics.median(ms)
            what = f"{label}" + (", XCD-contiguous chunk table" if remap else "") + (f", {persist} resident waves, loads one chunk ahead" if persist else "")
            rows.append({"what": what, "remap": remap, "persist_waves": persist, "descriptor_bytes": dl * 8, "ms": med, "TBps_written": nbytes / med / 1e9})
            print(f"{what:72s}: {med:7.3f} ms  {nbytes / med / 1e9:6.2f} TB/s")
    if a.chain_only:
        if a.json:
            json.dump({"bytes": nbytes, "window": window, "cases": rows}, open(a.json, "w"), indent=1)
        return
    lib.v2p_bench_wave_copy_phased.restype = ctypes.c_int
    lib.v2p_bench_wave_copy_phased.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                               ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int]
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    for pattern, dl, label in ((C2, 64, "C2 + 512 B desc"), (C3, 56, "C3 + 448 B desc")):
        for phases in (1, 2, 4, 8, 16, 32, 64):
            for touch in (0, 1, 2):
                if (phases == 1 and touch) or (a.phased_only and touch == 1):
                    continue
                ms = []
                for r in range(a.rounds + 1):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    rc = lib.v2p_bench_wave_copy_phased(ctypes.c_void_p(st.cuda_stream), src.data_ptr(), window, out.data_ptr(), nbytes, dsc.data_ptr(),
                                                        pattern, dl, 2, phases, sink.data_ptr(), touch)
                    assert rc == 0, rc
                    e1.record(st)
                    torch.cuda.synchronize()
                    if r:
                        ms.append(e0.elapsed_time(e1))
                med = statistics.median(ms)
                what = f"{label}, {phases} sub-launches" + (", descriptors touched before each" + (" (plain loads)" if touch == 2 else " (nt loads)") if touch else "")
                rows.append({"what": what, "phases": phases, "touch": touch, "ms": med, "TBps_written": nbytes / med / 1e9})
                print(f"{what:58s}: {med:7.3f} ms  {nbytes / med / 1e9:6.2f} TB/s")
    if a.json:
        json.dump({"bytes": nbytes, "window": window, "cases": rows}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
