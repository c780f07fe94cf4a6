#!/usr/bin/env python3
"""Ceilings for the stitch kernel's K2 phase (v2p_copy_bench_launch): a cache-resident source window streamed into HBM with
non-temporal 16-byte stores, per load flavour and source misalignment.  Prints TB/s of bytes written.

    python tools/copy_bench.py [--gb 16] [--window-mb 8] [--rounds 7]
"""
import argparse
import ctypes
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vcf2prot_amd import _native as N  # noqa: E402

MODES = {0: "byte-granular gather", 1: "aligned + lane exchange", 2: "two aligned loads", 3: "dword-aligned x4+x1",
         4: "stores only", 5: "loads only (aligned + lane exchange)", 6: "two aligned, load->store per pass",
         7: "two aligned, next loads before store", 8: "gather + 8 B/lane descriptor stream",
         9: "gather + header -> descriptor stream"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=16.0)
    ap.add_argument("--window-mb", type=float, default=8.0)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    lib = N.bench_lib()
    dev = torch.device("cuda", 0)
    nbytes = int(a.gb * (1 << 30)) // (32 << 10) * (32 << 10)
    window = int(a.window_mb * (1 << 20))
    src = torch.randint(0, 255, (window + (1 << 17),), dtype=torch.uint8, device=dev)
    out = torch.zeros(nbytes + 4096 + nbytes // 16 + nbytes // 1024 + (1 << 20), dtype=torch.uint8, device=dev)   # + descriptor and header streams (zeros)
    sink = torch.zeros(nbytes // (32 << 10) + 1, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream()
    res = []
    cases = [(4, 0, 0), (0, 0, 0), (0, 5, 0), (1, 5, 0), (2, 5, 0), (3, 5, 0), (5, 5, 0), (6, 5, 0), (7, 5, 0),
             (0, 5, 7000), (7, 5, 7000), (8, 5, 0), (9, 5, 0), (0, 5, 0)]
    for mode, shift, delay in cases:
        ms = []
        for r in range(a.rounds + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = lib.v2p_copy_bench_launch(ctypes.c_void_p(st.cuda_stream), src.data_ptr(), window, shift | (delay << 4), out.data_ptr(), nbytes, mode, sink.data_ptr())
            assert rc == 0
            e1.record(st)
            torch.cuda.synchronize()
            if r:
                ms.append(e0.elapsed_time(e1))
        med = statistics.median(ms)
        row = {"mode": mode, "what": MODES[mode], "shift": shift, "delay_cycles": delay, "ms_median": med, "ms_min": min(ms), "TBps": nbytes / med / 1e9}
        res.append(row)
        print(f"mode {mode} ({MODES[mode]:38s}) shift {shift:2d} delay {delay:5d}: median {med:7.3f} ms  min {min(ms):7.3f} ms  {row['TBps']:6.2f} TB/s")
    # persistent workgroups with the descriptor requested `depth` spans ahead
    for depth, db, grid in [(1, 0, 2048), (1, 8, 2048), (2, 8, 2048), (4, 8, 2048), (8, 8, 2048), (1, 4, 2048), (4, 4, 2048), (4, 8, 1024), (8, 8, 1024)]:
        ms = []
        for r in range(a.rounds + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = lib.v2p_copy_prefetch_launch(ctypes.c_void_p(st.cuda_stream), src.data_ptr(), window, 5, out.data_ptr(), nbytes, depth, db, grid)
            assert rc == 0
            e1.record(st)
            torch.cuda.synchronize()
            if r:
                ms.append(e0.elapsed_time(e1))
        med = statistics.median(ms)
        res.append({"mode": "persistent", "depth": depth, "desc_bytes_per_lane": db, "grid": grid, "ms_median": med, "TBps": nbytes / med / 1e9})
        print(f"persistent grid {grid} desc {db} B/lane requested {depth} span(s) ahead: median {med:7.3f} ms  {nbytes / med / 1e9:6.2f} TB/s")
    # mode 1 must reproduce mode 0 bit for bit
    for shift in (0, 3, 5, 12):
        outs = []
        for mode in (0, 1, 2, 3, 6, 7):
            lib.v2p_copy_bench_launch(ctypes.c_void_p(st.cuda_stream), src.data_ptr(), window, shift, out.data_ptr(), 1 << 26, mode, sink.data_ptr())
            torch.cuda.synchronize()
            outs.append(out[:1 << 26].clone())
        same = [bool(torch.equal(outs[0], o)) for o in outs[1:]]
        print(f"shift {shift}: modes 1,2,3,6,7 equal mode 0: {same}")
        assert all(same)
    if a.json:
        json.dump({"bytes": nbytes, "window": window, "cases": res}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
