#!/usr/bin/env python3
"""Does the step time depend on WHERE the buffers sit in HBM?  (development probe)  One C2 image; the result arena, the descriptor
array and the chunk table are moved inside over-allocated buffers, one at a time; median ms of 5 launches per placement."""
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vcf2prot_amd import _native as N  # noqa: E402
from vcf2prot_amd.cohort import Cohort  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
    samples = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    lib = N.bench_lib()        # (v2p_stitch_launch with the packed flag word: libv2p_bench.so)
    dev = torch.device("cuda", 0)
    cohort = Cohort.preset(wl, n_samples=samples)
    prot = cohort.proteome()
    resident = np.concatenate([prot, cohort.fasta_headers()])
    pack_kw = dict(x.split("=") for x in sys.argv[4:])
    img = cohort.pack(0, cohort.n_haplotypes, n_threads=min(64, os.cpu_count() or 1), **{k: int(v) for k, v in pack_kw.items()})
    chunks = np.ascontiguousarray(img.chunks)
    lib.v2p_order_chunks_for_xcds(chunks.ctypes.data, chunks.shape[0], img.desc.ctypes.data, img.desc.size, prot.size)
    bits = int(lib.v2p_stitch_launch_bits(chunks.ctypes.data, chunks.shape[0])) | 1
    stream = torch.cuda.current_stream()
    SLACK = 1 << 30
    d_prot = torch.zeros(resident.size + 128, dtype=torch.uint8, device=dev)
    d_prot[64:64 + resident.size] = torch.from_numpy(resident).to(dev)
    d_pay = torch.zeros(img.payload.size + 128, dtype=torch.uint8, device=dev)
    d_pay[64:64 + img.payload.size] = torch.from_numpy(img.payload).to(dev)
    d_status = torch.full((1,), -1, dtype=torch.int64, device=dev)
    big_out = torch.empty(img.out_bytes + SLACK + 4096, dtype=torch.uint8, device=dev)
    big_desc = torch.zeros(img.desc.size * 8 + SLACK + 4096, dtype=torch.uint8, device=dev)
    big_chunks = torch.zeros(chunks.size * 8 + SLACK + 4096, dtype=torch.uint8, device=dev)
    h_desc = torch.from_numpy(img.desc.view(np.uint8))
    h_chunks = torch.from_numpy(chunks.view(np.uint8).reshape(-1))
    print("bases: out %x desc %x chunks %x prot %x" % (big_out.data_ptr(), big_desc.data_ptr(), big_chunks.data_ptr(), d_prot.data_ptr()))

    def run(o_out, o_desc, o_chunks):
        big_desc[o_desc + 64:o_desc + 64 + h_desc.numel()] = h_desc.to(dev)
        big_chunks[o_chunks:o_chunks + h_chunks.numel()] = h_chunks.to(dev)
        ms = []
        for r in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            rc = lib.v2p_stitch_launch(ctypes.c_void_p(stream.cuda_stream), big_desc.data_ptr() + o_desc + 64, img.desc.size, big_chunks.data_ptr() + o_chunks, chunks.shape[0],
                                       d_prot.data_ptr() + 64, resident.size, d_pay.data_ptr() + 64, img.payload.size,
                                       big_out.data_ptr() + o_out, img.out_bytes, d_status.data_ptr(), bits, 0)
            assert rc == 0
            e1.record(stream)
            torch.cuda.synchronize()
            if r:
                ms.append(e0.elapsed_time(e1))
        return statistics.median(ms)

    offs = [0, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 16 << 20, 100 << 20, 128 << 20, 256 << 20, 512 << 20, 777 << 20, (1 << 30) - 4096]
    if len(sys.argv) > 3 and sys.argv[3] == "sub":        # sub-line / sub-row shifts of the arena: are 1 KiB-aligned rows better or worse?
        for o in [0, 16, 32, 48, 64, 128, 144, 512, 528, 1024, 0, 16, 64]:
            print(f"out    offset {o:>11d}: {run(o, 0, 0):.3f} ms", flush=True)
        return
    for name, idx in (("out", 0), ("desc", 1), ("chunks", 2)):
        for o in offs:
            args = [0, 0, 0]
            args[idx] = o
            print(f"{name:6s} offset {o:>11d}: {run(*args):.3f} ms", flush=True)
    for k in range(6):
        print(f"baseline again: {run(0, 0, 0):.3f} ms")


main()
