#!/usr/bin/env python3
"""Which buffer's placement decides a thin image's execute time?  One C2 image; A arenas x D descriptor arrays (+ chunk tables) allocated
separately (torch), every combination timed through the raw launcher (v2p_stitch_launch_opts)."""
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
    na = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    nd = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    lib = N.hip_lib()
    dev = torch.device("cuda", 0)
    cohort = Cohort.preset(wl, n_samples=samples)
    prot = cohort.proteome()
    img = cohort.pack(0, cohort.n_haplotypes, n_threads=min(64, os.cpu_count() or 1))
    chunks = np.ascontiguousarray(img.chunks)
    lib.v2p_order_chunks_for_xcds(chunks.ctypes.data, chunks.shape[0], img.desc.ctypes.data, img.desc.size, prot.size)
    bits = int(lib.v2p_stitch_launch_bits(chunks.ctypes.data, chunks.shape[0]))
    stream = torch.cuda.current_stream()
    d_prot = torch.zeros(prot.size + 128, dtype=torch.uint8, device=dev)
    d_prot[64:64 + prot.size] = torch.from_numpy(prot).to(dev)
    d_pay = torch.zeros(img.payload.size + 128, dtype=torch.uint8, device=dev)
    d_pay[64:64 + img.payload.size] = torch.from_numpy(img.payload).to(dev)
    d_status = torch.full((1,), -1, dtype=torch.int64, device=dev)
    outs = [torch.empty(img.out_bytes + 4096, dtype=torch.uint8, device=dev) for _ in range(na)]
    descs, chs = [], []
    for _ in range(nd):
        t = torch.zeros(img.desc.size * 8 + 256, dtype=torch.uint8, device=dev)
        t[64:64 + img.desc.size * 8] = torch.from_numpy(img.desc.view(np.uint8)).to(dev)
        descs.append(t)
        u = torch.from_numpy(chunks.view(np.uint8).reshape(-1)).to(dev)
        chs.append(u)
    opts = N.LaunchOpts(nontemporal=1, routing=bits)

    def run(o, d):
        ms = []
        for r in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            rc = N.stitch_launch(lib, stream.cuda_stream, descs[d].data_ptr() + 64, img.desc.size, chs[d].data_ptr(), chunks.shape[0], d_prot.data_ptr() + 64, prot.size,
                                 d_pay.data_ptr() + 64, img.payload.size, outs[o].data_ptr(), img.out_bytes, d_status.data_ptr(), opts)
            assert rc == 0
            e1.record(stream)
            torch.cuda.synchronize()
            if r:
                ms.append(e0.elapsed_time(e1))
        return statistics.median(ms)
    print("arenas:", [hex(x.data_ptr()) for x in outs])
    print("desc:  ", [hex(x.data_ptr()) for x in descs])
    for o in range(na):
        print("arena", o, " ".join(f"{run(o, d):.3f}" for d in range(nd)), flush=True)


main()
