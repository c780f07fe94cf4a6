#!/usr/bin/env python3
"""L1/address-path ceiling for 16-byte-per-lane gathers from an L2-resident window, aligned vs misaligned."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vcf2prot_amd import _native as N
lib = N.bench_lib()
window = 8 << 20
src = torch.randint(0, 255, (window + 4096,), dtype=torch.uint8, device="cuda")
blocks, iters = 2048, 2048
sink = torch.zeros(blocks * 4, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
res = {}
for mis in (0, 16, 4, 1, 5, 8, 13):
    for _ in range(2):
        lib.v2p_gather_bench_launch(ctypes.c_void_p(s), src.data_ptr(), window, mis, iters, blocks, sink.data_ptr())
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    lib.v2p_gather_bench_launch(ctypes.c_void_p(s), src.data_ptr(), window, mis, iters, blocks, sink.data_ptr())
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    res[f"misalign_{mis}"] = round(blocks * 4 * iters * 1024 / ms / 1e6, 1)     # GB/s
print(json.dumps(res))
