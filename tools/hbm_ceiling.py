#!/usr/bin/env python3
"""Measured HBM ceilings on this GPU for the access shapes the stitch kernel uses:
16-byte streaming stores (fill, nt and plain) and a 16-byte copy (torch).  GB/s, median of reps."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vcf2prot_amd import _native as N  # noqa: E402


def timed(fn, reps=10):
    ts = []
    for _ in range(3):
        fn()
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    lib = N.bench_lib()
    gb = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
    n = int(gb * 1e9) // 16 * 16
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    src = torch.empty(n, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    res = {"bytes": n}
    for name, nt in (("fill_nt", 1), ("fill_plain", 0)):
        ms = timed(lambda: lib.v2p_fill_launch(ctypes.c_void_p(s), out.data_ptr(), n, 0x2E2E2E2E, nt))
        res[name + "_GBs"] = n / ms / 1e6
    for span in (256, 1024, 2048, 4096):          # 16-byte blocks per workgroup: 4, 16, 32, 64 KiB
        for nt in (1, 0):
            ms = timed(lambda: lib.v2p_fill_launch(ctypes.c_void_p(s), out.data_ptr(), n, 0x2E2E2E2E, nt | (span << 8)))
            res[f"fill_span{span * 16 // 1024}K_{'nt' if nt else 'plain'}_GBs"] = round(n / ms / 1e6, 1)
    ms = timed(lambda: out.copy_(src))
    res["torch_copy_GBs_read_plus_write"] = 2 * n / ms / 1e6
    ms = timed(lambda: out.fill_(46))
    res["torch_fill_GBs"] = n / ms / 1e6
    print(json.dumps(res))


if __name__ == "__main__":
    main()
