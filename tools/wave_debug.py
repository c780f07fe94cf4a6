#!/usr/bin/env python3
"""Development tool: run a wave image through v2p_stitch_launch of a chosen library build (e.g. build_ab/wave_check.so, built with
-DV2P_WAVE_CHECK: every gather range-checked) and compare the arena with the per-block kernel's; prints the device status word.

    python tools/wave_debug.py --lib build_ab/wave_check.so C3 11 5
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vcf2prot_amd import _native as N  # noqa: E402
from vcf2prot_amd.cohort import Cohort  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("preset"); ap.add_argument("h0", type=int); ap.add_argument("n", type=int)
    ap.add_argument("--lib", default="")
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--wpg", type=int, default=0)
    a = ap.parse_args()
    if a.lib:
        N.HIP_LIB_PATH = os.path.abspath(a.lib)
    lib = N.bench_lib()        # (v2p_stitch_launch with the packed flag word: libv2p_bench.so)
    dev = torch.device("cuda", 0)
    c = Cohort.preset(a.preset)
    prot = c.proteome()
    d_prot = torch.zeros(prot.size + 128, dtype=torch.uint8, device=dev)
    d_prot[64:64 + prot.size] = torch.from_numpy(prot).to(dev)
    stream = torch.cuda.current_stream()
    outs = []
    for kernel in (2, 4):
        img = c.pack(a.h0, a.h0 + a.n, n_threads=a.threads, kernel=kernel)
        chunks = np.ascontiguousarray(img.chunks)
        d_pay = torch.zeros(img.payload.size + 128, dtype=torch.uint8, device=dev)
        d_pay[64:64 + img.payload.size] = torch.from_numpy(img.payload).to(dev)
        d_desc = torch.zeros(img.desc.size + 16, dtype=torch.int64, device=dev)       # (64 readable bytes either side)
        d_desc[8:8 + img.desc.size] = torch.from_numpy(img.desc.view(np.int64)).to(dev)
        d_chunks = torch.from_numpy(chunks.view(np.int64)).to(dev)
        d_out = torch.zeros(img.out_bytes + 32, dtype=torch.uint8, device=dev)
        d_status = torch.full((1,), -1, dtype=torch.int64, device=dev)
        flags = 1 | int(lib.v2p_stitch_launch_bits(chunks.ctypes.data, chunks.shape[0])) | (a.wpg << 28)
        rc = lib.v2p_stitch_launch(ctypes.c_void_p(stream.cuda_stream), d_desc.data_ptr() + 64, img.desc.size, d_chunks.data_ptr(), chunks.shape[0],
                                   d_prot.data_ptr() + 64, prot.size, d_pay.data_ptr() + 64, img.payload.size, d_out.data_ptr(), img.out_bytes,
                                   d_status.data_ptr(), flags, 0)
        torch.cuda.synchronize()
        st = int(d_status.item()) & 0xFFFFFFFFFFFFFFFF
        print(f"kernel={kernel} rc={rc} chunks={chunks.shape[0]} flags={flags:#x} status={st:#x}", end="")
        if st != 0xFFFFFFFFFFFFFFFF:
            reason, idx = st & 0xFF, st >> 8
            print(f"  -> reason {reason} index {idx:#x} (checked build: chunk {idx >> 8}, lane {idx & 0xFF}, site {reason - 100})", end="")
        print()
        outs.append(d_out[:img.out_bytes].cpu().numpy())
    bad = np.nonzero(outs[0] != outs[1])[0]
    print("arena", "EQUAL" if bad.size == 0 else f"DIFFERS at {bad[:12]} ({bad.size} bytes)")
    if bad.size:
        k = int(bad[0]) & ~15
        print("want", bytes(outs[0][k:k + 32]))
        print("got ", bytes(outs[1][k:k + 32]))
        # which chunk / row / lane every differing block belongs to
        dst = (chunks[:, 1] & np.uint64((1 << 48) - 1)).astype(np.int64)
        order = np.argsort(dst, kind="stable")
        sd = dst[order]
        blocks = np.unique(bad >> 4)
        seen = {}
        for blk in blocks:
            pos = int(blk) << 4
            ci = int(np.searchsorted(sd, pos, side="right")) - 1
            c0 = int(sd[ci]); head = c0 & 15
            end = int(sd[ci + 1]) if ci + 1 < sd.size else img.out_bytes
            b16 = pos - (c0 - head)
            wrong = np.nonzero(outs[0][pos:pos + 16] != outs[1][pos:pos + 16])[0]
            seen.setdefault((int(order[ci]), c0, end - c0, int((chunks[order[ci], 1] >> np.uint64(48)) & np.uint64(0x7FF))), []).append((b16 >> 10, (b16 >> 4) & 63, wrong.tolist()))
        for key, v in list(seen.items())[:6]:
            print("chunk", key[0], "dst", key[1], "bytes", key[2], "n", key[3], "blocks(row,lane,bytes):", v[:8], "..." if len(v) > 8 else "", len(v))
    return 0 if bad.size == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
