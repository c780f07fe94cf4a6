#!/usr/bin/env python3
"""Does a bare fill's rate depend on the allocation?  N buffers of G GiB alive at once, torch fill_ timed on each (three rounds)."""
import sys
import torch
n, g = int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 16
bufs = [torch.empty(g << 30, dtype=torch.uint8, device="cuda") for _ in range(n)]
for rnd in range(3):
    out = []
    for b in bufs:
        b.fill_(1); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            b.fill_(46)
        e1.record(); torch.cuda.synchronize()
        out.append(round(4 * (g << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e12, 3))
    print("round", rnd, "TB/s per buffer:", out, [hex(b.data_ptr()) for b in bufs] if rnd == 0 else "")
