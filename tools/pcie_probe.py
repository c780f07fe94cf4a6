#!/usr/bin/env python3
"""PCIe rates of pinned host buffers by where they were allocated (development probe): GB/s of 64 MB copies H2D / D2H for a buffer
allocated while the process was bound to one CPU of each NUMA node."""
import os
import sys
import time

import torch

def rate(t_host, t_dev, n=10):
    s = torch.cuda.Stream()
    out = {}
    with torch.cuda.stream(s):
        for name, (a, b) in {"h2d": (t_dev, t_host), "d2h": (t_host, t_dev)}.items():
            a.copy_(b, non_blocking=True); s.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                a.copy_(b, non_blocking=True)
            s.synchronize()
            out[name] = n * t_host.numel() / (time.perf_counter() - t0) / 1e9
    return out

def main():
    ncpu = os.cpu_count()
    dev = torch.device("cuda", 0)
    d = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    try:
        nodes = sorted(int(x[4:]) for x in os.listdir("/sys/devices/system/node") if x.startswith("node"))
    except OSError:
        nodes = []
    print("cpus", ncpu, "numa nodes", nodes)
    for cpu in sorted(set([0, ncpu // 4, ncpu // 2, 3 * ncpu // 4, ncpu - 1])):
        try:
            os.sched_setaffinity(0, {cpu})
        except OSError as e:
            print("affinity", cpu, e); continue
        h = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
        h.fill_(1)
        print("alloc on cpu", cpu, {k: round(v, 1) for k, v in rate(h, d).items()})
        del h
    os.sched_setaffinity(0, set(range(ncpu)))
    for mb in (1, 8, 64, 256):
        h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory(); h.fill_(1)
        dd = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
        print(mb, "MB", {k: round(v, 1) for k, v in rate(h, dd).items()})

main()
