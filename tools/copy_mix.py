#!/usr/bin/env python3
"""How a small streamed HBM read per workgroup (the stitch kernel's descriptor traffic) disturbs a write-saturated copy
(v2p_copy_mix_launch).  Prints ms and TB/s written per configuration of the read stream."""
import ctypes
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vcf2prot_amd import _native as N  # noqa: E402

lib = N.bench_lib()
dev = torch.device("cuda", 0)
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
nbytes = int(gb * (1 << 30)) // (240 * (32 << 10)) * (240 * (32 << 10))      # whole 'haplotypes' of 240 spans
window = 8 << 20
src = torch.randint(0, 255, (window + (1 << 17),), dtype=torch.uint8, device=dev)
out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
n_span = nbytes // (32 << 10)
dsc = torch.zeros(n_span * 4096 + (1 << 20), dtype=torch.uint8, device=dev)          # its own allocation
st = torch.cuda.current_stream()
rows = []


def run(bpl, every, stride, flags, label):
    ms = []
    for r in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        rc = lib.v2p_copy_mix_launch(ctypes.c_void_p(st.cuda_stream), src.data_ptr(), window, 5, out.data_ptr(), nbytes, dsc.data_ptr(), bpl, every, stride, flags)
        assert rc == 0
        e1.record(st)
        torch.cuda.synchronize()
        if r:
            ms.append(e0.elapsed_time(e1))
    med = statistics.median(ms)
    lanes = (every >> 16) or (64 if flags & 2 else 256)
    read_mb = (n_span // (every & 0xFFFF)) * lanes * bpl / 1e6 if bpl else 0
    rows.append({"label": label, "bytes_per_lane": bpl, "every": every, "stride": stride, "flags": flags, "ms": med, "TBps_written": nbytes / med / 1e9, "read_MB": read_mb})
    print(f"{label:58s} read {read_mb:7.1f} MB  {med:7.3f} ms  {nbytes / med / 1e9:5.2f} TB/s")


run(0, 1, 0, 0, "no descriptor stream")
for L in (64, 72, 80, 88, 96, 104, 112, 128):
    run(16, 1 | (L << 16), 16 * L, 16, f"16 B x {L} lanes per workgroup ({16 * L} B = {100 * 16 * L / 32768:.1f} %), 8 gathers then 8 stores")
run(0, 1, 0, 16, "no descriptor stream, 8 gathers then 8 stores per wave")
run(4, 1, 1024, 16, "4 B/lane, 8 gathers then 8 stores per wave")
run(8, 1, 2048, 16, "8 B/lane, 8 gathers then 8 stores per wave")
run(8, 1, 2048, 16 | 8, "8 B/lane, 8 gathers then 8 stores, XCD-permuted spans")
run(8, 1, 2048, 16 | (40 << 8), "8 B/lane, 8 gathers then 8 stores, 4 workgroups per CU")
run(0, 1, 0, 8, "no descriptor stream, XCD-permuted span order")
run(4, 1, 1024, 8, "4 B/lane, XCD-permuted span order")
run(8, 1, 2048, 8, "8 B/lane, XCD-permuted span order")
for bpl in (4, 8, 16):
    run(bpl, 1, 256 * bpl, 0, f"{bpl} B/lane, every workgroup, contiguous")
run(8, 2, 2048, 0, "8 B/lane, every 2nd workgroup")
run(8, 4, 2048, 0, "8 B/lane, every 4th workgroup")
run(16, 2, 4096, 0, "16 B/lane, every 2nd workgroup (same bytes as 8 B all)")
run(4, 1, 2048, 0, "4 B/lane, pieces 2 KiB apart")
run(8, 1, 4096, 0, "8 B/lane, pieces 4 KiB apart")
run(8, 1, 2048, 1, "8 B/lane, non-temporal loads")
run(16, 1, 1024, 2, "16 B/lane by one wave (1 KiB per workgroup)")
run(16, 1, 2048, 2, "16 B/lane by one wave, pieces 2 KiB apart")
run(8, 1, 2048, 4, "8 B/lane, requested after the first store")
run(4, 1, 1024, 4, "4 B/lane, requested after the first store")
run(0, 1, 0, 0, "no descriptor stream (again)")
for kib, wg in ((20, 8), (22, 7), (26, 6), (32, 5), (40, 4), (53, 3), (80, 2)):
    run(0, 1, 0, kib << 8, f"no descriptor stream, {wg} workgroups per CU")
    run(8, 1, 2048, kib << 8, f"8 B/lane, {wg} workgroups per CU")
run(16, 1, 4096, 40 << 8, "16 B/lane, 4 workgroups per CU")
run(16, 1, 4096, 80 << 8, "16 B/lane, 2 workgroups per CU")
def ballast(valu8, lds, bar):
    return (valu8 | (lds << 8) | (bar << 16)) << 16


for valu8, lds, bar in ((0, 0, 4), (0, 4, 0), (0, 8, 0), (6, 0, 0), (12, 0, 0), (25, 0, 0), (50, 0, 0), (12, 8, 4), (25, 8, 4)):
    run(0, 1, ballast(valu8, lds, bar), 0, f"ballast: {8 * valu8} VALU + {lds} LDS round trips per pass, {bar} barriers")
    if valu8 in (12, 25) and lds:
        run(8, 1, 2048 | ballast(valu8, lds, bar), 0, f"  same + 8 B/lane descriptor stream")
        run(4, 1, 1024 | ballast(valu8, lds, bar), 0, f"  same + 4 B/lane descriptor stream")
json.dump(rows, open(os.path.join("gpurun_out", "copy_mix.json"), "w"), indent=1)
