#!/usr/bin/env python3
"""In-kernel s_memtime stamps of the stitch kernel (diagnostic build, DBG=20): where a workgroup's
lifetime goes.  Never quote this build's run time; read the shares."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vcf2prot_amd import _native as N
from vcf2prot_amd.cohort import Cohort
lib = N.bench_lib(); dev = torch.device("cuda", 0)      # (s_memtime stamps are a V2P_BENCH_VARIANTS instantiation)
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
c = Cohort.preset(wl, n_samples=int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
var = int(sys.argv[3]) if len(sys.argv) > 3 else 0
img = c.pack(0, c.n_haplotypes, n_threads=64, kernel=2 if var in (1, 2) else 1)
prot = c.proteome()
chunks = np.ascontiguousarray(img.chunks)
lib.v2p_order_chunks_for_xcds(chunks.ctypes.data, chunks.shape[0], img.desc.ctypes.data, img.desc.size, prot.size)
def padded(a):
    t = torch.zeros(a.size + 128, dtype=torch.uint8, device=dev); t[64:64 + a.size] = torch.from_numpy(a).to(dev); return t
d_prot, d_pay = padded(prot), padded(img.payload)
d_desc = torch.zeros(img.desc.size + 16, dtype=torch.int64, device=dev); d_desc[8:8 + img.desc.size] = torch.from_numpy(img.desc.view(np.int64)).to(dev); d_chunks = torch.from_numpy(chunks.view(np.int64)).to(dev)
nch = chunks.shape[0]
dbg_bytes = nch * 4 * 64
bits = int(lib.v2p_stitch_launch_bits(chunks.ctypes.data, nch))
print(wl, 'chunks', nch, 'launch bits', hex(bits), 'out', img.out_bytes)
d_out = torch.zeros(img.out_bytes + 512 + dbg_bytes, dtype=torch.uint8, device=dev)
d_status = torch.full((1,), -1, dtype=torch.int64, device=dev)
s = torch.cuda.current_stream().cuda_stream
for flags in (1 | (var << 12), 1 | (var << 12) | (20 << 16)):
    lib.v2p_stitch_launch(ctypes.c_void_p(s), d_desc.data_ptr() + 64, img.desc.size, d_chunks.data_ptr(), nch, d_prot.data_ptr() + 64, prot.size,
                          d_pay.data_ptr() + 64, img.payload.size, d_out.data_ptr(), img.out_bytes, d_status.data_ptr(), flags | bits, 0)
torch.cuda.synchronize()
off = (img.out_bytes + 255) // 256 * 256
st = d_out[off:off + dbg_bytes].cpu().numpy().view(np.uint64).reshape(nch, 4, 8).astype(np.int64)
if var == 0:          # stitch4: its own stamp layout
    q = [st[:, :, k] for k in range(7)]
    ok = (q[6] > q[0]).all(axis=1)
    print("chunks stamped:", int(ok.sum()), "of", nch)
    def show4(name, x):
        x = x[ok].reshape(-1)
        print(f"{name:44s} median {np.median(x):9.0f}  mean {x.mean():9.0f}  p90 {np.percentile(x, 90):9.0f} cycles")
    show4("round start -> A done, scans issued", q[1] - q[0])
    show4("barrier + B + barrier + C/D", q[2] - q[1])
    show4("P (merge pass)", q[3] - q[2])
    show4("barrier after P", q[4] - q[3])
    show4("bulk: first look-up -> last store issued", q[5] - q[4])
    show4("  of which look-ups + gather issue", st[:, :, 7] >> 32)
    show4("  of which gathers / patches arriving", st[:, :, 7] & 0xFFFFFFFF)
    show4("  of which store issue", (q[5] - q[4]) - (st[:, :, 7] >> 32) - (st[:, :, 7] & 0xFFFFFFFF))
    show4("last store issued -> stores acked", q[6] - q[5])
    show4("whole round", q[6] - q[0])
    sys.exit(0)
t0, t1, t2, t3, t4 = (st[:, :, k] for k in range(5))
ok = (t4 > t0).all(axis=1)
print("chunks stamped:", int(ok.sum()), "of", nch)
def show(name, x):
    x = x[ok].reshape(-1)
    print(f"{name:34s} median {np.median(x):9.0f}  mean {x.mean():9.0f}  p90 {np.percentile(x, 90):9.0f} cycles")
show("entry -> descriptors arrived", t1 - t0)
show("descriptors -> tables ready (A-D)", t2 - t1)
show("K2: tables ready -> last store issued", t3 - t2)
show("last store issued -> stores acked", t4 - t3)
show("whole wave", t4 - t0)
show("K2 sum: block lookup (LDS chain)", st[:, :, 5])
show("K2 sum: gather issue -> data", st[:, :, 6])
show("K2 sum: merge + store issue", st[:, :, 7])
