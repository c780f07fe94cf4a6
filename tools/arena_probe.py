#!/usr/bin/env python3
"""Does an image's execute time depend on WHERE its buffers sit?  The same host-packed image in several batches of one process (all
alive at once: every batch gets its own arena, descriptor array and chunk table), each timed alone; prints the arena's device pointer
beside the time.     python tools/arena_probe.py --workload C2 [--samples N] [--batches 6] [--reps 12]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--samples", type=int, default=0)
    ap.add_argument("--batches", type=int, default=6)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--device-built", action="store_true")
    ap.add_argument("--order-blocks", default="", help="comma list: the host-packed image's chunk table dealt to the XCDs inside this many equal blocks of the arena (one batch per value, repeated --batches times)")
    ap.add_argument("--phase-mb", default="0", help="comma list of phase sizes (MB of image per phase; 0 = the library's choice) tried on every batch")
    ap.add_argument("--sc1", default="-1", help="comma list of store policies (-1 = the library's choice, 0 / 1 = plain nt / sc1 nt row stores)")
    a = ap.parse_args()
    import numpy as np
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.engine import Context
    samples = a.samples or {"C2": 1000, "C3": 10000, "C4": 2504, "C5": 50000}[a.workload]
    c = Cohort.preset(a.workload, n_samples=samples)
    nt = max(1, min(64, os.cpu_count() or 1))
    rows = []
    ob = [int(x) for x in a.order_blocks.split(",")] if a.order_blocks else []
    with Context(0, result_order=bool(ob)) as ctx:
        ctx.upload_proteome(c.proteome())
        batches = []
        if a.device_built:
            st = c.txstream(0, c.n_haplotypes, n_threads=nt)
            rs = ctx.upload_stream(st)
            st.close()
        else:
            img = c.pack(0, c.n_haplotypes, n_threads=nt)
        import ctypes
        from vcf2prot_amd import _native as N
        labels = []
        for k in range(a.batches * max(1, len(ob))):
            b = ctx.batch()
            if ob:
                nb = ob[k % len(ob)]
                labels.append(nb)
                ch = np.ascontiguousarray(img.chunks).copy()
                n = ch.shape[0]
                rot = 0
                if nb < 0:                                  # one order for the whole table, XCD x's sequence rotated by x * rot / 64 of its length
                    rot, nb = -nb, 1
                if nb == 0:                                 # haplotype-major inside an XCD's proteome slice (the development library's V2P_ORDER_WINDOWS=0), one order
                    os.environ["V2P_ORDER_WINDOWS"] = "0"; os.environ["V2P_ORDER_MAX_BLOCKS"] = "1"
                    N.bench_lib().v2p_order_chunks_for_xcds(ch.ctypes.data, n, img.desc.ctypes.data, img.desc.size, c.proteome().size)
                    del os.environ["V2P_ORDER_WINDOWS"], os.environ["V2P_ORDER_MAX_BLOCKS"]
                hap = nb >= 100000                          # 100000 + nb: nb blocks, haplotype-major inside each (development library's order)
                if hap:
                    nb -= 100000
                    os.environ["V2P_ORDER_WINDOWS"] = "0"; os.environ["V2P_ORDER_MAX_BLOCKS"] = "1"
                for j in range(nb):
                    c0, c1 = (n * j // nb) & ~7, ((n * (j + 1) // nb) & ~7) if j + 1 < nb else n
                    sub = np.ascontiguousarray(ch[c0:c1])
                    (N.bench_lib() if hap else N.hip_lib()).v2p_order_chunks_for_xcds(sub.ctypes.data, sub.shape[0], img.desc.ctypes.data, img.desc.size, c.proteome().size)
                    ch[c0:c1] = sub
                if hap:
                    del os.environ["V2P_ORDER_WINDOWS"], os.environ["V2P_ORDER_MAX_BLOCKS"]
                if rot:
                    m = n // 8 * 8
                    seqs = [ch[x:m:8].copy() for x in range(8)]
                    for x in range(8):
                        sh = (len(seqs[x]) * x * rot // 64) % max(1, len(seqs[x]))
                        seqs[x] = np.roll(seqs[x], -sh, axis=0)
                    for x in range(8):
                        ch[x:m:8] = seqs[x]
                b.set_packed(img.desc, ch, img.payload, img.hap_out_begin)
                b.finalize(); b.execute(); b.sync()
                batches.append(b)
                continue
            if a.device_built:
                b.build_and_execute(rs, 0, 0)
            else:
                b.set_packed(img.desc, img.chunks, img.payload, img.hap_out_begin)
                b.finalize()
                b.execute()
            b.sync()
            batches.append(b)
        dig0 = batches[0].digests()
        for rnd in range(2):
            for k, b in enumerate(batches):
                res = {}
                for pm in [int(x) for x in a.phase_mb.split(",")]:
                    for sc1 in [int(x) for x in a.sc1.split(",")]:
                        ctx.set_launch_opts(phase_bytes=pm << 20, store_sc1=sc1)
                        for _ in range(3):
                            b.execute()
                        b.sync()
                        t0 = time.perf_counter()
                        for _ in range(a.reps):
                            b.execute()
                        b.sync()
                        res[f"{pm}MB/sc1={sc1}"] = round((time.perf_counter() - t0) / a.reps * 1e3, 4)
                ctx.set_launch_opts()
                ptr = b.device_out()
                row = {"round": rnd, "batch": k, "order_blocks": labels[k] if labels else None, "arena": hex(ptr), "ms": res, "same_digests": bool(np.array_equal(b.digests(), dig0))}
                print(json.dumps(row), flush=True)
                rows.append(row)
        for b in batches:
            b.close()


if __name__ == "__main__":
    main()
