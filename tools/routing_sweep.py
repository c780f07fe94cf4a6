#!/usr/bin/env python3
"""Are the library's routing rules -- which kernel takes an image (wave from v2p_routing_rules().wave_bytes_per_task = 24 result bytes per task, dense below), phase size (28 MB
for images whose descriptors are more than 3 % of their result, else 64 MB), store policy ("sc1 nt" for thin images), block order
(one order for thin images) -- near the best forced choice AWAY from the four cohorts they were measured on?

    python tools/routing_sweep.py [--quick] > profiles/r04_routing_sweep.json

Grid: transcript length {150, 250, 400, 800, 1600} x alterations per altered transcript {1, 2, 3, 4, 6, 8, 12, 16} x proteome {8, 56} MB
(a transcript enters a haplotype's Task vector only if it is altered, so "0.25 alterations per transcript" is not a point of this
boundary).  Per point, in ONE process, alternating: the host packer's own choice with the launcher's defaults (`lib`), the
device-built rows image (`rows`, what the product ships), and every forced combination; median of the alternated runs.
"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="a 3 x 3 x 1 sub-grid")
    ap.add_argument("--target-gb", type=float, default=1.5)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--L", type=int, nargs="*", help="transcript lengths instead of the grid's")
    ap.add_argument("--K", type=int, nargs="*", help="alteration counts instead of the grid's")
    ap.add_argument("--P", type=int, nargs="*", help="proteome sizes (MB) instead of the grid's")
    a = ap.parse_args()
    import torch
    from vcf2prot_amd import build, _native as N
    build.build_all()
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.engine import Context
    from vcf2prot_amd.txstream import build_on_device_auto
    blib = N.bench_lib()                     # (v2p_order_chunks_for_xcds with V2P_ORDER_MAX_BLOCKS: the forced one-block order)
    Ls = [150, 400, 1600] if a.quick else [150, 250, 400, 800, 1600]
    Ks = [1, 4, 16] if a.quick else [1, 2, 3, 4, 6, 8, 12, 16]
    Ps = [8] if a.quick else [8, 56]
    Ls, Ks, Ps = a.L or Ls, a.K or Ks, a.P or Ps
    nt = min(64, os.cpu_count() or 1)
    points = []
    for P in Ps:
        for L in Ls:
            for K in Ks:
                T = max(1000, int(P * 1e6 / L))
                altered = min(T, 5000)
                haps = max(8, int(a.target_gb * 1e9 / (altered * L)))
                c = Cohort.preset("C3", mean_len=float(L), n_transcripts=T, alts_fixed=K, altered_per_hap=altered, n_samples=(haps + 1) // 2)
                n_h = c.n_haplotypes
                prot = c.proteome()
                res_bytes = int(c.result_sizes(0, n_h, n_threads=nt).sum())
                variants = {}
                # two contexts: `ctx` orders chunk tables itself (the product), `cro` launches them as given (forced orders)
                with Context(0, development=True) as ctx, Context(0, result_order=True, development=True) as cro:
                    ctx.upload_proteome(prot)
                    cro.upload_proteome(prot)
                    ts = torch.cuda.Stream()

                    def forced_batch(kernel, one_block=False):
                        img = c.pack(0, n_h, n_threads=nt, kernel=kernel)
                        ch = np.ascontiguousarray(img.chunks)
                        if one_block:
                            os.environ["V2P_ORDER_MAX_BLOCKS"] = "1"
                        try:
                            blib.v2p_order_chunks_for_xcds(ch.ctypes.data, ch.shape[0], img.desc.ctypes.data, img.desc.size, prot.size)
                        finally:
                            os.environ.pop("V2P_ORDER_MAX_BLOCKS", None)
                        bb = cro.batch()
                        bb.set_packed(img.desc, ch, img.payload, img.hap_out_begin)
                        bb.finalize()
                        return bb, 8.0 * img.desc.size > 0.03 * res_bytes
                    base_img = c.pack(0, n_h, n_threads=nt)
                    bits = base_img.launch_bits
                    lib_kernel = 4 if bits & 4 else (3 if bits & 2 else 2)
                    bpt = res_bytes / max(base_img.n_tasks, 1)
                    # (name, context, batch, launch opts)
                    todo = []
                    b = ctx.batch(); b.set_packed(base_img.desc, base_img.chunks, base_img.payload, base_img.hap_out_begin); b.finalize()
                    todo.append(("lib", ctx, b, {}))
                    st = c.txstream(0, n_h, n_threads=nt)
                    rb = ctx.batch()
                    info = build_on_device_auto(rb, st, res_bytes)
                    st.close()
                    todo.append(("rows", ctx, rb, {}))
                    del base_img
                    refused = {}
                    for kernel in (4, 3, 2):
                        try:
                            bb, rich = forced_batch(kernel)
                        except Exception as e:                        # (e.g. a wave image of 5-byte tasks: the packer refuses -- recorded with the point)
                            refused[f"kernel{kernel}"] = repr(e)[:200]
                            continue
                        if kernel == 4:
                            for phase in (28, 64):
                                for sc1 in (0, 1):
                                    todo.append((f"wave,phase={phase},sc1={sc1}", cro, bb, dict(phase_bytes=phase << 20, store_sc1=sc1)))
                            b1, _ = forced_batch(4, one_block=True)
                            todo.append(("wave,one-block", cro, b1, dict(phase_bytes=(28 if rich else 64) << 20, store_sc1=0 if rich else 1)))
                        else:
                            todo.append(("dense" if kernel == 3 else "per-block", cro, bb, {}))
                    ok, errors, ref_dig = [], dict(refused), None
                    for name, cx, bb, opts in todo:                   # warm every image once; every variant must produce the same haplotypes
                        try:
                            cx.set_launch_opts(**opts); bb.execute(); bb.sync()
                            dig = bb.digests()
                            if ref_dig is None:
                                ref_dig = dig
                            elif not np.array_equal(dig, ref_dig):
                                raise RuntimeError("digests differ from the library's image")
                            ok.append((name, cx, bb, opts))
                        except Exception as e:                        # (recorded, not hidden: `errors` of the point)
                            errors[name] = repr(e)[:300]
                            print(f"L={L} K={K} P={P} {name}: {errors[name]}", file=sys.stderr, flush=True)
                    all_batches = [bb for _, _, bb, _ in todo]
                    todo = ok
                    times = {name: [] for name, _, _, _ in todo}
                    ctx.set_stream(ts.cuda_stream); cro.set_stream(ts.cuda_stream)
                    for _ in range(a.rounds):
                        for name, cx, bb, opts in todo:
                            cx.set_launch_opts(**opts)
                            bb.execute()                              # untimed: the two contexts hold a proteome copy each, and the variant that follows a switch would pay for a cold one
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record(ts); bb.execute(); e1.record(ts); bb.sync()
                            times[name].append(e0.elapsed_time(e1))
                    ctx.set_stream(0); cro.set_stream(0)
                    ctx.set_launch_opts(); cro.set_launch_opts()
                    seen = set()
                    for bb in all_batches:
                        if id(bb) not in seen:
                            seen.add(id(bb)); bb.close()
                    variants = {k: statistics.median(v) for k, v in times.items()}
                forced = {k: v for k, v in variants.items() if k not in ("lib", "rows")}
                if not forced or "lib" not in variants or "rows" not in variants:
                    points.append({"L": L, "alterations": K, "proteome_MB": P, "errors": errors, "ms": variants, "lib_over_best": 0.0, "rows_over_best": 0.0})
                    c.close(); torch.cuda.empty_cache()
                    continue
                best_name = min(forced, key=forced.get)
                pt = {"L": L, "alterations": K, "proteome_MB": P, "transcripts": T, "haplotypes": n_h, "result_bytes": res_bytes, "bytes_per_task": bpt,
                      "lib_kernel": {4: "wave", 3: "dense", 2: "per-block"}[lib_kernel], "rows_kernel": info["kernel"],
                      "ms": variants, "errors": errors, "best_forced": best_name, "lib_over_best": variants["lib"] / forced[best_name], "rows_over_best": variants["rows"] / forced[best_name]}
                points.append(pt)
                print(json.dumps(pt), file=sys.stderr, flush=True)
                c.close()
                torch.cuda.empty_cache()
    worst = max(points, key=lambda p: p["lib_over_best"])
    worst_rows = max(points, key=lambda p: p["rows_over_best"])
    print(json.dumps({"what": __doc__.split("\n\n")[0], "points": points, "worst_lib_over_best": worst["lib_over_best"], "worst_point": {k: worst[k] for k in ("L", "alterations", "proteome_MB")},
                      "worst_rows_over_best": worst_rows["rows_over_best"], "worst_rows_point": {k: worst_rows[k] for k in ("L", "alterations", "proteome_MB")}}, indent=1))


if __name__ == "__main__":
    main()
