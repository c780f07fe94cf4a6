"""What `bench.py --gpus N` (strong scaling, the north star's C3 cohort) will time on every rank, measured on ONE GPU: for
N = 1, 2, 4, 8 the ranges `shard.shard_by_bytes` gives the ranks are run one after the other through bench.cohort_leg -- the
same code a rank runs: the resident stream, the ONE call, then the timed re-executes -- and the slowest range is the N-rank
step (ranks share nothing on the data path: no collective inside the loop).  predicted_efficiency =
T(1) / (N * max_r T(N, r)): what is lost is the ranges' imbalance and whatever a smaller image costs per byte (fewer, shorter
phases; launch gaps).  It is a prediction from one device, not a measurement of eight -- the driver's SCALE_rNN.json is that.

    python tools/strong_scaling_probe.py [--workload C3] [--samples 0] [--steps 20] [--ns 1,2,4,8] > profiles/rNN_strong_scaling_probe.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--samples", type=int, default=0)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ns", default="1,2,4,8")
    ap.add_argument("--settle-ms", type=float, default=50.0, help="bench.py --clock-settle-ms")
    ap.add_argument("--repeat", type=int, default=2, help="passes per range (N > 1); the faster one counts")
    a = ap.parse_args()
    import torch
    if not torch.cuda.is_available():
        sys.exit("needs an MI355X")
    import bench
    from vcf2prot_amd import build
    build.build_all()
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.shard import shard_by_bytes
    samples = a.samples or bench.DEFAULT_SAMPLES["strong"][a.workload]
    n_threads = max(1, min(64, os.cpu_count() or 1))
    cohort = Cohort.preset(a.workload, n_samples=samples)
    sizes = cohort.result_sizes(0, cohort.n_haplotypes, n_threads=n_threads)
    out = {"workload": a.workload, "samples": samples, "haplotypes": cohort.n_haplotypes, "steps": a.steps, "clock_settle_ms": a.settle_ms, "points": []}
    t1 = None
    for n in [int(x) for x in a.ns.split(",")]:
        ranges = shard_by_bytes(sizes.tolist(), n)
        ranks = []
        for r, (h0, h1) in enumerate(ranges):
            # (every range twice, the faster pass kept: one device plays the ranks one after the other, and a single pass of a range that
            # takes a millisecond and a half is at the mercy of whatever the box does beside it)
            legs = [bench.cohort_leg(a.workload, samples, h0, h1, a.steps, a.warmup, n_threads, "none", host_packed=False, label=f"rank {r} of {n}", settle_ms=a.settle_ms)
                    for _ in range(a.repeat if n > 1 else 1)]
            leg = min(legs, key=lambda x: x["elapsed_s"])
            k = leg["kernel_ms"]
            ranks.append({"rank": r, "haplotypes": h1 - h0, "result_bytes": leg["result_bytes"], "ms_per_step_wall": 1e3 * leg["elapsed_s"] / len(k),
                          "kernel_ms_avg": sum(k) / len(k), "one_shot_total_ms": min(x["one_shot"]["total_ms"] for x in legs),
                          "one_shot_total_ms_gpu_busy_before": min(x["one_shot"]["total_ms_gpu_busy_before"] for x in legs), "passes": len(legs),
                          "image_form_timed": leg.get("image_form_timed")})
        worst = max(x["ms_per_step_wall"] for x in ranks)
        worst1 = max(x["one_shot_total_ms_gpu_busy_before"] for x in ranks)
        if t1 is None and n == 1:
            t1 = (worst, worst1)
        b = [x["result_bytes"] for x in ranks]
        p = {"n": n, "step_ms_slowest_rank": worst, "one_shot_ms_slowest_rank": worst1, "bytes_imbalance": max(b) / (sum(b) / len(b)) - 1.0, "ranks": ranks}
        if t1:
            p["predicted_efficiency"] = t1[0] / (n * worst)
            p["predicted_efficiency_one_shot"] = t1[1] / (n * worst1)
        out["points"].append(p)
        print(json.dumps({k: v for k, v in p.items() if k != "ranks"}), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
