#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory into the files committed under profiles/."""
import csv
import glob
import json
import os
import sys

src, tag, out = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(out, exist_ok=True)


def stitch_rows(path):
    return [r for r in csv.DictReader(open(path)) if "stitch" in r["Kernel_Name"] and "_kernel" in r["Kernel_Name"]]


summary = {"tag": tag, "command": "python3 bench.py --no-cpu-baseline --no-pcie " + " ".join(sys.argv[4:])}
stats = os.path.join(src, "trace", "trace_kernel_stats.csv")
if os.path.exists(stats):
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:
            r["Name"] = r["Name"][:110]
            w.writerow(r)
    for r in rows:
        if "stitch" in r["Name"] and "_kernel" in r["Name"]:
            summary.setdefault("stitch_kernels", []).append({"name": r["Name"][:60], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                                              "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]), "pct_of_gpu_time": float(r["Percentage"])})
counters = {}
for f in glob.glob(os.path.join(src, "pmc_*", "*counter_collection.csv")):
    for r in stitch_rows(f):
        counters.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
summary["counters_per_launch"] = {k: sum(v) / len(v) for k, v in sorted(counters.items())}
c = summary["counters_per_launch"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB-ish units of 1024 B; on gfx950 FETCH_SIZE
    # reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is exact for 16-B/lane stores.
    fetch = 2.0 * c["FETCH_SIZE"] * 1024.0
    write = c["WRITE_SIZE"] * 1024.0
    summary["hbm"] = {"fetch_bytes_corrected_x2": fetch, "write_bytes": write, "traffic_bytes_per_launch": fetch + write,
                      "note": "separate --pmc passes; FETCH_SIZE doubled per the gfx950 correction (calibration: descriptor "
                              "stream + proteome first touch + payload ~ 1.14e9 B expected, 2*FETCH_SIZE reads the same)"}
if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
    summary["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
ceil = os.path.join(src, "hbm_ceiling.json")
if os.path.exists(ceil):
    summary["measured_ceilings_GBs"] = json.load(open(ceil))
log = os.path.join(src, "bench_trace.log")
if os.path.exists(log):
    for line in open(log):
        if line.startswith("{"):
            summary["bench_line_under_profiler"] = json.loads(line)
json.dump(summary, open(os.path.join(out, f"{tag}_summary.json"), "w"), indent=1)
if "hbm" in summary:
    bl = summary.get("bench_line_under_profiler", {})
    wl = (bl.get("config", {}).get("workload", "C2:") or "C2:").split(":")[0]
    tpath = os.path.join(out, "traffic_latest.json")
    allw = json.load(open(tpath)) if os.path.exists(tpath) else {}
    allw[wl] = {"workload": wl, "haplotypes": bl.get("config", {}).get("haplotypes_rank0"),
                "hbm_bytes_per_launch": summary["hbm"]["traffic_bytes_per_launch"], "source": f"profiles/{tag}_summary.json"}
    json.dump(allw, open(tpath, "w"), indent=1)
print(json.dumps({k: summary[k] for k in summary if k not in ("bench_line_under_profiler",)}, indent=1)[:3000])
