#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory into the files committed under profiles/."""
import csv
import glob
import json
import os
import sys

src, tag, out = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(out, exist_ok=True)


def is_step_kernel(name):
    """kernels of one execute() of the image: the stitch kernels and the read-ahead kernel of phase 0 (launch_stitch: phases)"""
    return ("stitch" in name and "_kernel" in name) or "touch_image_kernel" in name


def stitch_rows(path):
    return [r for r in csv.DictReader(open(path)) if is_step_kernel(r["Kernel_Name"])]


summary = {"tag": tag, "command": "python3 bench.py --no-cpu-baseline --no-pcie --no-c2 --no-host-packed --verify sample --steps 10 --warmup 3 " + " ".join(sys.argv[4:])}
# the bench line of the traced run: how many steps it timed (nothing here assumes 10)
bench_line = {}
_log = os.path.join(src, "bench_trace.log")
if os.path.exists(_log):
    for _line in open(_log):
        if _line.startswith("{"):
            bench_line = json.loads(_line)


def n_builds(rows):
    """image builds in a trace: one rows_parse_kernel launch each, the writing pass of a two-pass build (instance <.., 2>) not counted
    (the tile-bytes kernels no longer say: a resident stream brings its tile tables, made once at its upload)"""
    import re
    return sum(1 for r in rows if "rows_parse_kernel" in r["Kernel_Name"] and not re.search(r"rows_parse_kernel<[^>]*, 2>", r["Kernel_Name"])) or 1
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
        if is_step_kernel(r["Name"]):
            summary.setdefault("stitch_kernels", []).append({"name": r["Name"][:60], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"]) if "TotalDurationNs" in r else float(r["AverageNs"]) * int(r["Calls"]),
                                                              "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]), "pct_of_gpu_time": float(r["Percentage"])})
    # one execute() = one STEP = the launches of all its phases (launch_stitch cuts a wave / long-run image into phases of 64 MB of
    # image; a pure wave image has ONE touch_image_kernel launch per step, the later read-aheads ride on the stitch launches)
    ks = summary.get("stitch_kernels", [])
    touch = [k for k in ks if "touch_image" in k["name"]]
    if touch and touch[0]["calls"]:
        steps = touch[0]["calls"]
        summary["steps_profiled"] = steps
        summary["launches_per_step"] = {k["name"][:40]: k["calls"] / steps for k in ks}
        summary["kernel_ms_per_step"] = sum(k["total_ns"] for k in ks) / steps / 1e6
# the TIMED steps alone (the run also executes the image right after its build, on a fresh arena, and warms up): every execute begins
# with a touch_image_kernel launch; the last `steps` executes of the trace are the ones bench.py times
trace = glob.glob(os.path.join(src, "trace", "*kernel_trace.csv"))
if trace:
    all_rows = list(csv.DictReader(open(trace[0])))
    rows = [r for r in all_rows if is_step_kernel(r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    execs, build = [], {}
    phased = any("touch_image" in r["Kernel_Name"] for r in rows)      # (dense / per-block images: one launch per execute, no read-ahead kernel)
    for r in rows:
        if "touch_image" in r["Kernel_Name"] or not execs or not phased:
            execs.append(0.0)
        execs[-1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    k = int(bench_line.get("steps", 10))
    builds = n_builds(all_rows)
    summary["image_builds_in_trace"] = builds
    if len(execs) >= k:
        summary["timed_steps"] = {"steps": k, "kernel_ms_per_step": sum(execs[-k:]) / k, "min": min(execs[-k:]), "max": max(execs[-k:]), "all_executes_ms": [round(x, 3) for x in execs]}
    for r in csv.DictReader(open(trace[0])):
        n = r["Kernel_Name"]
        if any(t in n for t in ("rows_", "scan_", "sub_", "xcd_")):
            key = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:]
            build.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    if build:
        # a PADDED image (rich streams, round 5) has no compaction inside the one call: rows_compact_kernel / rows_chunks_dense_kernel then belong to
        # densify() -- once per batch that is executed again or downloaded -- and are reported by themselves, per conversion
        dens = [k2 for k2 in build if "rows_chunks_dense_kernel" in k2]
        if dens:
            n_dens = len(build[dens[0]])
            dk = [k2 for k2 in build if "rows_chunks_dense_kernel" in k2 or "rows_compact_kernel" in k2]
            summary["densify_kernels_ms_per_conversion"] = {k2: sum(build[k2]) / n_dens for k2 in dk}
            summary["densify_conversions_in_trace"] = n_dens
            for k2 in dk:
                del build[k2]
        summary["build_kernels_ms_per_build"] = {k2: sum(v) / builds for k2, v in sorted(build.items(), key=lambda kv: -sum(kv[1]))}
        summary["build_kernels_ms_total_per_build"] = sum(sum(v) for v in build.values()) / builds
counters, n_steps = {}, {}
for f in glob.glob(os.path.join(src, "pmc_*", "*counter_collection.csv")):
    rows = stitch_rows(f)
    steps = len({r["Dispatch_Id"] for r in rows if "touch_image" in r["Kernel_Name"]}) or len({r["Dispatch_Id"] for r in rows})   # (no read-ahead kernel: one launch per execute)
    for r in rows:
        counters[r["Counter_Name"]] = counters.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        n_steps[r["Counter_Name"]] = steps
# the image-build kernels of the same run (build_rows.hip): FETCH_SIZE / WRITE_SIZE per build (the run builds the image twice)
bc, pmc_builds = {}, {}
for f in glob.glob(os.path.join(src, "pmc_*", "*counter_collection.csv")):
    frows = list(csv.DictReader(open(f)))
    for cn in {r["Counter_Name"] for r in frows}:
        pmc_builds[cn] = n_builds([r for r in frows if r["Counter_Name"] == cn])
    for r in frows:
        n = r["Kernel_Name"]
        if any(t in n for t in ("rows_", "scan_", "sub_", "xcd_")) and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            key = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:]
            bc.setdefault(key, {}).setdefault(r["Counter_Name"], 0.0)
            bc[key][r["Counter_Name"]] += float(r["Counter_Value"])
if bc:
    nbf, nbw = float(pmc_builds.get("FETCH_SIZE", 1) or 1), float(pmc_builds.get("WRITE_SIZE", 1) or 1)
    summary["image_builds_in_pmc_passes"] = pmc_builds
    if any("rows_chunks_dense_kernel" in k for k in bc):            # (densify's kernels: not part of a build, see above)
        summary["densify_traffic_bytes_all_conversions_of_a_pass"] = {k: {"fetch_bytes_corrected_x2": 2.0 * v.get("FETCH_SIZE", 0.0) * 1024.0, "write_bytes": v.get("WRITE_SIZE", 0.0) * 1024.0}
                                                                      for k, v in bc.items() if "rows_chunks_dense_kernel" in k or "rows_compact_kernel" in k}
        bc = {k: v for k, v in bc.items() if "rows_chunks_dense_kernel" not in k and "rows_compact_kernel" not in k}
    summary["build_traffic_bytes_per_build"] = {k: {"fetch_bytes_corrected_x2": 2.0 * v.get("FETCH_SIZE", 0.0) * 1024.0 / nbf, "write_bytes": v.get("WRITE_SIZE", 0.0) * 1024.0 / nbw} for k, v in sorted(bc.items())}
    summary["build_traffic_total_bytes_per_build"] = sum(v["fetch_bytes_corrected_x2"] + v["write_bytes"] for v in summary["build_traffic_bytes_per_build"].values())
# per STEP (= per execute() of the image: all phases' launches summed)
summary["counters_per_step"] = {k: v / max(n_steps[k], 1) for k, v in sorted(counters.items())}
c = summary["counters_per_step"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB-ish units of 1024 B; on gfx950 FETCH_SIZE
    # reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is exact for 16-B/lane stores.
    fetch = 2.0 * c["FETCH_SIZE"] * 1024.0
    write = c["WRITE_SIZE"] * 1024.0
    summary["hbm"] = {"fetch_bytes_corrected_x2": fetch, "write_bytes": write, "traffic_bytes_per_step": fetch + write,
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
                "hbm_bytes_per_launch": summary["hbm"]["traffic_bytes_per_step"], "source": f"profiles/{tag}_summary.json"}
    json.dump(allw, open(tpath, "w"), indent=1)
print(json.dumps({k: summary[k] for k in summary if k not in ("bench_line_under_profiler",)}, indent=1)[:3000])
