#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC traffic passes for tools/decode_bench.py.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_decode
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
python3 tools/decode_bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o decode -- python3 tools/decode_bench.py --no-cpu-baseline "$@" > $OUT/bench_trace.json 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 tools/decode_bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python3 tools/decode_bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $OUT/bench_write.log 2>&1
python3 - <<'PY'
import csv, glob, json, os, re
def kname(k):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", k)
    return m.group(1) + (m.group(2) or "").replace("unsigned int", "u32").replace("unsigned long", "u64")
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "prof_decode")
per = {}
for f in glob.glob(os.path.join(out, "pmc_*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "v2p::" not in k:
            continue
        name = kname(k)
        per.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
summary = {"command": "python3 tools/decode_bench.py --no-cpu-baseline", "kernels": {}}
tot = 0.0
for name, c in sorted(per.items()):
    fetch = 2.0 * 1024.0 * sum(c.get("FETCH_SIZE", [0])) / max(1, len(c.get("FETCH_SIZE", [0])))     # gfx950 correction as for the stitch kernel
    write = 1024.0 * sum(c.get("WRITE_SIZE", [0])) / max(1, len(c.get("WRITE_SIZE", [0])))
    summary["kernels"][name] = {"hbm_fetch_bytes_corrected_x2": fetch, "hbm_write_bytes": write}
    tot += fetch + write
summary["hbm_bytes_per_pass"] = tot
stats = os.path.join(out, "trace", "decode_kernel_stats.csv")
if os.path.exists(stats):
    for r in csv.DictReader(open(stats)):
        if "v2p::" in r["Name"]:
            name = kname(r["Name"])
            summary["kernels"].setdefault(name, {})["avg_ns"] = float(r["AverageNs"])
try:
    summary["bench_line"] = json.loads(open(os.path.join(out, "bench.json")).read().strip().split("\n")[-1])
except Exception as e:
    summary["bench_line_error"] = str(e)
json.dump(summary, open(os.path.join(out, "decode_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "bench_line"}, indent=1))
PY
