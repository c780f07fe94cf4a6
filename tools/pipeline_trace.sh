#!/bin/bash
# Run on the GPU box (gpurun): a kernel + memory-copy trace of the stream-fed pipeline taking C3 whole from Task vectors to host bytes
# (tools/stream_pipeline_probe.py, one configuration) -- how much of the wall time the link and the GPU are busy, and beside each other.
#   bash tools/pipeline_trace.sh   ->  gpurun_out/profiles_r06/r06_pipeline_trace.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pipeline_trace; mkdir -p $OUT $ROOT/gpurun_out/profiles_r06; export TMPDIR=/tmp; cd $ROOT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -o t -- python3 tools/stream_pipeline_probe.py --slice-mb 1152 --slots 4 > $OUT/probe.json 2> $OUT/probe.err
python3 - > $ROOT/gpurun_out/profiles_r06/r06_pipeline_trace.txt <<PY
import csv, glob, json
def load(pat):
    rows = []
    for f in glob.glob(pat):
        rows += list(csv.DictReader(open(f)))
    return rows
k = load("$OUT/t/*kernel_trace.csv"); m = load("$OUT/t/*memory_copy_trace.csv")
def span(rows, a="Start_Timestamp", b="End_Timestamp"):
    iv = sorted((int(r[a]), int(r[b])) for r in rows)
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    if cur_e is not None: busy += cur_e - cur_s
    return busy, (iv[0][0] if iv else 0), (max(e for _, e in iv) if iv else 0)
# the LAST pass of the probe: everything behind the last but one 'digest' burst is hard to cut exactly -- report the whole run and per direction
d2h = [r for r in m if "DEVICE_TO_HOST" in r.get("Direction", "")]; h2d = [r for r in m if "HOST_TO_DEVICE" in r.get("Direction", "")]
big_d2h = [r for r in d2h if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 2_000_000]       # the arenas (> 2 ms each)
print("stream-fed pipeline, C3 whole, 1152 MB slices x 4 slots (tools/stream_pipeline_probe.py: one untimed pass + two timed ones + the reference one call)")
print(open("$OUT/probe.json").read().strip()[:600])
for name, rows in (("kernels", k), ("H2D copies", h2d), ("D2H copies", d2h), ("D2H copies of arenas (> 2 ms)", big_d2h)):
    b, s, e = span(rows)
    print(f"{name:34s} n = {len(rows):6d}   busy {b/1e6:9.1f} ms   first .. last {(e-s)/1e6:9.1f} ms")
if big_d2h:
    b, s, e = span(big_d2h)
    inside_k = [r for r in k if int(r["Start_Timestamp"]) >= s and int(r["End_Timestamp"]) <= e]
    inside_h = [r for r in h2d if int(r["Start_Timestamp"]) >= s and int(r["End_Timestamp"]) <= e]
    print(f"between the first and the last arena copy ({(e-s)/1e6:.1f} ms of wall): arenas on the link {b/1e6:.1f} ms = {100*b/(e-s):.0f} %, kernels busy {span(inside_k)[0]/1e6:.1f} ms, H2D busy {span(inside_h)[0]/1e6:.1f} ms -- all three beside each other")
PY
cat $ROOT/gpurun_out/profiles_r06/r06_pipeline_trace.txt
