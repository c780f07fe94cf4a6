mkdir -p gpurun_out/r3ag; export TMPDIR=/tmp
for w in C5 C3; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3ag/$w -o t -- python3 bench.py --workload $w --no-cpu-baseline --no-pcie --no-north-star --verify sample --steps 5 --warmup 2 > gpurun_out/r3ag/$w.log 2>&1
python3 - <<PY
import csv,json
for r in csv.DictReader(open("gpurun_out/r3ag/$w/t_kernel_stats.csv")):
    if 'v2p' in r['Name']: print("$w", r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
for l in open("gpurun_out/r3ag/$w.log"):
    if l.startswith('{'): print(json.loads(l).get('device_image_build'))
PY
done
