mkdir -p gpurun_out/r3aa; export TMPDIR=/tmp
for w in C3 C2; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3aa/$w -o t -- python3 bench.py --workload $w --no-cpu-baseline --no-pcie --no-device-build --no-north-star --verify sample --steps 20 --warmup 3 > gpurun_out/r3aa/$w.log 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/r3aa/$w/t_kernel_stats.csv")):
    if 'v2p' in r['Name']: print("$w", r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
rows=[r for r in csv.DictReader(open("gpurun_out/r3aa/$w/t_kernel_trace.csv")) if 'v2p' in r['Kernel_Name']]
rows=rows[-40:]
t0=int(rows[0]['Start_Timestamp'])
for r in rows[:14]: print(r['Kernel_Name'][5:30], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
PY
done
