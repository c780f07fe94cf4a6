cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/pmc_parse; mkdir -p $OUT
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$N -o p -- python3 tools/build_bench.py --workload C3 --samples 2000 --reps 1 --no-exec --kernel 6 > $OUT/$N.log 2>&1
done
python3 - <<PY
import csv,glob
for f in sorted(glob.glob("$OUT/*/*/*counter_collection.csv")+glob.glob("$OUT/*/*counter_collection.csv")):
    seen={}
    for r in csv.DictReader(open(f)):
        if 'rows_parse' in r['Kernel_Name']:
            seen.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
    for k,v in seen.items(): print(k, v[-1])
PY
