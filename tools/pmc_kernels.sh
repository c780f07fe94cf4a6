#!/bin/bash
# SQ counters of every kernel of tools/kernel_probe.py (three passes), summarised per kernel name: usage pmc_kernels.sh <tag> <probe args...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-k}; shift || true
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/a -o a -- python3 tools/kernel_probe.py "$@" > $OUT/a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/b -o b -- python3 tools/kernel_probe.py "$@" > $OUT/b.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/c -o c -- python3 tools/kernel_probe.py "$@" > $OUT/c.log 2>&1
python3 - <<PY
import csv,glob,json
res={}
for f in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][-48:]
        d=res.setdefault(n,{}); d.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
        d.setdefault('_dur_'+r['Counter_Name'],[]).append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
out={}
for n,d in res.items():
    o={k:v[-1] for k,v in d.items() if not k.startswith('_')}
    durs=[v[-1] for k,v in d.items() if k.startswith('_dur_')]
    o['duration_us']=min(durs)/1e3
    out[n]=o
json.dump(out,open("$OUT/summary.json","w"),indent=1)
for n,o in sorted(out.items(), key=lambda kv:-kv[1]['duration_us'])[:8]:
    print(n, {k:(round(v,1) if isinstance(v,float) else v) for k,v in o.items()})
PY
