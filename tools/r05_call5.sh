#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call5
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 ) > $OUT/pytest.log 2>&1
tail -16 $OUT/pytest.log
( time timeout 900 python3 bench.py --steps 20 --warmup 5 ) > $OUT/bench_line.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.err
python3 - <<'PY'
import json,os
p=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r05_call5/bench_line.json"
for line in open(p):
    if line.startswith("{"):
        d=json.loads(line)
        print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"].get("box_fill_GBps"), d["roofline"].get("frac_of_box_fill"))
        print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d["one_shot"].items() if k!="what"})
        print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.get("c2_cohort",{}).get("one_shot",{}).items() if k!="what"})
        print(d.get("c2_cohort",{}).get("ms"), d.get("c2_cohort",{}).get("frac"), d.get("host_packed"))
PY
SKIP_CEILING= bash tools/profile_round.sh r05_C3whole > $OUT/profile_C3whole.log 2>&1
tail -5 $OUT/profile_C3whole.log
