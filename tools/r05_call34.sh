#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 2400 python tools/routing_sweep.py > gpurun_out/r05_routing_sweep.json 2> gpurun_out/r05_routing_sweep.err; echo rc=$?
python3 - <<'P'
import json,statistics
d=json.load(open('gpurun_out/r05_routing_sweep.json'))
for P in (8,56):
    r=[p['rows_over_best'] for p in d['points'] if p['proteome_MB']==P and p.get('rows_over_best')]
    l=[p['lib_over_best'] for p in d['points'] if p['proteome_MB']==P and p.get('lib_over_best')]
    print('P',P,'rows median',round(statistics.median(r),3),'max',round(max(r),3),'>1.05',sum(1 for x in r if x>1.05),'| lib median',round(statistics.median(l),3),'max',round(max(l),3))
print(d['worst_rows_over_best'], d['worst_rows_point'])
P
