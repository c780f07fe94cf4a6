cd $GRAFT_REPO_ROOT
for W in "C3 2000" "C2 500" "C4 313"; do set -- $W
  timeout 600 python3 tools/build_bench.py --workload $1 --samples $2 --reps 2 --check --kernel 6 --exec-reps 9 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['workload'], d['build_kernels_ms'], 'host', d['ab_execute_ms_host_packed'], 'dev', d['ab_execute_ms_device_built'], d['chunks'], d['host_chunks'])"
done
