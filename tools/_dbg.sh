cd $GRAFT_REPO_ROOT
for D in 0 1 2 3 4 7; do
  echo "== dbg $D"
  V2P_ROWS_DBG=$D timeout 300 python3 tools/build_bench.py --workload C3 --samples 2000 --reps 3 --no-exec --kernel 6 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['build_kernels_ms'])" 
done
