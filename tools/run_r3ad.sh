mkdir -p gpurun_out/r3ad
timeout 900 python -m pytest tests/test_gpu_wave_kernel.py tests/test_gpu_whole_cohorts.py -x -q 2>&1 | tail -2
for w in "C2 1000" "C3 2000" "C4 313"; do set -- $w
for i in 1 2; do
timeout 600 python tools/ab.py --workload $1 --samples $2 --rounds 8 "kernel=4" 2>&1 | grep "kernel=4" | sed "s/^/$1 ride /" | tee -a gpurun_out/r3ad/ab.txt
V2P_PHASE_OWN_TOUCH=1 timeout 600 python tools/ab.py --workload $1 --samples $2 --rounds 8 "kernel=4" 2>&1 | grep "kernel=4" | sed "s/^/$1 own  /" | tee -a gpurun_out/r3ad/ab.txt
done; done
