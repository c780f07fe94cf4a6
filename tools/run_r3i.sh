mkdir -p gpurun_out/r3i
run() { timeout 600 python tools/ab.py "$@" 2>&1 | grep "kernel=" | sed "s/^/$TAG /" >> gpurun_out/r3i/len.txt; }
TAG="C3 len400"; run --workload C3 --samples 2000 --rounds 6 "kernel=4" "kernel=0"
TAG="C3 len800"; run --workload C3 --samples 1000 --mean-len 800 --rounds 6 "kernel=4" "kernel=0"
TAG="C3 len1600"; run --workload C3 --samples 500 --mean-len 1600 --rounds 6 "kernel=4" "kernel=0"
TAG="C3 len200"; run --workload C3 --samples 4000 --mean-len 200 --rounds 6 "kernel=4" "kernel=0"
TAG="C2 len400"; run --workload C2 --samples 500 --rounds 6 "kernel=4" "kernel=0"
TAG="C2 len200"; run --workload C2 --samples 1000 --mean-len 200 --rounds 6 "kernel=4" "kernel=0"
TAG="C2 len100"; run --workload C2 --samples 2000 --mean-len 100 --rounds 6 "kernel=4" "kernel=0"
TAG="C2 len800"; run --workload C2 --samples 250 --mean-len 800 --rounds 6 "kernel=4" "kernel=0"
cat gpurun_out/r3i/len.txt
