mkdir -p gpurun_out/r3af; export TMPDIR=/tmp
pr() { python3 -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        j=json.loads(l); print('$2', 'ms_per_step', round(j['ms_per_step'],3), 'kernel_ms_avg', round(j['roofline']['kernel_ms_avg'],3), 'min', round(j['roofline']['kernel_ms_min'],3))
"; }
F="--no-cpu-baseline --no-pcie --no-device-build --no-north-star --no-speedup-ref --verify sample"
python3 bench.py $F > gpurun_out/r3af/a.log 2>&1; pr gpurun_out/r3af/a.log "flags, no profiler"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3af/t -o t -- python3 bench.py $F > gpurun_out/r3af/b.log 2>&1; pr gpurun_out/r3af/b.log "flags, rocprofv3 kernel-trace"
python3 bench.py $F > gpurun_out/r3af/c.log 2>&1; pr gpurun_out/r3af/c.log "flags, no profiler (again)"
python3 bench.py --no-north-star > gpurun_out/r3af/d.log 2>&1; pr gpurun_out/r3af/d.log "default minus north star"
python3 bench.py --no-cpu-baseline --no-north-star > gpurun_out/r3af/e.log 2>&1; pr gpurun_out/r3af/e.log "no cpu baseline, no north star"
python3 bench.py --no-cpu-baseline --no-north-star --no-pcie > gpurun_out/r3af/f.log 2>&1; pr gpurun_out/r3af/f.log "no cpu baseline, no north star, no pcie"
