timeout 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_vcf_to_fasta.py -x -q 2>&1 | tail -3
for rep in 1 2; do for v in cur head; do timeout 200 python tools/decode_bench.py --no-cpu-baseline --steps 10 --lib build_ab/$v.so 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernels_ms'].items()})"; done; done
bash tools/pmc_decode_sq.sh > gpurun_out/pmc_decode_sq.txt 2>&1
