timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_vcf_to_fasta.py -x -q 2>&1 | tail -4
V2P_E2E_RUNS=2 python tools/e2e_vcf_bench.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['this_engine']['seconds'], d['this_engine']['decode_kernels_ms'])"
python tools/decode_bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['kernels_ms'])"
