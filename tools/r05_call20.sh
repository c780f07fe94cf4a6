#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=gpurun_out/r05_call20; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_vcf_to_fasta.py -q -x 2>&1 | tail -3
timeout 600 python tools/decode_bench.py --format min > $OUT/decode_min.json 2> $OUT/decode_min.err; python3 -c "
import json;d=json.load(open('$OUT/decode_min.json'));print({k:d[k] for k in d if k in ('ms','ms_per_pass','kernels_ms','frac','algorithmic_GBps','value')} or list(d)[:40])"
timeout 600 python tools/decode_bench.py --format rich > $OUT/decode_rich.json 2> $OUT/decode_rich.err; tail -c 600 $OUT/decode_rich.json
