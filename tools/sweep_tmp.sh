timeout 900 python -m pytest tests/test_gpu_vcf_to_fasta.py -x -q 2>&1 | tail -30
