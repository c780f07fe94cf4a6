timeout 900 python -m pytest tests/test_gpu_harness.py -x -q 2>&1 | tail -12
