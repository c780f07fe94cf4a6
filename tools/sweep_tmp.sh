timeout 900 python -m pytest tests/test_gpu_decode.py -x -q -k "capacity" 2>&1 | tail -12
