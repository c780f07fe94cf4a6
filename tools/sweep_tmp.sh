timeout 900 python -m pytest tests/test_random_kats.py -x -q 2>&1 | tail -12
