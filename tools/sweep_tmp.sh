timeout 900 python -m pytest tests/test_cohort_golden.py -x -q -m gpu --durations=3 2>&1 | tail -12
