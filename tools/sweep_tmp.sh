timeout 900 python -m pytest tests -m gpu -x -q --durations=5 2>&1 | tail -12
