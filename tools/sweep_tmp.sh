python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py 2>&1 | tail -1
python bench.py --xcd-order 0 --no-cpu-baseline --steps 10 2>&1 | tail -1 | cut -c1-200
