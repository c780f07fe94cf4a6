python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for ilp in 1 2; do for x in 0 1; do
python bench.py --no-cpu-baseline --steps 10 --ilp $ilp --xcd-order $x 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('ilp',$ilp,'xcd',$x,'ms',round(j['roofline']['kernel_ms_avg'],3),'min',round(j['roofline']['kernel_ms_min'],3),'frac',round(j['roofline']['frac'],3))"
done; done
