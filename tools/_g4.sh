cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('every4', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['launches'])"
python bench.py --no-cpu-baseline --steps 12 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('every1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['launches'])"
done
timeout 600 python -m pytest tests/test_bench_ranks_gpu.py -m gpu -q -x 2>&1 | tail -2
