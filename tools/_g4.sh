cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_sa_fused_gpu.py tests/test_prompter_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2 3; do
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused', d['value'], d['ms_per_step'])"
MSR3D_FPS_QUERY=0 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two  ', d['value'], d['ms_per_step'])"
MSR3D_ATTN_FWD_WAVES=8 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused+8w', d['value'], d['ms_per_step'])"
done
