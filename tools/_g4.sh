cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_lora_gpu.py -m gpu -q -x 2>&1 | tail -3
python tools/prof_lora_grad.py 2>&1 | grep pair
timeout 300 python bench.py --full-step --llm-fp8 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
