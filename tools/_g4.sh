cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_llama_stack_gpu.py tests/test_full_step_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python bench.py --full-step --llm-fp8 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
timeout 300 python bench.py --full-step --llm-fp8 --no-cpu-baseline --batch 20 2>/dev/null | tail -1 | cut -c1-200
