cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lora_gpu.py tests/test_llama_layer_gpu.py tests/test_llama_stack_gpu.py tests/test_full_step_gpu.py tests/test_lora_fp8_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python bench.py --full-step --llm-fp8 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
bash tools/_g5.sh
