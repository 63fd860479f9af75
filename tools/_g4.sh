cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_lora_fp8_gpu.py -m gpu -q -x 2>&1 | tail -5
timeout 300 python tools/prof_fp8_gemm.py 2>&1 | grep "lora=1" | grep "K= 4096\|K=11008\|K= 1024"
