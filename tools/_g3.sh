cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lora_gpu.py tests/test_lora_fp8_gpu.py tests/test_llama_layer_gpu.py -q -x 2>&1 | tail -2
python - <<'P'
import torch, sys
sys.path.insert(0,'.')
from msr3d_amd.llm.lora import _skinny, PAD_R
dev=torch.device('cuda')
for (M,K) in [(2304,4096),(2304,11008),(11520,4096)]:
    x=torch.randn(M,K,device=dev).bfloat16(); a=torch.randn(16,K,device=dev).bfloat16(); u=torch.empty(M,PAD_R,device=dev,dtype=torch.bfloat16)
    f=lambda:_skinny(M,16,K,x,a,u,PAD_R,1.0,dev)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize(); t=e0.elapsed_time(e1)/50*1e3
    print(f"skinny {M}x16x{K}: {t:.1f} us = {M*K*2/t/1e6:.2f} TB/s")
P
python bench.py --full-step --llm-fp8 --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys;j=json.loads(sys.stdin.read());print('full-step fp8',j['value'],j['ms_per_step'])"
