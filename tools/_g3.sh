cd $GRAFT_REPO_ROOT
for f in "" "--llm-fp8"; do
python bench.py --llm-layer $f --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys;j=json.loads(sys.stdin.read());print('llm-layer $f ms',j['ms_per_step'],j['tflops'])"
python bench.py --full-step $f --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys;j=json.loads(sys.stdin.read());print('full-step $f',j['value'],j['ms_per_step'],j['loss'],j['hbm_allocated_gb'])"
done
python - <<'P'
import torch, sys, time
sys.path.insert(0,'.')
from msr3d_amd.llm.lora import quant_rows_fp8, _gemm_fp8, _gemm, PAD_R
dev=torch.device('cuda')
for (M,N,K) in [(2304,4096,4096),(2304,11008,4096),(2304,4096,11008),(11520,4096,4096)]:
    x=torch.randn(M,K,device=dev).bfloat16(); w=(torch.randn(N,K,device=dev)/K**.5).bfloat16()
    u=torch.zeros(M,PAD_R,device=dev).bfloat16(); b2=torch.zeros(N,PAD_R,device=dev).bfloat16()
    xq,sx=quant_rows_fp8(x); wq,sw=quant_rows_fp8(w); y=torch.empty(M,N,device=dev,dtype=torch.bfloat16)
    def t(f,n=20):
        for _ in range(3): f()
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
    tf=t(lambda:_gemm_fp8(M,N,K,xq,sx,wq,sw,u,b2,y,dev)); tb=t(lambda:_gemm(M,N,K,PAD_R,x,K,w,K,u,PAD_R,b2,PAD_R,y,N,False,1.0,dev)); tq=t(lambda:quant_rows_fp8(x))
    fl=2.0*M*N*K
    print(f"{M}x{N}x{K}: fp8 {tf:.1f} us = {fl/tf/1e6:.0f} TF | bf16 {tb:.1f} us = {fl/tb/1e6:.0f} TF | quant rows {tq:.1f} us")
P
