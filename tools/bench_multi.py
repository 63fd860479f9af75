"""Microbenchmark of msr3d_gemm_multi_f32 on the schedule's launch groups (M = 960 tokens);
(The A/B switches this tool was written with are gone from the library: single-problem launches take the
panel kernel, mixed launches the tiled one.)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msr3d_amd import _lib

M, D, FF, W, E, KE = 960, 256, 2048, 816, 4096, 768
dev = torch.device("cuda", 0)
lib = _lib.load()
t = lambda *s: torch.randn(*s, device=dev)


def prob(**kw):
    return kw


def dw(dy, n_out, x, k_in, out, db):
    return dict(a_kc=0, b_kc=0, M=n_out, N=k_in, K=M, A=dy, lda=n_out, B=x, ldb=k_in, C=out, ldc=k_in, beta=1.0, colsum=db)


def run(label, probs, iters=100):
    arr = (_lib.GemmProblem * len(probs))()
    keep = []
    flop = 0
    for q, kw in zip(arr, probs):
        for k, v in kw.items():
            if isinstance(v, torch.Tensor):
                keep.append(v); v = v.data_ptr()
            setattr(q, k, v)
        flop += 2.0 * kw["M"] * kw["N"] * kw["K"]
    st = _lib.current_stream_ptr(dev)
    for _ in range(5):
        rc = lib.msr3d_gemm_multi_f32(len(probs), arr, st); assert rc == 0, rc
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.msr3d_gemm_multi_f32(len(probs), arr, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{label:34s} {us:7.1f} us  {flop / us / 1e6:6.1f} TF/s")


h, W2, ffn, b2 = t(M, FF), t(D, FF), torch.zeros(M, D, device=dev), t(D)
run("ffn2 fwd (K=2048)", [prob(a_kc=1, b_kc=1, M=M, N=D, K=FF, A=h, lda=FF, B=W2, ldb=FF, C=ffn, ldc=D, bias=b2, beta=1.0)])
emb, Wp, x0 = t(M, KE), t(D, KE), torch.zeros(M, D, device=dev)
run("proj fwd (K=768)", [prob(a_kc=1, b_kc=1, M=M, N=D, K=KE, A=emb, lda=KE, B=Wp, ldb=KE, C=x0, ldc=D, bias=b2, beta=1.0)])
d_pre, W1, d_t, d_ffn, tt = t(M, FF), t(FF, D), t(M, D), t(M, D), t(M, D)
gW2, gb2, gW1, gb1 = torch.zeros(D, FF, device=dev), torch.zeros(D, device=dev), torch.zeros(FF, D, device=dev), torch.zeros(FF, device=dev)
run("B4: dx1 + dW2 + dW1", [prob(a_kc=1, b_kc=0, M=M, N=D, K=FF, A=d_pre, lda=FF, B=W1, ldb=D, C=d_t, ldc=D, beta=1.0),
                            dw(d_ffn, D, h, FF, gW2, gb2), dw(d_pre, FF, tt, D, gW1, gb1)])
run("   dx1 alone", [prob(a_kc=1, b_kc=0, M=M, N=D, K=FF, A=d_pre, lda=FF, B=W1, ldb=D, C=d_t, ldc=D, beta=1.0)])
run("   dW2 alone", [dw(d_ffn, D, h, FF, gW2, gb2)])
run("   dW1 alone", [dw(d_pre, FF, tt, D, gW1, gb1)])
g, Wl, d_tok, tok = t(M, E), t(E, D), torch.zeros(M, D, device=dev), t(M, D)
gWl, gbl = torch.zeros(E, D, device=dev), torch.zeros(E, device=dev)
run("B6: llm dx + dW", [prob(a_kc=1, b_kc=0, M=M, N=D, K=E, A=g, lda=E, B=Wl, ldb=D, C=d_tok, ldc=D, beta=1.0), dw(g, E, tok, D, gWl, gbl)])
dq, wv, d_xin, xin, d_fc, ctx = t(M, W), t(W, D), t(M, D), t(M, D), t(M, D), t(M, D)
gwv, gbv, gfc, gbfc = torch.zeros(W, D, device=dev), torch.zeros(W, device=dev), torch.zeros(D, D, device=dev), torch.zeros(D, device=dev)
run("B2: dx qkvc + dWqkvc + dWfc", [prob(a_kc=1, b_kc=0, M=M, N=D, K=W, A=dq, lda=W, B=wv, ldb=D, C=d_xin, ldc=D, beta=1.0),
                                    dw(dq, W, xin, D, gwv, gbv), dw(d_fc, D, ctx, D, gfc, gbfc)])
