"""Microbenchmark of msr3d_strip_gemm_f32 variants at the bench shape (M = 960): HIP-event time per
launch (back-to-back launches, so the per-launch floor is the in-stream one, ~2.5 us)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msr3d_amd import _lib, hipops

M, D = 960, 256
dev = torch.device("cuda", 0)
lib = _lib.load()
seed = hipops.seed_word(dev)


def run(label, iters=200, **kw):
    s = _lib.StripGemm()
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v); v = v.data_ptr()
        setattr(s, k, v if v is not None else 0)
    st = _lib.current_stream_ptr(dev)
    for _ in range(10):
        rc = lib.msr3d_strip_gemm_f32(ctypes.byref(s), st); assert rc == 0, rc
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.msr3d_strip_gemm_f32(ctypes.byref(s), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    flop = 2.0 * M * kw["N"] * D
    print(f"{label:44s} {us:7.1f} us  {flop / us / 1e6:6.1f} TF/s")


t = lambda *s: torch.randn(*s, device=dev)
a0, a1, a2 = t(M, D), t(M, D), t(M, D)
g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
o0, o1, o2 = t(M, D), t(M, D), t(M, D)
s1, s2 = torch.rand(M, 2, device=dev) + 0.5, torch.rand(M, 2, device=dev) + 0.5
for N in (256, 816, 2048, 4096):
    W, bias = t(N, D) / 16, t(N)
    C, Cpre = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    for ng in ((0,) if N < 2048 else (1, 2, 4)):
        run(f"kc plain bias N={N} ng={ng}", M=M, N=N, pro=0, epi=0, b_kc=1, groups_per_wg=ng, a0=a0, W=W, ldw=D, bias=bias, C=C, ldc=N)
    run(f"kc ln bias N={N}", M=M, N=N, pro=2, epi=0, b_kc=1, a0=a0, a1=a1, a2=a2, g1=g, b1=b, eps1=1e-5, p1=0.1, salt1=3, seed=seed,
        o0=o0, ost1=s1, o1=o1, W=W, ldw=D, bias=bias, C=C, ldc=N)
    if N == 2048:
        for ng in (1, 2):
            run(f"kc ln2(p=0) bias ng={ng}", M=M, N=N, pro=3, epi=0, b_kc=1, groups_per_wg=ng, a0=a0, a1=a1, g1=g, b1=b, eps1=1e-5, g2=g, b2=b, eps2=1e-5,
                o0=o0, ost1=s1, o2=o2, ost2=s2, o1=o1, W=W, ldw=D, bias=bias, C=C, ldc=N)
            run(f"kc ln2(p=.1) bias ng={ng}", M=M, N=N, pro=3, epi=0, b_kc=1, groups_per_wg=ng, a0=a0, a1=a1, g1=g, b1=b, eps1=1e-5, g2=g, b2=b, eps2=1e-5,
                p1=0.1, salt1=1, p2=0.1, salt2=2, seed=seed, o0=o0, ost1=s1, o2=o2, ost2=s2, o1=o1, W=W, ldw=D, bias=bias, C=C, ldc=N)
            run(f"kc plain gelu(p=0) ng={ng}", M=M, N=N, pro=0, epi=1, b_kc=1, groups_per_wg=ng, a0=a0, W=W, ldw=D, bias=bias, C=C, ldc=N, Cpre=Cpre)
            run(f"kc plain gelu(p=.1) ng={ng}", M=M, N=N, pro=0, epi=1, b_kc=1, groups_per_wg=ng, a0=a0, W=W, ldw=D, bias=bias, C=C, ldc=N, Cpre=Cpre,
                p_drop=0.1, salt=9, seed=seed)
            run(f"kc ln2(p=.1) gelu(p=.1) ng={ng}  [K4]", M=M, N=N, pro=3, epi=1, b_kc=1, groups_per_wg=ng, a0=a0, a1=a1, g1=g, b1=b, eps1=1e-5, g2=g, b2=b,
                eps2=1e-5, p1=0.1, salt1=1, p2=0.1, salt2=2, seed=seed, o0=o0, ost1=s1, o2=o2, ost2=s2, o1=o1, W=W, ldw=D, bias=bias,
                C=C, ldc=N, Cpre=Cpre, p_drop=0.1, salt=9)
# backward products
for N in (256, 2048):
    W = t(D, N) / 16
    C, pre = torch.empty(M, N, device=dev), t(M, N)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    run(f"kr plain N={N}", M=M, N=N, pro=0, epi=0, b_kc=0, a0=a0, W=W, ldw=N, C=C, ldc=N)
    run(f"kr lnbwd N={N}", M=M, N=N, pro=4, epi=0, b_kc=0, a0=a0, a1=a1, st1=s1, g1=g, p1=0.1, salt1=2, seed=seed, o0=o0, o1=o1,
        dg1=dg, db1=db, W=W, ldw=N, C=C, ldc=N)
    run(f"kr lnbwd gelubwd(p=.1) N={N}  [B5]", M=M, N=N, pro=4, epi=2, b_kc=0, a0=a0, a1=a1, st1=s1, g1=g, p1=0.1, salt1=2, seed=seed,
        o0=o0, o1=o1, dg1=dg, db1=db, W=W, ldw=N, C=C, ldc=N, pre_in=pre, p_drop=0.1, salt=9)
    run(f"kr ln2bwd N={N}  [B3a]", M=M, N=N, pro=5, epi=0, b_kc=0, a0=a0, a1=a1, a2=a2, st1=s1, st2=s2, g1=g, g2=g, p1=0.1, salt1=1,
        p2=0.1, salt2=2, seed=seed, o0=o0, o1=o1, dg1=dg, db1=db, dg2=dg, db2=db, W=W, ldw=N, C=C, ldc=N)
