"""msr3d_strip_gemm_f32 (csrc/strip_gemm.hip): every prologue / epilogue against float64 torch
evaluations of the reference's formulation (transformers.py:250-251,324-328: dropout + residual +
LayerNorm chains in front of a Linear, and autograd's backward of them), at the path's shapes and at
ragged ones (M not a multiple of 64, N not a multiple of 64).  Dropout: the kernels' own masks are
recovered from a p-only run (the mask is a pure function of seed / salt / index), then the dropped
formulation is checked exactly like the dropout-free one.  Tolerance 2e-5 rel-L2 (fp32 products on
f32-input MFMA; measured ~1e-6)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
D = 256


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def call(**kw):
    from msr3d_amd import _lib
    s = _lib.StripGemm()
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            v = v.data_ptr()
        setattr(s, k, v if v is not None else 0)
    lib = _lib.load()
    rc = lib.msr3d_strip_gemm_f32(ctypes.byref(s), _lib.current_stream_ptr(torch.device("cuda")))
    _lib.check(rc, "msr3d_strip_gemm_f32")
    torch.cuda.synchronize()


def seed_word():
    from msr3d_amd import hipops
    return hipops.seed_word(torch.device("cuda", torch.cuda.current_device()))


def mask_of(M, p, salt):
    """The keep-mask (M, 256) the row kernels draw for (p, salt): run dropout on ones through
    msr3d_dropout_add_ln_fwd's saved pre-norm sum."""
    from msr3d_amd import _lib
    ones = torch.ones(M, D, device="cuda")
    y, s, st = torch.empty_like(ones), torch.empty_like(ones), torch.empty(M, 2, device="cuda")
    g = torch.ones(D, device="cuda")
    lib = _lib.load()
    vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)   # noqa: E731
    rc = lib.msr3d_dropout_add_ln_fwd(M, D, vp(ones), vp(None), vp(g), vp(g), ctypes.c_float(1e-5),
                                      ctypes.c_float(p), vp(seed_word()), salt, vp(y), vp(s), vp(st),
                                      _lib.current_stream_ptr(torch.device("cuda")))
    _lib.check(rc, "mask probe")
    torch.cuda.synchronize()
    return (s > 0).double() / (1.0 - p)


def ln64(v, g, b, eps=1e-5):
    return F.layer_norm(v, (D,), g.double(), b.double(), eps)


@pytest.mark.parametrize("M,N", [(960, 816), (130, 256), (64, 64), (977, 200)])
def test_plain_and_add_prologues(M, N):
    torch.manual_seed(M + N)
    a0, a1 = torch.randn(M, D, device="cuda"), torch.randn(M, D, device="cuda")
    v1, v2 = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
    W, bias = torch.randn(N, D, device="cuda") / 16, torch.randn(N, device="cuda")
    C = torch.full((M, N), float("nan"), device="cuda")
    call(M=M, N=N, pro=0, epi=0, b_kc=1, a0=a0, W=W, ldw=D, bias=bias, C=C, ldc=N)
    assert rel(C, a0.double() @ W.double().T + bias.double()) < 2e-5
    o1 = torch.empty(M, D, device="cuda")
    C.fill_(float("nan"))
    call(M=M, N=N, pro=1, epi=0, b_kc=1, a0=a0, a1=a1, g1=v1, b1=v2, o1=o1, W=W, ldw=D, bias=bias, C=C, ldc=N)
    x = a0.double() + a1.double() + v1.double() + v2.double()
    assert rel(o1, x) < 1e-6
    assert rel(C, x @ W.double().T + bias.double()) < 2e-5


@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("M,N,with_pos", [(960, 816, True), (960, 4096, False), (100, 128, True)])
def test_ln_prologue(M, N, with_pos, p):
    torch.manual_seed(N)
    a0, a1, a2 = (torch.randn(M, D, device="cuda") for _ in range(3))
    g, b = torch.rand(D, device="cuda") + 0.5, torch.randn(D, device="cuda")
    W, bias = torch.randn(N, D, device="cuda") / 16, torch.randn(N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    o0, o1, st = torch.empty(M, D, device="cuda"), torch.empty(M, D, device="cuda"), torch.empty(M, 2, device="cuda")
    call(M=M, N=N, pro=2, epi=0, b_kc=1, a0=a0, a1=a1, a2=a2 if with_pos else None, g1=g, b1=b, eps1=1e-5, p1=p,
         salt1=77, seed=seed_word(), o0=o0, ost1=st, o1=o1, W=W, ldw=D, bias=bias, C=C, ldc=N)
    keep = mask_of(M, p, 77) if p > 0 else 1.0
    v = a0.double() * keep + a1.double()
    y = ln64(v, g, b) + (a2.double() if with_pos else 0)
    assert rel(o0, v) < 1e-6 and rel(o1, y) < 1e-5
    assert rel(st[:, 0], v.mean(1)) < 1e-4
    assert rel(st[:, 1], 1 / torch.sqrt(v.var(1, unbiased=False) + 1e-5)) < 1e-5
    assert rel(C, y @ W.double().T + bias.double()) < 2e-5


@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("M,N", [(960, 2048), (70, 192)])
def test_ln2_prologue_gelu_epilogue(M, N, p):
    torch.manual_seed(3)
    a0, a1 = torch.randn(M, D, device="cuda"), torch.randn(M, D, device="cuda")
    g1, b1, g2, b2 = (torch.rand(D, device="cuda") + 0.5 for _ in range(4))
    W, bias = torch.randn(N, D, device="cuda") / 16, torch.randn(N, device="cuda")
    C, Cpre = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    o0, o1, o2 = (torch.empty(M, D, device="cuda") for _ in range(3))
    s1, s2 = torch.empty(M, 2, device="cuda"), torch.empty(M, 2, device="cuda")
    call(M=M, N=N, pro=3, epi=1, b_kc=1, a0=a0, a1=a1, g1=g1, b1=b1, eps1=1e-5, p1=p, salt1=5, g2=g2, b2=b2,
         eps2=1e-5, p2=p, salt2=6, seed=seed_word(), o0=o0, ost1=s1, o2=o2, ost2=s2, o1=o1, W=W, ldw=D, bias=bias,
         C=C, ldc=N, Cpre=Cpre, p_drop=0.0, salt=0)
    k1 = mask_of(M, p, 5) if p > 0 else 1.0
    k2 = mask_of(M, p, 6) if p > 0 else 1.0
    v1 = a0.double() * k1 + a1.double()
    v2 = ln64(v1, g1, b1) * k2 + a1.double()
    t = ln64(v2, g2, b2)
    pre = t @ W.double().T + bias.double()
    assert rel(o0, v1) < 1e-6 and rel(o2, v2) < 1e-5 and rel(o1, t) < 1e-5
    assert rel(Cpre, pre) < 2e-5 and rel(C, F.gelu(pre)) < 2e-5
    if p > 0:        # the FFN dropout of the epilogue: each element either 0 or gelu / (1 - p)
        call(M=M, N=N, pro=3, epi=1, b_kc=1, a0=a0, a1=a1, g1=g1, b1=b1, eps1=1e-5, p1=p, salt1=5, g2=g2, b2=b2,
             eps2=1e-5, p2=p, salt2=6, seed=seed_word(), o0=o0, ost1=s1, o2=o2, ost2=s2, o1=o1, W=W, ldw=D,
             bias=bias, C=C, ldc=N, Cpre=Cpre, p_drop=p, salt=9)
        full = F.gelu(pre) / (1 - p)
        dropped = C == 0
        assert 0.5 * p < dropped.double().mean() < 1.5 * p
        assert rel(C[~dropped], full[~dropped]) < 2e-5


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_lnbwd_prologue_gelubwd_epilogue_matches_autograd(p):
    """d_pre = d(loss)/d(pre) for out = LN(drop(h W2^T...)): here the pieces the kernel covers --
    LN backward, its dropout, dx = d_ffn W2 and the GELU (+ FFN dropout) backward."""
    M, N = 200, 256            # N = FFN width of the test
    torch.manual_seed(11)
    pre = torch.randn(M, N, device="cuda", dtype=torch.float64, requires_grad=True)
    W2 = (torch.randn(D, N, device="cuda") / 16)
    t = torch.randn(M, D, device="cuda", dtype=torch.float64, requires_grad=True)
    g, b = torch.rand(D, device="cuda") + 0.5, torch.randn(D, device="cuda")
    k_ffn = mask_of_wide(M, N, p, 21) if p > 0 else 1.0
    k_2 = mask_of(M, p, 22) if p > 0 else 1.0
    h = F.gelu(pre) * k_ffn
    ffn = h @ W2.double().T
    v = ffn * k_2 + t
    out = ln64(v, g, b)
    dy = torch.randn(M, D, device="cuda")
    out.backward(dy.double())
    mean, rstd = v.mean(1), 1 / torch.sqrt(v.var(1, unbiased=False) + 1e-5)
    st = torch.stack([mean, rstd], 1).float().contiguous()
    o0, o1 = torch.empty(M, D, device="cuda"), torch.empty(M, D, device="cuda")
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    C = torch.empty(M, N, device="cuda")
    call(M=M, N=N, pro=4, epi=2, b_kc=0, a0=dy, a1=v.detach().float().contiguous(), st1=st, g1=g, p1=p, salt1=22,
         seed=seed_word(), o0=o0, o1=o1, dg1=dg, db1=db, W=W2, ldw=N, C=C, ldc=N,
         pre_in=pre.detach().float().contiguous(), p_drop=p, salt=21)
    assert rel(o1, t.grad) < 2e-5                                  # residual gradient = LN-bwd
    assert rel(C, pre.grad) < 3e-5
    xh = (v.detach() - mean[:, None]) * rstd[:, None]
    assert rel(dg, (dy.double() * xh).sum(0)) < 2e-5 and rel(db, dy.double().sum(0)) < 2e-5


def mask_of_wide(M, N, p, salt):
    """Keep-mask of the GEMM-epilogue dropout (index row * N + col): from a GELU-epilogue run on an
    operand that makes every pre-activation positive."""
    a0 = torch.ones(M, D, device="cuda")
    W = torch.ones(N, D, device="cuda") / D
    C = torch.empty(M, N, device="cuda")
    call(M=M, N=N, pro=0, epi=1, b_kc=1, a0=a0, W=W, ldw=D, C=C, ldc=N, p_drop=p, salt=salt, seed=seed_word())
    return (C > 0).double() / (1.0 - p)


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_ln2bwd_prologue_matches_autograd(p):
    M = 333
    torch.manual_seed(12)
    fc = torch.randn(M, D, device="cuda", dtype=torch.float64, requires_grad=True)
    x = torch.randn(M, D, device="cuda", dtype=torch.float64, requires_grad=True)
    g1, b1, g2, b2 = (torch.rand(D, device="cuda") + 0.5 for _ in range(4))
    Wfc = torch.randn(D, D, device="cuda") / 16
    k1 = mask_of(M, p, 31) if p > 0 else 1.0
    k2 = mask_of(M, p, 32) if p > 0 else 1.0
    v1 = fc * k1 + x
    v2 = ln64(v1, g1, b1) * k2 + x
    t = ln64(v2, g2, b2)
    dt = torch.randn(M, D, device="cuda")
    t.backward(dt.double())
    st = lambda v: torch.stack([v.mean(1), 1 / torch.sqrt(v.var(1, unbiased=False) + 1e-5)], 1).float().contiguous()  # noqa: E731
    o0, o1 = torch.empty(M, D, device="cuda"), torch.empty(M, D, device="cuda")
    grads = [torch.zeros(D, device="cuda") for _ in range(4)]
    C = torch.empty(M, D, device="cuda")
    call(M=M, N=D, pro=5, epi=0, b_kc=0, a0=dt, a1=v1.detach().float().contiguous(), a2=v2.detach().float().contiguous(),
         st1=st(v1.detach()), st2=st(v2.detach()), g1=g1, g2=g2, p1=p, salt1=31, p2=p, salt2=32, seed=seed_word(),
         o0=o0, o1=o1, dg1=grads[0], db1=grads[1], dg2=grads[2], db2=grads[3], W=Wfc, ldw=D, C=C, ldc=D)
    assert rel(o0, fc.grad) < 2e-5 and rel(o1, x.grad) < 2e-5
    assert rel(C, fc.grad @ Wfc.double()) < 3e-5                   # d_ctx = d_fc Wfc
    xh2 = (v2.detach() - v2.detach().mean(1, keepdim=True)) * st(v2.detach())[:, 1:2].double()
    assert rel(grads[2], (dt.double() * xh2).sum(0)) < 2e-5 and rel(grads[3], dt.double().sum(0)) < 2e-5


def test_gemm_multi_three_layouts_in_one_launch():
    from msr3d_amd import _lib
    torch.manual_seed(4)
    M, N, K = 960, 256, 2048
    dy = torch.randn(M, K, device="cuda")
    W = torch.randn(K, N, device="cuda") / 32
    x = torch.randn(M, N, device="cuda")
    dx = torch.randn(M, N, device="cuda")
    dx0 = dx.clone()
    dW, db = torch.zeros(K, N, device="cuda"), torch.zeros(K, device="cuda")
    y = torch.zeros(M, N, device="cuda")
    Wf, bf = torch.randn(N, K, device="cuda") / 32, torch.randn(N, device="cuda")
    arr = (_lib.GemmProblem * 3)()
    def fill(q, **kw):
        for k, v in kw.items():
            setattr(q, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    fill(arr[0], a_kc=1, b_kc=0, M=M, N=N, K=K, A=dy, lda=K, B=W, ldb=N, C=dx, ldc=N, beta=1.0)
    fill(arr[1], a_kc=0, b_kc=0, M=K, N=N, K=M, A=dy, lda=K, B=x, ldb=N, C=dW, ldc=N, beta=1.0, colsum=db)
    fill(arr[2], a_kc=1, b_kc=1, M=M, N=N, K=K, A=dy, lda=K, B=Wf, ldb=K, C=y, ldc=N, bias=bf, beta=1.0)
    rc = _lib.load().msr3d_gemm_multi_f32(3, arr, _lib.current_stream_ptr(torch.device("cuda")))
    _lib.check(rc, "msr3d_gemm_multi_f32")
    torch.cuda.synchronize()
    assert rel(dx, dx0.double() + dy.double() @ W.double()) < 2e-5
    assert rel(dW, dy.double().T @ x.double()) < 2e-5 and rel(db, dy.double().sum(0)) < 2e-5
    assert rel(y, dy.double() @ Wf.double().T + bf.double()) < 2e-5


def test_pos_embed_and_scene_prologue_match_the_module_formulation():
    import msr3d_amd.model  # noqa: F401
    from msr3d_amd import _lib, hipops
    torch.manual_seed(8)
    B, L, KF = 5, 37, 63
    M = B * L
    loc = torch.randn(B, L, 6, device="cuda")
    loc[..., 3:] = loc[..., 3:].abs()
    valid = torch.rand(B, L, device="cuda") > 0.3
    al = torch.randn(B, 3, device="cuda")
    ao = F.normalize(torch.randn(B, 4, device="cuda"), dim=-1)
    freqs = torch.linspace(1.0, 15, steps=10, device="cuda")
    pw, ff = torch.empty(B, L, L, 5, device="cuda"), torch.empty(B, L, KF, device="cuda")
    loc6, pad = torch.empty(B, L, 6, device="cuda"), torch.empty(B, L, dtype=torch.uint8, device="cuda")
    vout = torch.empty(B, L, dtype=torch.uint8, device="cuda")
    al2, ao2 = torch.empty_like(al), torch.empty_like(ao)
    vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)   # noqa: E731
    lib, st = _lib.load(), _lib.current_stream_ptr(torch.device("cuda"))
    rc = lib.msr3d_scene_prologue(B, L, vp(loc), vp(valid.view(torch.uint8)), vp(al), vp(ao), vp(freqs), 10, 1,
                                  ctypes.c_float(1e-10), vp(pw), vp(ff), vp(loc6), vp(pad), vp(vout), vp(al2), vp(ao2),
                                  st)
    _lib.check(rc, "msr3d_scene_prologue")
    assert torch.equal(al2, al) and torch.equal(ao2, ao)
    assert torch.equal(pw, hipops.pairwise_locs_center5(loc))
    assert torch.equal(ff, hipops.agent_fourier(loc, al, ao))
    assert torch.equal(loc6, loc) and torch.equal(pad.bool(), ~valid) and torch.equal(vout.bool(), valid)

    enc_a = torch.nn.Sequential(torch.nn.Linear(KF, D), torch.nn.LayerNorm(D)).cuda().double()
    enc_b = torch.nn.Sequential(torch.nn.Linear(3, D), torch.nn.LayerNorm(D)).cuda().double()
    f32 = lambda t: t.detach().float().contiguous()   # noqa: E731
    wa, ba, ga, bta = (f32(t) for t in (enc_a[0].weight, enc_a[0].bias, enc_a[1].weight, enc_a[1].bias))
    wb, bb, gb, btb = (f32(t) for t in (enc_b[0].weight, enc_b[0].bias, enc_b[1].weight, enc_b[1].bias))
    pos, sa, sb = (torch.empty(M, D, device="cuda") for _ in range(3))
    sta, stb = torch.empty(M, 2, device="cuda"), torch.empty(M, 2, device="cuda")
    rc = lib.msr3d_pos_embed_fwd(M, KF, vp(ff), vp(loc6), vp(wa), vp(ba), vp(ga), vp(bta), ctypes.c_float(1e-5),
                                 vp(wb), vp(bb), vp(gb), vp(btb), ctypes.c_float(1e-5), vp(pos), vp(sa), vp(sta),
                                 vp(sb), vp(stb), st)
    _lib.check(rc, "msr3d_pos_embed_fwd")
    want = enc_a(ff.double().view(M, KF)) + enc_b(loc6.double().view(M, 6)[:, 3:])
    assert rel(pos, want) < 1e-5
    # backward: three upstream gradients, LN backwards, parameter gradients, column sums
    d = [torch.randn(M, D, device="cuda") for _ in range(3)]
    want.backward((d[0] + d[1] + d[2]).double())
    da, db = torch.empty(M, D, device="cuda"), torch.empty(M, D, device="cuda")
    acc = [torch.zeros(D, device="cuda") for _ in range(6)]
    rc = lib.msr3d_pos_embed_bwd(M, vp(d[0]), vp(d[1]), vp(d[2]), vp(sa), vp(sta), vp(ga), vp(sb),
                                 vp(stb), vp(gb), vp(da), vp(db), *[vp(t) for t in acc], st)
    _lib.check(rc, "msr3d_pos_embed_bwd")
    torch.cuda.synchronize()
    assert rel(acc[0], enc_a[1].weight.grad) < 2e-5 and rel(acc[1], enc_a[1].bias.grad) < 2e-5
    assert rel(acc[2], enc_b[1].weight.grad) < 2e-5 and rel(acc[3], enc_b[1].bias.grad) < 2e-5
    assert rel(acc[4], d[0].double().sum(0)) < 2e-5 and rel(acc[5], acc[4]) < 1e-5
    # d_lin: check through the weight gradients they imply
    assert rel(da.double().T @ ff.double().view(M, KF), enc_a[0].weight.grad) < 2e-5
    assert rel(db.double().sum(0), enc_b[0].bias.grad) < 2e-5


def test_gemm_multi_ragged_problems_panel_and_tiled():
    """The schedule's other launch groups, including the ragged ones (816 output rows / 816-long
    reduction, a 64-long tail chunk at K = 960, N = 63 / 3 weight gradients that stay on the tiled
    kernel), through both kernels behind msr3d_gemm_multi_f32 (subprocess-free: the switch is read
    once per process, so the tiled variant is exercised through problems the panel kernel rejects)."""
    from msr3d_amd import _lib
    torch.manual_seed(14)
    M, D, W = 960, 256, 816
    dq, wv, xin = torch.randn(M, W, device="cuda"), torch.randn(W, D, device="cuda") / 16, torch.randn(M, D, device="cuda")
    d_x = torch.randn(M, D, device="cuda")
    d_x0 = d_x.clone()
    gw, gb = torch.zeros(W, D, device="cuda"), torch.zeros(W, device="cuda")
    ff = torch.randn(M, 63, device="cuda")
    d_la = torch.randn(M, D, device="cuda")
    gl, gbl = torch.zeros(D, 63, device="cuda"), torch.zeros(D, device="cuda")
    arr = (_lib.GemmProblem * 3)()
    def fill(q, **kw):
        for k, v in kw.items():
            setattr(q, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    fill(arr[0], a_kc=1, b_kc=0, M=M, N=D, K=W, A=dq, lda=W, B=wv, ldb=D, C=d_x, ldc=D, beta=1.0)
    fill(arr[1], a_kc=0, b_kc=0, M=W, N=D, K=M, A=dq, lda=W, B=xin, ldb=D, C=gw, ldc=D, beta=1.0, colsum=gb)
    fill(arr[2], a_kc=0, b_kc=0, M=D, N=63, K=M, A=d_la, lda=D, B=ff, ldb=63, C=gl, ldc=63, beta=1.0, colsum=gbl)
    rc = _lib.load().msr3d_gemm_multi_f32(3, arr, _lib.current_stream_ptr(torch.device("cuda")))
    _lib.check(rc, "msr3d_gemm_multi_f32")
    torch.cuda.synchronize()
    assert rel(d_x, d_x0.double() + dq.double() @ wv.double()) < 2e-5
    assert rel(gw, dq.double().T @ xin.double()) < 2e-5 and rel(gb, dq.double().sum(0)) < 2e-5
    assert rel(gl, d_la.double().T @ ff.double()) < 2e-5 and rel(gbl, d_la.double().sum(0)) < 2e-5
    # beta = 0 with K-splits (the planner clears C itself) and a forward product with bias, K = 768
    emb, Wp, bp = torch.randn(M, 768, device="cuda"), torch.randn(D, 768, device="cuda") / 28, torch.randn(D, device="cuda")
    y = torch.full((M, D), float("nan"), device="cuda")
    one = (_lib.GemmProblem * 1)()
    fill(one[0], a_kc=1, b_kc=1, M=M, N=D, K=768, A=emb, lda=768, B=Wp, ldb=768, C=y, ldc=D, bias=bp, beta=0.0)
    rc = _lib.load().msr3d_gemm_multi_f32(1, one, _lib.current_stream_ptr(torch.device("cuda")))
    _lib.check(rc, "msr3d_gemm_multi_f32")
    assert rel(y, emb.double() @ Wp.double().T + bp.double()) < 2e-5
