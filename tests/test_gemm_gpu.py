"""GPU: the f32-MFMA token GEMM (msr3d_gemm_f32, all three operand layouts via
hipops.linear forward/backward) against float64 torch.  Exact-f32 fma chains, only the
summation order differs: tolerance rel-L2 <= 1e-5."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [  # (M, K, N)
    (960, 256, 256), (960, 256, 816), (960, 256, 2048), (960, 2048, 256), (960, 256, 4096),
    (960, 768, 256), (960, 63, 256), (960, 3, 256), (61 * 3, 256, 48), (7, 5, 3), (64, 32, 64),
    (65, 33, 67), (1, 256, 256), (130, 84, 256),
    # short reduction, wide output: the A-resident forward kernel, ragged in M and N
    (1000, 144, 520), (70, 16, 600), (129, 256, 1028), (3, 240, 4100),
    # tall and skinny (SharedMLP layers of the unfrozen backbone): forward and dx on the A-resident kernel
    (9000, 64, 128), (8200, 144, 64), (8192, 16, 64),
]


def rel(a, b):
    return float((a.detach().double() - b.detach().double()).norm() / b.detach().double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("M,K,N", SHAPES)
@pytest.mark.parametrize("gelu", [False, True])
def test_linear_fwd_bwd(M, K, N, gelu):
    from msr3d_amd import hipops
    if gelu and N % 4:
        pytest.skip("fused GELU needs N % 4 == 0")
    torch.manual_seed(M * 7 + K * 3 + N)
    x = torch.randn(M, K, device="cuda", requires_grad=True)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).requires_grad_()
    b = torch.randn(N, device="cuda", requires_grad=True)
    gy = torch.randn(M, N, device="cuda")
    y = hipops.linear(x, w, b, gelu=gelu)
    y.backward(gy)
    xd, wd, bd = (t.detach().double().requires_grad_() for t in (x, w, b))
    yd = F.linear(xd, wd, bd)
    if gelu:
        yd = F.gelu(yd)
    yd.backward(gy.double())
    assert rel(y, yd) < 1e-5
    assert rel(x.grad, xd.grad) < 1e-5
    assert rel(w.grad, wd.grad) < 1e-5
    assert rel(b.grad, bd.grad) < 1e-5


def test_linear_batched_leading_dims_and_no_bias():
    from msr3d_amd import hipops
    x = torch.randn(4, 60, 256, device="cuda", requires_grad=True)
    w = torch.randn(128, 256, device="cuda", requires_grad=True)
    y = hipops.linear(x, w)
    assert y.shape == (4, 60, 128)
    y.sum().backward()
    assert rel(y, F.linear(x.double(), w.double())) < 1e-5
    assert x.grad.shape == x.shape and w.grad.shape == w.shape
    # non-contiguous input (a transposed view) is handled
    xt = torch.randn(256, 60, device="cuda").t()
    assert rel(hipops.linear(xt, w), F.linear(xt.double(), w.double())) < 1e-5


def test_split_k_is_bit_reproducible_and_matches_the_atomic_path(monkeypatch):
    """The workspace meeting point adds the K-splits in a fixed order: repeated launches are
    bit-identical (forward, dx, dW, db); the memset + atomicAdd path (no workspace) agrees to
    rounding; counters are left zero."""
    from msr3d_amd import hipops
    monkeypatch.setattr(hipops, "_deterministic", [True])
    torch.manual_seed(0)
    outs = []
    for rep in range(3):
        x = torch.randn(960, 2048, device="cuda").requires_grad_()
        w = torch.randn(256, 2048, device="cuda").requires_grad_()
        b = torch.randn(256, device="cuda").requires_grad_()
        torch.manual_seed(1)
        x.data.normal_(); w.data.normal_(); b.data.normal_()
        y = hipops.linear(x, w, b)
        y.backward(torch.ones_like(y) * 0.5)
        outs.append((y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone()))
    for t0, t1, t2 in zip(*outs):
        assert torch.equal(t0, t1) and torch.equal(t0, t2)
    ws = hipops._workspace(torch.device("cuda", torch.cuda.current_device()))
    assert int(ws[:1024].abs().sum()) == 0
    monkeypatch.setattr(hipops, "_deterministic", [False])
    x = torch.randn(960, 2048, device="cuda").requires_grad_()
    w = torch.randn(256, 2048, device="cuda").requires_grad_()
    b = torch.randn(256, device="cuda").requires_grad_()
    torch.manual_seed(1)
    x.data.normal_(); w.data.normal_(); b.data.normal_()
    y = hipops.linear(x, w, b)
    y.backward(torch.ones_like(y) * 0.5)
    for got, want in zip((y, x.grad, w.grad, b.grad), outs[0]):
        assert rel(got, want) < 1e-6


def test_split_k_with_fused_gelu_and_small_workspace():
    """Raw C-ABI: GELU epilogue after an ordered split-K sum; a workspace too small for the
    preferred split count is used with fewer splits; NULL workspace still works."""
    import ctypes
    from msr3d_amd import _lib
    lib = _lib.load()
    M, N, K = 128, 64, 4096
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda") / K ** 0.5
    bias = torch.randn(N, device="cuda")
    want = F.gelu(F.linear(A.double(), B.double(), bias.double()))
    p = lambda t: ctypes.c_void_p(t.data_ptr())     # noqa: E731
    for ws_bytes in (16 << 20, 4096 + 2 * (64 * 64 + 64) * 4 * 2 + 64, 0):
        ws = torch.zeros(max(ws_bytes, 4) // 4, dtype=torch.int32, device="cuda")
        C = torch.full((M, N), float("nan"), device="cuda")
        pre = torch.empty_like(C)
        rc = lib.msr3d_gemm_f32(1, 1, M, N, K, p(A), K, p(B), K, p(C), N, p(bias), p(pre), 1,
                                ctypes.c_float(0.0), ctypes.c_float(0.0), None, 0,
                                p(ws) if ws_bytes else None,
                                ctypes.c_size_t(ws_bytes), None)
        assert rc == 0
        torch.cuda.synchronize()
        assert rel(C, want) < 1e-5, ws_bytes
        assert rel(F.gelu(pre.double()), want) < 1e-5
        assert int(ws[:1024].abs().sum()) == 0


def test_two_streams_with_separate_lanes_do_not_interfere(monkeypatch):
    from msr3d_amd import hipops
    monkeypatch.setattr(hipops, "_deterministic", [True])
    x = torch.randn(960, 768, device="cuda")
    w = torch.randn(768, 768, device="cuda") / 28
    want = F.linear(x.double(), w.double())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    res = []
    for _ in range(20):
        a = hipops.linear(x, w)
        with torch.cuda.stream(side), hipops.gemm_lane(1):
            b = hipops.linear(x, w)
        res += [a, b]
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for r in res:
        assert torch.equal(r, res[0])
    assert rel(res[0], want) < 1e-5
