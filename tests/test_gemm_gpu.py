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
]


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


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
