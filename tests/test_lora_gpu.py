"""LoRALinear (msr3d_amd/llm/lora.py) against float64 evaluations of peft's formulation
y = x W^T + s (x A^T) B^T on the SAME bf16-rounded operands (the kernel's arithmetic: bf16 MFMA, fp32
accumulate; the low-rank intermediates u = s x A^T and v = s dy B are held in bf16, which the
reference evaluation mirrors), forward, dx, dA and dB; Vicuna-7B projection shapes and ragged token
counts; plus a roofline line for the 4096 x 4096 projection."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def bf(t):
    return t.to(torch.bfloat16).double()


@pytest.mark.parametrize("M,K,N,r", [(2304, 4096, 4096, 16), (300, 4096, 11008, 16), (129, 11008, 4096, 16),
                                     (64, 256, 192, 32)])
def test_lora_linear_matches_the_float64_formulation(M, K, N, r):
    from msr3d_amd.llm import LoRALinear
    torch.manual_seed(M + N)
    lin = LoRALinear(K, N, r=r, lora_alpha=16, device="cuda")
    lin.load_base_weight(torch.randn(N, K, device="cuda") / K ** 0.5)
    with torch.no_grad():
        lin.lora_B.weight.copy_(torch.randn(N, r, device="cuda") * 0.05)     # (peft starts B at zero)
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16).requires_grad_(True)
    y = lin(x)
    assert y.dtype == torch.bfloat16 and y.shape == (M, N)
    s = lin.scaling
    W, A, B = lin.weight.double(), bf(lin.lora_A.weight), bf(lin.lora_B.weight)
    xd = x.detach().double()
    u = bf((s * (xd @ A.T)).float())                 # the bf16 intermediate
    want = xd @ W.T + u @ B.T
    assert rel(y.float(), want) < 2 ** -8            # output rounding to bf16 (2^-9 per element) dominates
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    y.backward(dy)
    dyd = dy.double()
    v = bf((s * (dyd @ B)).float())
    assert rel(x.grad.float(), dyd @ W + v @ A) < 2 ** -8
    assert rel(lin.lora_A.weight.grad, v.T @ xd) < 1e-4          # fp32 accumulation over the tokens
    assert rel(lin.lora_B.weight.grad, dyd.T @ u) < 1e-4
    # against peft's exact-arithmetic formulation (no bf16 intermediates): bf16-level agreement
    exact = xd @ W.T + s * (xd @ A.T) @ B.T
    assert rel(y.float(), exact) < 1e-2


def test_lora_starts_as_the_frozen_layer_and_refuses_cpu():
    from msr3d_amd.llm import LoRALinear
    lin = LoRALinear(256, 128, device="cuda")
    lin.load_base_weight(torch.randn(128, 256, device="cuda") / 16)
    x = torch.randn(10, 256, device="cuda").to(torch.bfloat16)
    assert rel(lin(x).float(), x.double() @ lin.weight.double().T) < 2 ** -8      # B = 0: no update yet
    assert set(dict(lin.named_parameters())) == {"lora_A.weight", "lora_B.weight"}   # W is a frozen buffer
    with pytest.raises(RuntimeError, match="GPU only"):
        lin(torch.randn(3, 256))


def test_roofline_line_for_the_projection(capsys):
    from msr3d_amd.llm import LoRALinear
    M, K, N = 2304, 4096, 4096                       # 4 sequences x 576 tokens, q_proj of Vicuna-7B
    lin = LoRALinear(K, N, device="cuda")
    lin.load_base_weight(torch.randn(N, K, device="cuda") / 64)
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16).requires_grad_(True)
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        lin(x).backward(dy)
    torch.cuda.synchronize()
    n = 20
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(n):
        y = lin(x)
    e[1].record()
    for _ in range(n):
        y = lin(x)
        y.backward(dy)
    e[2].record()
    torch.cuda.synchronize()
    fw = e[0].elapsed_time(e[1]) / n
    fb = e[1].elapsed_time(e[2]) / n
    flop = 2.0 * M * N * K
    with capsys.disabled():
        print(f"\n[lora] fwd {fw*1e3:.0f} us = {flop/fw/1e9:.0f} TFLOP/s ({flop/fw/1e9/2500:.2f} of 2.5 PF bf16); "
              f"fwd+bwd {fb*1e3:.0f} us = {2*flop/fb/1e9:.0f} TFLOP/s on the two big products")


def test_transposed_weight_follows_load_state_dict():
    """weight_t (W^T for dx = dy W) is a cache of `weight`; after load_state_dict or any other
    write to `weight` the backward must use the NEW transpose."""
    from msr3d_amd.llm import LoRALinear
    torch.manual_seed(0)
    K, N = 256, 192
    a = LoRALinear(K, N, r=16, device="cuda")
    a.load_base_weight(torch.randn(N, K, device="cuda") / K ** 0.5)
    b = LoRALinear(K, N, r=16, device="cuda")
    b.load_state_dict(a.state_dict())                         # `weight` arrives here, weight_t does not
    x = torch.randn(64, K, device="cuda").to(torch.bfloat16)
    dy = torch.randn(64, N, device="cuda").to(torch.bfloat16)
    outs = []
    for m in (a, b):
        xi = x.clone().requires_grad_(True)
        m(xi).backward(dy)
        outs.append(xi.grad.float())
    assert rel(outs[1], outs[0]) < 1e-6
    with torch.no_grad():
        b.weight.mul_(2.0)                                    # an in-place write bumps the version
    xi = x.clone().requires_grad_(True)
    b(xi).backward(dy)
    want = (dy.double() @ b.weight.double()).float()
    assert rel(xi.grad.float(), want) < 2 ** -7


@pytest.mark.parametrize("M,N,K,R", [(2304, 4096, 4096, 64), (2304, 1024, 512, 0), (1000, 4096, 2048, 0), (4600, 512, 1024, 64),
                                     (144, 256, 64, 0)])
def test_wide_gemm_exact_on_integer_operands_run_after_run(M, N, K, R):
    """Race screen of the wide-tile kernel's hand-off (LDS-DMA stages, counted vmcnt, two wave groups half a K step
    apart): small-integer bf16 operands make every product and every fp32 partial sum exact, so ONE stale or
    early-read fragment anywhere shows as a wrong integer.  Ten runs per shape, fp32 output."""
    import ctypes
    from msr3d_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    ri = lambda *s: torch.randint(-3, 4, s, device="cuda", generator=g).to(torch.bfloat16)   # noqa: E731
    P, Q = ri(M, K), ri(N, K)
    P2, Q2 = ri(M, max(R, 8)), ri(N, max(R, 8))
    want = P.float() @ Q.float().T + (P2[:, :R].float() @ Q2[:, :R].float().T if R else 0)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    lib, st = _lib.load(), _lib.current_stream_ptr(torch.device("cuda"))
    for _ in range(10):
        C = torch.full((M, N), float("nan"), device="cuda")
        rc = lib.msr3d_bf16_gemm_lowrank(M, N, K, R, vp(P), K, vp(Q), K, vp(P2) if R else None, P2.shape[1],
                                         vp(Q2) if R else None, Q2.shape[1], vp(C), N, 1, ctypes.c_float(1.0), st)
        _lib.check(rc, "msr3d_bf16_gemm_lowrank")
        assert torch.equal(C, want)
