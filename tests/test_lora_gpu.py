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
                                     (64, 256, 192, 32), (4000, 512, 256, 16)])      # (>= 3584 rows: 16-row skinny tiles)
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


@pytest.mark.parametrize("M,K,N,r,njobs", [(2304, 4096, 11008, 16, 2), (576, 512, 1024, 16, 2), (100, 256, 72, 32, 2),
                                            (2304, 4096, 4096, 16, 1), (64, 128, 128, 16, 2), (1031, 136, 264, 16, 2)])
def test_lora_grad_pair_one_launch_bit_reproducible(M, K, N, r, njobs):
    """msr3d_lora_grad_pair: dA = v^T x and dB^T = u^T dy in ONE launch, a workgroup per 64 output columns over all
    tokens: vs float64, accumulate on / off, ragged token counts, run-to-run identical."""
    import ctypes

    from msr3d_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + K)
    bf = lambda *s: torch.randn(*s, generator=g, device="cuda").to(torch.bfloat16)      # noqa: E731
    v, u = torch.zeros(M, 64, dtype=torch.bfloat16, device="cuda"), torch.zeros(M, 64, dtype=torch.bfloat16, device="cuda")
    v[:, :r], u[:, :r] = bf(M, r), bf(M, r)
    x, dy = bf(M, K), bf(M, N)
    st = _lib.current_stream_ptr(torch.device("cuda:0"))
    refA = (v[:, :r].double().t() @ x.double())
    refB = (dy.double().t() @ u[:, :r].double())
    outs = []
    for acc in (0, 1, 1):
        dA = torch.full((r, K), 3.0, device="cuda")
        dB = torch.full((N, r), -2.0, device="cuda")
        jobs = (_lib.LoraGradJob * 2)(_lib.LoraGradJob(K, v.data_ptr(), 64, x.data_ptr(), K, dA.data_ptr(), 0),
                                      _lib.LoraGradJob(N, u.data_ptr(), 64, dy.data_ptr(), N, dB.data_ptr(), 1))
        assert lib.msr3d_lora_grad_pair(M, r, njobs, jobs, ctypes.c_float(0.5), acc, st) == 0
        torch.cuda.synchronize()
        base = 3.0 if acc else 0.0
        err = (dA.double() - (base + 0.5 * refA)).abs().max() / refA.abs().max()
        assert err < 2e-6, err
        if njobs == 2:
            base = -2.0 if acc else 0.0
            err = (dB.double() - (base + 0.5 * refB)).abs().max() / refB.abs().max()
            assert err < 2e-6, err
        else:
            assert torch.all(dB == -2.0)
        outs.append((dA.clone(), dB.clone()))
    assert torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1])
    # argument checks: C not a multiple of 8, a misaligned operand, three jobs
    bad = (_lib.LoraGradJob * 2)(_lib.LoraGradJob(K - 4, v.data_ptr(), 64, x.data_ptr(), K, dA.data_ptr(), 0), jobs[1])
    assert lib.msr3d_lora_grad_pair(M, r, 2, bad, ctypes.c_float(1.0), 1, st) == -22
    bad = (_lib.LoraGradJob * 2)(_lib.LoraGradJob(K, v.data_ptr() + 2, 64, x.data_ptr(), K, dA.data_ptr(), 0), jobs[1])
    assert lib.msr3d_lora_grad_pair(M, r, 2, bad, ctypes.c_float(1.0), 1, st) == -22
    assert lib.msr3d_lora_grad_pair(M, r, 3, jobs, ctypes.c_float(1.0), 1, st) == -22


def test_lora_shadows_one_launch_for_all_pairs():
    """msr3d_lora_shadows via lora.refresh_shadows: the four bf16 images of several pairs at once == the per-tensor
    conversions; rebuilt only when A / B were written; padding columns stay zero."""
    from msr3d_amd.llm import lora
    torch.manual_seed(0)
    mods = [lora.LoRALinear(256, 512, r=16, device="cuda"), lora.LoRALinear(512, 128, r=32, device="cuda"),
            lora.LoRALinear(4096, 11008, r=16, device="cuda")]
    for m in mods:
        with torch.no_grad():
            m.lora_B.weight.normal_()
    lora.refresh_shadows(mods)
    for m in mods:
        a_pad, b2, bt_pad, at2 = m._shadow
        A, Bw, r = m.lora_A.weight.detach(), m.lora_B.weight.detach(), m.r
        assert torch.equal(a_pad, A.to(torch.bfloat16)) and torch.equal(bt_pad, Bw.t().to(torch.bfloat16))
        assert torch.equal(b2[:, :r], Bw.to(torch.bfloat16)) and torch.equal(at2[:, :r], A.t().to(torch.bfloat16))
        assert not b2[:, r:].any() and not at2[:, r:].any()
        assert m._shadow_key == m._pair_key()
    keep = mods[0]._shadow[0].clone()
    with torch.no_grad():
        mods[1].lora_A.weight.mul_(2.0)                 # only this pair is stale now
    lora.refresh_shadows(mods)
    assert torch.equal(mods[0]._shadow[0], keep)
    assert torch.equal(mods[1]._shadow[0], mods[1].lora_A.weight.detach().to(torch.bfloat16))
    assert mods[1]._shadows()[0] is mods[1]._shadow[0]


@pytest.mark.parametrize("M,N,K,R", [(2304, 4096, 4096, 64), (300, 512, 256, 0)])
def test_wide_gemm_accumulate_adds_into_a_bf16_output(M, N, K, R):
    """msr3d_bf16_gemm_lowrank_acc: C += P Q^T + P2 Q2^T on the wide-tile kernel (the shared d-input buffer of projections
    that read one tensor); outside that kernel's domain the entry refuses."""
    from msr3d_amd.llm import lora
    g = torch.Generator(device="cuda").manual_seed(M + N)
    mk = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 0.5).to(torch.bfloat16)      # noqa: E731
    P, Q, C0 = mk(M, K), mk(N, K) / K ** 0.5, mk(M, N)
    P2 = mk(M, 64) if R else None
    Q2 = mk(N, 64) if R else None
    C = C0.clone()
    assert lora._gemm_acc(M, N, K, R, P, K, Q, K, P2, 64 if R else 0, Q2, 64 if R else 0, C, N, 1.0, P.device)
    want = P.double() @ Q.double().t() + C0.double()
    if R:
        want = want + P2.double() @ Q2.double().t()
    torch.cuda.synchronize()
    assert rel(C, want) < 4e-3
    assert not lora._gemm_acc(64, N, K, R, P, K, Q, K, P2, 64 if R else 0, Q2, 64 if R else 0, C, N, 1.0, P.device)
