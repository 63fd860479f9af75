"""fp8 (OCP e4m3) operands for the frozen projections of the LoRA-Llama layers (csrc/lora_fp8.hip):
the row quantiser bit for bit against a numpy restatement of round-to-nearest-even e4m3, the MX-instruction product
against float64 on the DEQUANTISED operands (what the kernel is supposed to compute exactly, up to fp32 accumulation),
and the LoRALinear / decoder layer in fp8 mode against the bf16 path and the transformers fixture with the tolerance
the 3-bit mantissa sets (stated per test)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def e4m3_decode(code):
    code = code.astype(np.int32)
    s, e, m = code >> 7, (code >> 3) & 15, code & 7
    v = np.where(e == 0, np.ldexp(m.astype(np.float64), -9), np.ldexp(1.0 + m / 8.0, e - 7))
    return np.where(s == 1, -v, v)


def e4m3_encode_rne(x):
    """float64 array with |x| <= 448 -> e4m3fn codes, round to nearest even (ties on the 3-bit mantissa grid)."""
    codes = np.arange(127, dtype=np.int32)                      # 0 .. 0x7e: the non-negative finite values, increasing
    vals = e4m3_decode(codes)
    a = np.abs(x)
    hi = np.searchsorted(vals, a, side="left").clip(0, 126)
    lo = (hi - 1).clip(0, 126)
    dlo, dhi = a - vals[lo], vals[hi] - a
    pick_hi = (dhi < dlo) | ((dhi == dlo) & ((codes[hi] & 1) == 0))
    c = np.where(pick_hi, codes[hi], codes[lo])
    c = np.where(a >= vals[126], 126, c)
    return (c | ((x < 0) | ((x == 0) & np.signbit(x))).astype(np.int32) << 7).astype(np.uint8)


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("M,K", [(37, 4096), (256, 11008), (5, 128)])
def test_row_quantiser_is_rne_e4m3_with_absmax_scales(M, K):
    from msr3d_amd.llm.lora import quant_rows_fp8
    torch.manual_seed(M)
    x = (torch.randn(M, K) * torch.exp(torch.randn(M, 1) * 2)).bfloat16().cuda()
    x[min(3, M - 1)] = 0                                           # a zero row: scale 1, codes 0
    q, sc = quant_rows_fp8(x)
    xd = x.double().cpu().numpy()
    amax = np.abs(xd).max(1)
    want_scale = np.where(amax > 0, (amax.astype(np.float32) * np.float32(1.0 / 448.0)), np.float32(1.0)).astype(np.float32)
    assert np.array_equal(sc.cpu().numpy(), want_scale)
    inv = (np.float32(1.0) / want_scale).astype(np.float32)
    scaled = (xd.astype(np.float32) * inv[:, None]).astype(np.float32).astype(np.float64)     # the kernel's fp32 product
    assert np.array_equal(q.cpu().numpy(), e4m3_encode_rne(scaled))


@pytest.mark.parametrize("M,N,K,lora", [(2304, 4096, 4096, True), (300, 512, 256, True), (144, 256, 128, False), (2304, 4096, 11008, True),
                                        # more tiles than CUs (2.7 and 5 rounds), short K loops with / without the LoRA stage
                                        (2304, 11008, 1024, True), (11520, 4096, 512, False), (6000, 4096, 256, True),
                                        (6000, 4096, 256, False), (5000, 11008, 384, False)])
def test_fp8_product_is_exact_on_the_dequantised_operands(M, N, K, lora):
    from msr3d_amd.llm.lora import PAD_R, _gemm_fp8, quant_rows_fp8
    torch.manual_seed(N + K)
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    xq, sx = quant_rows_fp8(x)
    wq, sw = quant_rows_fp8(w)
    u = b2 = None
    if lora:
        u = torch.zeros(M, PAD_R, device="cuda").bfloat16()
        b2 = torch.zeros(N, PAD_R, device="cuda").bfloat16()
        u[:, :16] = (torch.randn(M, 16, device="cuda") * 0.3).bfloat16()
        b2[:, :16] = (torch.randn(N, 16, device="cuda") * 0.3).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    _gemm_fp8(M, N, K, xq, sx, wq, sw, u, b2, y, x.device)
    torch.cuda.synchronize()
    xd = torch.from_numpy(e4m3_decode(xq.cpu().numpy())) * sx.double().cpu()[:, None]
    wd = torch.from_numpy(e4m3_decode(wq.cpu().numpy())) * sw.double().cpu()[:, None]
    want = xd @ wd.T
    if lora:
        want = want + u.double().cpu() @ b2.double().cpu().T
    assert rel(y, want) < 4e-3                                     # bf16 output rounding (2^-9) + fp32 accumulation
    # and against the unquantised product: the e4m3 rounding of both operands, ~2^-4 / sqrt(3) each
    full = x.double().cpu() @ w.double().cpu().T + (u.double().cpu() @ b2.double().cpu().T if lora else 0)
    assert rel(y, full) < 6e-2
    # msr3d_fp8_gemm_lowrank_acc: C += the product (what the input gradients of q / k / v meet through)
    c0 = torch.randn(M, N, device="cuda").bfloat16()
    acc = c0.clone()
    _gemm_fp8(M, N, K, xq, sx, wq, sw, u, b2, acc, x.device, accumulate=True)
    torch.cuda.synchronize()
    assert rel(acc, want + c0.double().cpu()) < 4e-3
    assert float((acc.float() - (y.float() + c0.float())).abs().max()) <= 2.0 ** -6 * float((y.float().abs() + c0.float().abs()).max())


def test_lora_linear_fp8_against_the_bf16_module():
    from msr3d_amd.llm import LoRALinear
    torch.manual_seed(0)
    M, K, N = 2304, 4096, 4096
    a = LoRALinear(K, N, device="cuda", base="bf16")
    b = LoRALinear(K, N, device="cuda", base="fp8")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    with torch.no_grad():
        for m in (a, b):
            m.load_base_weight(w)
        a.lora_B.weight.normal_(std=0.05)
        b.lora_A.weight.copy_(a.lora_A.weight)
        b.lora_B.weight.copy_(a.lora_B.weight)
    assert b.weight_q.dtype == torch.uint8 and b.weight_q.shape == (N, K) and b.weight_t_q.shape == (K, N)
    x = torch.randn(M, K, device="cuda").bfloat16()
    gy = (torch.randn(M, N, device="cuda") * 0.01).bfloat16()
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = a(xa), b(xb)
    ya.backward(gy)
    yb.backward(gy)
    assert rel(yb, ya) < 6e-2 and rel(xb.grad, xa.grad) < 6e-2
    # the LoRA gradients see the quantisation only through dy's path: dA = (s dy B)^T x uses the bf16 x and dy
    assert rel(b.lora_A.weight.grad, a.lora_A.weight.grad) < 1e-2 and rel(b.lora_B.weight.grad, a.lora_B.weight.grad) < 1e-2
    with pytest.raises(ValueError):
        LoRALinear(192, 256, base="fp8")


def test_decoder_layer_fp8_against_the_transformers_fixture():
    """The LlamaDecoderLayer fixture of tests/test_llama_layer_gpu.py with the frozen projections in e4m3: output within
    5e-2 rel-L2, dx and the LoRA gradients within 1e-1 (bf16 path: 2-3e-2)."""
    from msr3d_amd.llm import LoRALlamaDecoderLayer
    from tests.helpers import llama_layer_weights
    g = dict(np.load(os.path.join(GOLD, "llama_layer_seed0.npz")))
    hidden, heads, inter, r, alpha, B, T = (int(v) for v in g["cfg"])
    w = llama_layer_weights(int(g["seed"]), hidden, inter, r)
    layer = LoRALlamaDecoderLayer(hidden, heads, inter, r=r, lora_alpha=alpha, rms_eps=float(g["eps"]),
                                  rope_theta=float(g["theta"]), device="cuda", base="fp8")
    names = ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]
    with torch.no_grad():
        for n in names:
            m = (layer.self_attn if n in layer.self_attn else layer.mlp)[n]
            m.load_base_weight(torch.from_numpy(w[n]).cuda())
            m.lora_A.weight.copy_(torch.from_numpy(w[n + ".A"]))
            m.lora_B.weight.copy_(torch.from_numpy(w[n + ".B"]))
        layer.input_layernorm_weight.copy_(torch.from_numpy(w["ln1"]))
        layer.post_attention_layernorm_weight.copy_(torch.from_numpy(w["ln2"]))
    x = torch.from_numpy(g["x"]).cuda().bfloat16().requires_grad_(True)
    keep = torch.from_numpy(g["keep"]).cuda()
    y = layer(x, attention_mask=keep)
    rows = keep.bool()
    gy = torch.from_numpy(g["gy"]).cuda() * rows[..., None]
    y.backward(gy.bfloat16())
    assert rel(y.float()[rows], torch.from_numpy(g["y"]).cuda()[rows]) < 5e-2
    assert rel(x.grad.float()[rows], torch.from_numpy(g["dx"]).cuda()[rows]) < 1e-1
    for n in names:
        m = (layer.self_attn if n in layer.self_attn else layer.mlp)[n]
        assert rel(m.lora_A.weight.grad, g["dA/" + n]) < 1e-1, n
        assert rel(m.lora_B.weight.grad, g["dB/" + n]) < 1e-1, n


@pytest.mark.parametrize("M,K,N", [(2304, 4096, 16), (300, 11008, 48), (129, 256, 32), (16, 4096, 64)])
def test_skinny_product_with_fused_quantisation_is_both_kernels_bit_for_bit(M, K, N):
    """msr3d_bf16_gemm_skinny_quant == msr3d_bf16_gemm_skinny + msr3d_quant_rows_fp8 on the same tensor, bit for bit
    (codes, scales, the r-row product and its zero padding); rows of zeros keep scale 1."""
    import ctypes

    from msr3d_amd import _lib
    from msr3d_amd.llm.lora import quant_rows_fp8
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + K + N)
    x = (torch.randn(M, K, generator=g, device="cuda") * torch.rand(M, 1, generator=g, device="cuda") * 8).to(torch.bfloat16)
    x[M // 2] = 0
    a = torch.randn(N, K, generator=g, device="cuda").to(torch.bfloat16)
    st = _lib.current_stream_ptr(torch.device("cuda:0"))
    p = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
    u0 = torch.full((M, 64), 7.0, dtype=torch.bfloat16, device="cuda")
    u1 = u0.clone()
    assert lib.msr3d_bf16_gemm_skinny(M, N, K, p(x), K, p(a), K, p(u0), 64, 64, ctypes.c_float(0.5), st) == 0
    q0, s0 = quant_rows_fp8(x)
    q1 = torch.empty_like(q0)
    s1 = torch.empty_like(s0)
    assert lib.msr3d_bf16_gemm_skinny_quant(M, N, K, p(x), K, p(a), K, p(u1), 64, 64, ctypes.c_float(0.5), p(q1), K, p(s1),
                                            st) == 0
    torch.cuda.synchronize()
    assert torch.equal(u0, u1) and torch.equal(q0, q1) and torch.equal(s0, s1)
    assert float(s1[M // 2]) == 1.0 and not q1[M // 2].any()
    assert lib.msr3d_bf16_gemm_skinny_quant(M, N, K, p(x), K, p(a), K, p(u1), 64, 64, ctypes.c_float(0.5), None, K, p(s1),
                                            st) == -22
