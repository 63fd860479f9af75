"""GPU: fused spatial-attention core (msr3d_spatial_attn_fwd/bwd) against the composite
torch formulation of the same module evaluated in float64.  fp32 kernels, exact-f32 MFMA:
tolerance rel-L2 <= 2e-5 on outputs and gradients (attention outputs' stated tolerance)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("B,L,pad_frac", [(4, 60, 0.3), (3, 61, 0.0), (2, 64, 0.5), (5, 17, 0.2), (1, 1, 0.0),
                                          (2, 121, 0.3), (2, 128, 0.0), (3, 65, 0.5), (1, 120, 0.1)])
def test_fused_attention_matches_composite_fp64(B, L, pad_frac):
    from msr3d_amd.modules.layers.transformers import MultiHeadAttentionSpatial
    torch.manual_seed(B * 100 + L)
    m = MultiHeadAttentionSpatial(256, 8, dropout=0.0, spatial_multihead=True, spatial_dim=5,
                                  spatial_attn_fusion="cond").cuda()
    with torch.no_grad():          # make the spatial term matter (incl. the 1e-6 clamp region)
        m.lang_cond_fc.weight.mul_(8.0)
        m.lang_cond_fc.bias.normal_(0, 3.0)
    md = MultiHeadAttentionSpatial(256, 8, dropout=0.0, spatial_multihead=True, spatial_dim=5,
                                   spatial_attn_fusion="cond").cuda().double()
    md.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    md.use_fused_core = False
    x = torch.randn(B, L, 256, device="cuda", requires_grad=True)
    pl = torch.randn(B, L, L, 5, device="cuda")
    mask = torch.rand(B, L, device="cuda") < pad_frac
    mask[:, 0] = False                                   # at least one valid key per sample
    g = torch.randn(B, L, 256, device="cuda")

    y, p = m(x, x, x, pl, key_padding_mask=mask)
    (y * g).sum().backward()
    xd = x.detach().double().requires_grad_()
    yd, pd = md(xd, xd, xd, pl.double(), key_padding_mask=mask)
    (yd * g.double()).sum().backward()

    assert p.shape == pd.shape == (8, B, L, L)
    assert rel(p, pd) < 2e-5
    assert rel(y, yd) < 2e-5
    assert rel(x.grad, xd.grad) < 5e-5
    for (n, a), (_, b) in zip(m.named_parameters(), md.named_parameters()):
        if n == "w_ks.bias":      # mathematically zero gradient
            assert a.grad.abs().max() < 1e-3
            continue
        assert rel(a.grad, b.grad) < 5e-5, n


def test_fused_core_is_used_and_composite_still_available():
    from msr3d_amd import hipops
    from msr3d_amd.modules.layers.transformers import MultiHeadAttentionSpatial
    m = MultiHeadAttentionSpatial(256, 8, dropout=0.0, spatial_dim=5, spatial_attn_fusion="cond").cuda()
    x = torch.randn(2, 60, 256, device="cuda")
    assert hipops.spatial_attn_cond_supported(x, 8, 5, 8)
    assert hipops.spatial_attn_cond_supported(torch.randn(2, 121, 256, device="cuda"), 8, 5, 8)
    assert not hipops.spatial_attn_cond_supported(torch.randn(2, 129, 256, device="cuda"), 8, 5, 8)
    pl = torch.randn(2, 60, 60, 5, device="cuda")
    y1, _ = m(x, x, x, pl)
    m.use_fused_core = False
    y2, _ = m(x, x, x, pl)
    assert rel(y1, y2) < 2e-5
    # L > 128 (beyond the stress config) takes the composite path
    x3 = torch.randn(1, 130, 256, device="cuda")
    m.use_fused_core = True
    y3, p3 = m(x3, x3, x3, torch.randn(1, 130, 130, 5, device="cuda"))
    assert y3.shape == (1, 130, 256) and p3.shape == (8, 1, 130, 130)


# ---------------------------------------------------------------------------------------
# optional operand precisions (include/msr3d_hip.h MSR3D_MMA_*): bf16 forward + backward, fp8
# (OCP e4m3) forward.  Two checks each:
#   * against a float64 evaluation that rounds the SAME operands to the SAME format (q, k, v, and P
#     after the fp32 softmax): what is left is accumulation order and the rare operand whose
#     fp32 / fp64 value straddles a rounding boundary -- rel-L2 <= 2e-3;
#   * against the exact float64 result: the precision's own error, stated per format below.
# ---------------------------------------------------------------------------------------
def _core_inputs(B, L, seed, pad_frac=0.25):
    torch.manual_seed(seed)
    D, H = 256, 8
    qkvc = torch.randn(B, L, 3 * D + H * 6, device="cuda")
    qkvc[..., :2 * D] *= 0.8
    pl = torch.randn(B, L, L, 5, device="cuda")
    mask = torch.rand(B, L, device="cuda") < pad_frac
    mask[:, 0] = False
    return qkvc, pl, mask


def _core_reference(qkvc, pl, mask, rnd, p_scale=1.0):
    """float64 restatement of transformers.py:205-248 on the packed projections, with `rnd`
    applied where the kernel rounds its matrix operands."""
    B, L, W = qkvc.shape
    D, H, dh = 256, 8, 32
    x = qkvc.double()
    q, k, v = (rnd(x[..., j * D:(j + 1) * D]).view(B, L, H, dh).permute(0, 2, 1, 3) for j in range(3))
    cond = x[..., 3 * D:].view(B, L, H, 6).permute(0, 2, 1, 3)           # (B,H,L,6)
    s = torch.einsum("bhld,bhtd->bhlt", q, k) / (dh ** 0.5)
    z = cond[..., 0:1] + torch.einsum("bhld,bltd->bhlt", cond[..., 1:], pl.double())
    loc = torch.sigmoid(z)
    logits = torch.log(loc.clamp_min(1e-6)) + s
    logits = logits.masked_fill(mask[:, None, None, :], float("-inf"))
    p = torch.softmax(logits, dim=-1)
    pr = rnd(p.float().double() * p_scale) / p_scale                        # fp32 softmax, then rounded
    ctx = torch.einsum("bhlt,bhtd->bhld", pr, v).permute(0, 2, 1, 3).reshape(B, L, D)
    return ctx, p


def _round_to(dtype):
    return lambda t: t.float().to(dtype).double()


@pytest.mark.parametrize("mma,dtype,tol_exact", [("bf16", torch.bfloat16, 1e-2),
                                                 ("fp8", torch.float8_e4m3fn, 8e-2)])
@pytest.mark.parametrize("B,L", [(4, 60), (2, 61), (2, 120), (1, 17)])
def test_reduced_precision_forward_against_emulation(mma, dtype, tol_exact, B, L):
    from msr3d_amd import hipops
    qkvc, pl, mask = _core_inputs(B, L, seed=L + B)
    prev = hipops.set_attention_mma(mma)
    try:
        with torch.no_grad():
            ctx, probs = hipops.spatial_attn_cond(qkvc, pl, mask, 8, 256)
    finally:
        hipops.set_attention_mma(prev)
    want, p_want = _core_reference(qkvc, pl, mask, _round_to(dtype), 256.0 if mma == "fp8" else 1.0)
    exact, p_exact = _core_reference(qkvc, pl, mask, lambda t: t)
    assert rel(ctx, want) < 2e-3
    assert rel(probs, p_want) < 2e-3                 # probabilities are returned unrounded (fp32)
    assert rel(ctx, exact) < tol_exact
    assert rel(ctx, exact) > 1e-5                    # the reduced-precision kernel did run
    assert rel(probs, p_exact) < tol_exact


@pytest.mark.parametrize("B,L", [(4, 60), (2, 61), (2, 120)])
def test_bf16_attention_module_forward_backward(B, L):
    """Whole MultiHeadAttentionSpatial in bf16 mode against the float64 composite: outputs and every
    gradient within 1e-2 rel-L2 (bf16 operands, fp32 accumulation and softmax)."""
    from msr3d_amd import hipops
    from msr3d_amd.modules.layers.transformers import MultiHeadAttentionSpatial
    torch.manual_seed(7 * L + B)
    m = MultiHeadAttentionSpatial(256, 8, dropout=0.0, spatial_multihead=True, spatial_dim=5,
                                  spatial_attn_fusion="cond").cuda()
    md = MultiHeadAttentionSpatial(256, 8, dropout=0.0, spatial_multihead=True, spatial_dim=5,
                                   spatial_attn_fusion="cond").cuda().double()
    md.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    md.use_fused_core = False
    x = torch.randn(B, L, 256, device="cuda", requires_grad=True)
    pl = torch.randn(B, L, L, 5, device="cuda")
    mask = torch.rand(B, L, device="cuda") < 0.3
    mask[:, 0] = False
    g = torch.randn(B, L, 256, device="cuda")
    prev = hipops.set_attention_mma("bf16")
    try:
        y, p = m(x, x, x, pl, key_padding_mask=mask)
        (y * g).sum().backward()
    finally:
        hipops.set_attention_mma(prev)
    xd = x.detach().double().requires_grad_()
    yd, pd = md(xd, xd, xd, pl.double(), key_padding_mask=mask)
    (yd * g.double()).sum().backward()
    assert rel(y, yd) < 1e-2 and rel(p, pd) < 1e-2
    assert rel(y, yd) > 1e-6
    assert rel(x.grad, xd.grad) < 1e-2
    for (n, a), (_, b) in zip(m.named_parameters(), md.named_parameters()):
        if n == "w_ks.bias":
            assert a.grad.abs().max() < 1e-2
            continue
        assert rel(a.grad, b.grad) < 1e-2, n


def test_fp8_attention_is_forward_only():
    from msr3d_amd import hipops
    qkvc, pl, mask = _core_inputs(2, 60, seed=3)
    qkvc.requires_grad_()
    prev = hipops.set_attention_mma("fp8")
    try:
        ctx, _ = hipops.spatial_attn_cond(qkvc, pl, mask, 8, 256)
        with pytest.raises(RuntimeError, match="forward-only"):
            ctx.sum().backward()
    finally:
        hipops.set_attention_mma(prev)
    with pytest.raises(ValueError):
        hipops.set_attention_mma("fp4")
