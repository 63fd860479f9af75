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
