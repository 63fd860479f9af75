"""GPU: the data-only front kernels (pairwise spatial features, agent-frame Fourier
features) against the torch formulation that mirrors the reference (and, through
tests/test_prompter_gpu.py, against the reference's own outputs)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,L", [(4, 60), (3, 61), (1, 1), (2, 128)])
def test_pairwise_locs_kernel(B, L):
    from msr3d_amd import hipops
    from msr3d_amd.modules.utils import calc_pairwise_locs
    torch.manual_seed(B + L)
    loc = torch.rand(B, L, 6, device="cuda") * 8
    loc[:, -3:] = 0.0                      # padded rows at the origin (also coincident centres)
    got = hipops.pairwise_locs_center5(loc)
    want = calc_pairwise_locs(loc[:, :, :3], loc[:, :, 3:])
    assert got.shape == want.shape == (B, L, L, 5)
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("B,L", [(4, 60), (2, 61), (1, 1)])
def test_agent_fourier_kernel(B, L):
    from msr3d_amd import hipops
    from msr3d_amd.model.ose3d_situation import generate_fourier_features
    from msr3d_amd.modules.utils import transform_to_agent_coor
    torch.manual_seed(B * 7 + L)
    loc = torch.rand(B, L, 6, device="cuda") * 8
    anchor = torch.rand(B, 3, device="cuda") * 8
    yaw = torch.rand(B, device="cuda") * 6.28 - 3.14
    quat = torch.stack([torch.zeros_like(yaw), torch.zeros_like(yaw), torch.sin(yaw / 2), torch.cos(yaw / 2)], 1)
    got = hipops.agent_fourier(loc, anchor, quat)
    want = generate_fourier_features(transform_to_agent_coor(loc[:, :, :3], anchor, quat))
    assert got.shape == want.shape == (B, L, 63)
    # arguments reach |pi * 11 m * 15| ~ 500 rad: a 1-ulp difference in p' moves sin by ~3e-5
    assert torch.allclose(got[..., :3], want[..., :3], rtol=1e-5, atol=1e-5)
    assert (got - want).abs().max() < 2e-3
    assert torch.allclose(got, want, rtol=0, atol=2e-3)
    # no transform: plain Fourier features are (near) bit-identical
    got2 = hipops.agent_fourier(loc)
    want2 = generate_fourier_features(loc[:, :, :3].contiguous())
    assert torch.allclose(got2, want2, rtol=0, atol=2e-5)


@pytest.mark.parametrize("seed", [0, 1])
def test_pairwise_kernels_against_the_references_own_output(seed):
    """Not through the torch mirror: the pairwise-location slab of the stand-alone kernel AND of the one-launch scene
    prologue against `calc_pairwise_locs` as the REFERENCE evaluated it (tests/golden/prompter_transform_seed*.npz,
    captured by tests/golden/make_golden.py from /root/reference/modules/utils.py:88-137)."""
    import ctypes
    import numpy as np
    from msr3d_amd import _lib, hipops
    from tests.helpers import load_golden, rel_l2
    g = load_golden("transform", seed)
    loc = torch.from_numpy(g["obj_locs"]).cuda().float().contiguous()
    want = g["pairwise_locs"]
    got = hipops.pairwise_locs_center5(loc)
    assert got.shape == want.shape and rel_l2(got.cpu().numpy(), want) < 1e-6
    B, L = loc.shape[:2]
    valid = torch.from_numpy(g["obj_masks"]).cuda().bool().contiguous()
    al = torch.from_numpy(g["anchor_locs"]).cuda().float().contiguous()
    ao = torch.from_numpy(g["anchor_orientation"]).cuda().float().contiguous()
    freqs = torch.linspace(1.0, 15, steps=10, device="cuda")
    pw, ff = torch.empty(B, L, L, 5, device="cuda"), torch.empty(B, L, 63, device="cuda")
    loc6, pad = torch.empty(B, L, 6, device="cuda"), torch.empty(B, L, dtype=torch.uint8, device="cuda")
    vout = torch.empty(B, L, dtype=torch.uint8, device="cuda")
    vp = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    rc = _lib.load().msr3d_scene_prologue(B, L, vp(loc), vp(valid.view(torch.uint8)), vp(al), vp(ao), vp(freqs), 10, 1,
                                          ctypes.c_float(1e-10), vp(pw), vp(ff), vp(loc6), vp(pad), vp(vout), None, None,
                                          _lib.current_stream_ptr(torch.device("cuda")))
    _lib.check(rc, "msr3d_scene_prologue")
    assert rel_l2(pw.cpu().numpy(), want) < 1e-6
    assert np.array_equal(pad.cpu().numpy().astype(bool), ~g["obj_masks"].astype(bool))


@pytest.mark.parametrize("B,O", [(3, 60), (2, 20), (8, 120), (1, 1)])
def test_scene_prologue_agent_against_the_module_formulation(B, O):
    """msr3d_scene_prologue_agent (situation_type 'as_object': the agent as token 0) through the C ABI against the
    module's own formulation (/root/reference/model/ose3d_situation.py:334-349 mirrored in OSE3DSituation._with_anchor_token):
    boxes, masks, pairwise features over the O + 1 tokens and the Fourier rows of the quaternion."""
    import ctypes
    from msr3d_amd import _lib
    from msr3d_amd.model.ose3d_situation import generate_fourier_features
    from msr3d_amd.modules.utils import calc_pairwise_locs
    torch.manual_seed(B * 31 + O)
    dev = torch.device("cuda")
    L = O + 1
    loc = (torch.rand(B, O, 6, device=dev) * 8).contiguous()
    valid = torch.rand(B, O, device=dev) > 0.3
    al = (torch.rand(B, 3, device=dev) * 8).contiguous()
    yaw = torch.rand(B, device=dev) * 6.28 - 3.14
    ao = torch.stack([torch.zeros_like(yaw), torch.zeros_like(yaw), torch.sin(yaw / 2), torch.cos(yaw / 2)], 1).contiguous()
    size = torch.tensor([0.7, 1.1, 1.6], device=dev)
    freqs = torch.linspace(1.0, 15, steps=10, device=dev)
    pw, ff = torch.empty(B, L, L, 5, device=dev), torch.empty(B, L, 63, device=dev)
    loc6, pad = torch.empty(B, L, 6, device=dev), torch.empty(B, L, dtype=torch.uint8, device=dev)
    vout, qf = torch.empty(B, L, dtype=torch.uint8, device=dev), torch.empty(B, 84, device=dev)
    al_out, ao_out = torch.empty_like(al), torch.empty_like(ao)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    rc = _lib.load().msr3d_scene_prologue_agent(B, O, vp(loc), vp(valid.view(torch.uint8).contiguous()), vp(al), vp(ao), vp(size),
                                                vp(freqs), 10, ctypes.c_float(1e-10), vp(pw), vp(ff), vp(loc6), vp(pad),
                                                vp(vout), vp(qf), vp(al_out), vp(ao_out), _lib.current_stream_ptr(dev))
    _lib.check(rc, "msr3d_scene_prologue_agent")
    want_loc = torch.cat((torch.cat((al, size.expand(B, 3)), 1).unsqueeze(1), loc), 1)
    want_valid = torch.cat((torch.ones(B, 1, dtype=torch.bool, device=dev), valid), 1)
    assert torch.equal(loc6, want_loc) and torch.equal(vout.bool(), want_valid) and torch.equal(pad.bool(), ~want_valid)
    assert torch.equal(al_out, al) and torch.equal(ao_out, ao)
    want_pw = calc_pairwise_locs(want_loc[:, :, :3], want_loc[:, :, 3:])
    assert torch.allclose(pw, want_pw, rtol=1e-6, atol=1e-7)
    want_qf = generate_fourier_features(ao.unsqueeze(1)).reshape(B, 84)
    assert torch.allclose(qf, want_qf, rtol=0, atol=2e-5)
    # EINVAL paths: the status comes back, nothing is launched
    assert _lib.load().msr3d_scene_prologue_agent(B, 128, vp(loc), vp(valid.view(torch.uint8)), vp(al), vp(ao), vp(size),
                                                  vp(freqs), 10, ctypes.c_float(1e-10), vp(pw), vp(ff), vp(loc6), vp(pad),
                                                  vp(vout), vp(qf), None, None, _lib.current_stream_ptr(dev)) != 0


@pytest.mark.parametrize("B,L", [(3, 61), (2, 21), (8, 121)])
def test_anchor_front_kernels_against_float64(B, L):
    """msr3d_anchor_front_fwd / _bwd through the C ABI against the same arithmetic in float64 torch with autograd: the token
    assembly (agent / object rows), LayerNorm(loc_layers[0](loc6)), and in backward the location layer's output gradient,
    the LayerNorm parameter gradients and the five column sums."""
    import ctypes
    from msr3d_amd import _lib
    torch.manual_seed(B + L)
    dev, D = torch.device("cuda"), 256
    M = B * L
    f = lambda *s: torch.randn(*s, device=dev)   # noqa: E731
    x0, a_ori, anchor, typ, ori, loc6 = f(M, D), f(B, D), f(D), f(2, D), f(D), torch.rand(M, 6, device=dev) * 4
    Wl, bl, gam, bet = f(D, 6) * 0.3, f(D) * 0.1, 1 + 0.1 * f(D), 0.1 * f(D)
    pos, s_lin, stats, xin0 = torch.empty(M, D, device=dev), torch.empty(M, D, device=dev), torch.empty(M, 2, device=dev), torch.empty(M, D, device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
    st = _lib.current_stream_ptr(dev)
    rc = _lib.load().msr3d_anchor_front_fwd(B, L, vp(x0), vp(a_ori), vp(anchor), vp(typ), vp(ori), vp(loc6), vp(Wl), vp(bl),
                                            vp(gam), vp(bet), ctypes.c_float(1e-5), vp(pos), vp(s_lin), vp(stats), vp(xin0), None, st)
    _lib.check(rc, "msr3d_anchor_front_fwd")
    # float64 reference with autograd
    leaves = [t.double().requires_grad_(True) for t in (x0, a_ori, anchor, typ, ori, Wl, bl, gam, bet)]
    x0d, aod, and_, tyd, ord_, Wd, bd, gd, btd = leaves
    agent = torch.zeros(M, dtype=torch.bool, device=dev)
    agent[::L] = True
    v = torch.where(agent[:, None], and_[None] + aod.repeat_interleave(L, 0) + tyd[1][None], x0d + ord_[None] + tyd[0][None])
    lin = loc6.double() @ Wd.t() + bd
    posd = torch.nn.functional.layer_norm(lin, (D,), gd, btd, 1e-5)
    want = v + posd
    rel = lambda a, b: float((a.double() - b).norm() / b.norm().clamp_min(1e-30))   # noqa: E731
    assert rel(xin0, want) < 1e-6 and rel(pos, posd) < 1e-6 and rel(s_lin, lin) < 1e-6
    d0, d1, d2 = f(M, D), f(M, D), f(M, D)
    # d v = d0; d pos = d0 + d1 + d2
    (v * d0.double()).sum().backward(retain_graph=True)
    (posd * (d0 + d1 + d2).double()).sum().backward()
    d_lin, dg, db = torch.empty(M, D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    oa, ob, oc, ga, gb = (torch.zeros(D, device=dev) for _ in range(5))
    rc = _lib.load().msr3d_anchor_front_bwd(B, L, vp(d0), vp(d1), vp(d2), vp(s_lin), vp(stats), vp(gam), vp(d_lin), vp(dg), vp(db),
                                            vp(oa), vp(ob), vp(oc), vp(ga), vp(gb), st)
    _lib.check(rc, "msr3d_anchor_front_bwd")
    torch.cuda.synchronize()
    assert rel(dg, gd.grad) < 1e-5 and rel(db, btd.grad) < 1e-5
    assert rel(oa, tyd.grad[0]) < 1e-5 and rel(ob, ord_.grad) < 1e-5 and rel(oc, ord_.grad) < 1e-5   # (float atomics: not bit-equal)
    assert rel(ga, tyd.grad[1]) < 1e-5 and rel(gb, and_.grad) < 1e-5
    # the location layer's output gradient: d_lin^T loc6 = dW, colsum = db
    assert rel(d_lin.double().t() @ loc6.double(), Wd.grad) < 1e-5 and rel(d_lin.double().sum(0), bd.grad) < 1e-5
    # d a_ori = the agent rows of d0; d x0 = the object rows of d0
    assert rel(d0[::L], aod.grad) < 1e-7
