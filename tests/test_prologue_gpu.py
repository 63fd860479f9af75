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
