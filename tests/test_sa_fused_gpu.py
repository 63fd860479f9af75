"""GPU parity of the fused set-abstraction kernels (msr3d_sa_fps2 / msr3d_sa_level):
  * the index work inside them (two FPS levels, two ball queries) is bit-exact vs the oracle;
  * features vs (a) the composite path (HIP index ops + torch conv/BN/ReLU/max) and (b) the
    REFERENCE's encoder output in tests/golden/.  fp32 on the matrix cores, k-order differs
    from MIOpen's: tolerance rel-L2 <= 2e-5 (measured ~1e-6)."""
import numpy as np
import pytest
import torch

from oracle import pn2
from tests.helpers import build_prompter, fill_state_dict, load_golden, rel_l2

pytestmark = pytest.mark.gpu
TOL = 2e-5


def make_net(seed, n_points=(32, 16, None)):
    from msr3d_amd.modules.layers.pointnet import PointNetPP
    net = PointNetPP(sa_n_points=list(n_points), sa_n_samples=[32, 32, None],
                     sa_radii=[0.2, 0.4, None],
                     sa_mlps=[[3, 64, 64, 128], [128, 128, 128, 256], [256, 256, 512, 768]])
    sd = fill_state_dict(net.state_dict(), seed)
    if seed % 2:   # negative BN gammas: the affine must be applied BEFORE the max
        for k in sd:
            if k.endswith("bn.bn.weight"):
                sd[k] = sd[k] * torch.where(torch.arange(sd[k].numel()) % 3 == 0, -1.0, 1.0)
    net.load_state_dict(sd)
    return net.cuda().eval()


def clouds(seed, b, P=1024):
    from msr3d_amd.synth import synth_batch
    d = synth_batch(seed, 1, O=b, P=P, n_valid=max(1, b - 2))
    return d["obj_fts"][0].contiguous()     # (b, P, 6), last two objects = all-ones padding


@pytest.mark.parametrize("b,seed,npts", [(5, 0, (32, 16, None)), (8, 1, (32, 16, None)),
                                         (1, 2, (32, 16, None)), (3, 3, (30, 16, None)),
                                         (4, 4, (17, 16, None))])
@pytest.mark.parametrize("fps_query", [True, False])
def test_fused_indices_bit_exact_and_features_close(b, seed, npts, fps_query, monkeypatch):
    """(fps_query: level 1's ball query beside the furthest-point sampling in ONE launch, msr3d_sa_fps2_query -- or the
    two launches; both bit-exact against the oracle)"""
    from msr3d_amd.pointnet2 import fused
    monkeypatch.setattr(fused, "_FPS_QUERY", fps_query)
    net = make_net(seed, npts)
    pts = clouds(seed, b).cuda()
    with torch.no_grad():
        assert fused.can_fuse(net, pts)
        out, dbg = fused.forward(net, pts, return_internals=True)
        net.use_fused = False
        ref = net(pts)
        net.use_fused = True
    m1 = npts[0]
    xyz = pts[..., :3].contiguous().cpu().numpy()
    i1 = pn2.furthest_point_sampling(xyz, m1)
    assert np.array_equal(dbg["idx1"].cpu().numpy(), i1)
    nx1 = np.take_along_axis(xyz, i1[..., None].astype(np.int64).repeat(3, -1), 1)
    assert np.array_equal(dbg["new_xyz1"].cpu().numpy(), nx1)
    assert np.array_equal(dbg["ball1"].cpu().numpy(), pn2.ball_query(nx1, xyz, 0.2, 32))
    i2 = pn2.furthest_point_sampling(nx1, 16)
    assert np.array_equal(dbg["idx2"].cpu().numpy(), i2)
    nx2 = np.take_along_axis(nx1, i2[..., None].astype(np.int64).repeat(3, -1), 1)
    assert np.array_equal(dbg["new_xyz2"].cpu().numpy(), nx2)
    assert np.array_equal(dbg["ball2"].cpu().numpy(), pn2.ball_query(nx2, nx1, 0.4, 32))
    assert torch.isfinite(out).all()
    assert rel_l2(out.cpu().numpy(), ref.cpu().numpy()) < TOL


def test_fused_levels_match_composite_levels():
    """Level by level against the composite modules (channel-major there, point-major here)."""
    from msr3d_amd.pointnet2 import fused
    net = make_net(7)
    pts = clouds(7, 6).cuda()
    with torch.no_grad():
        _, dbg = fused.forward(net, pts, return_internals=True)
        xyz = pts[..., :3].contiguous()
        feats = pts[..., 3:].transpose(1, 2).contiguous()
        x1, f1 = net.encoder[0](xyz, feats)
        x2, f2 = net.encoder[1](x1, f1)
        _, f3 = net.encoder[2](x2, f2)
    assert rel_l2(dbg["feat1"].transpose(1, 2).cpu().numpy(), f1.cpu().numpy()) < TOL
    assert rel_l2(dbg["feat2"].transpose(1, 2).cpu().numpy(), f2.cpu().numpy()) < TOL
    assert rel_l2(dbg["pooled"].cpu().numpy(), f3.squeeze(-1).cpu().numpy()) < TOL


@pytest.mark.parametrize("variant,seed", [("transform", 0), ("anchor", 1)])
def test_fused_encoder_vs_reference_golden(variant, seed):
    g = load_golden(variant, seed)
    model = build_prompter(variant, seed, device="cuda")
    fts = torch.from_numpy(g["obj_fts"]).cuda()
    with torch.no_grad():
        enc, _ = model.obj_encoder(fts)
    assert getattr(model.obj_encoder.pcd_net, "_fused_plan", None) is not None, "fused path not taken"
    assert rel_l2(enc.cpu().numpy(), g["enc_out"]) < TOL


def test_fused_falls_back_when_backbone_trains():
    from msr3d_amd.pointnet2 import fused
    net = make_net(0)
    pts = clouds(0, 2).cuda()
    assert not fused.can_fuse(net, pts)           # grad enabled + trainable params
    with torch.no_grad():
        assert fused.can_fuse(net, pts)
        net.train()
        assert not fused.can_fuse(net, pts)       # BN on batch statistics
    net.eval()
    out = net(pts)                                 # composite path, autograd works
    out.sum().backward()
    assert net.encoder[0].mlps[0][0][0].weight.grad is not None


def test_padding_slots_skipped_gives_identical_features():
    """`skip_padded`: masked slots (the dataset's constant padding cloud) are not encoded again;
    every row -- real objects and padding -- is bit-identical to the full encoding, for mask
    patterns incl. all-valid, all-padding, odd/even pairs (level 3 shares a workgroup between two
    objects)."""
    import msr3d_amd.modules  # noqa: F401
    from msr3d_amd.modules.vision.pcd_pointnet_encoder import PcdObjEncoder
    from msr3d_amd.synth import synth_batch
    from tests.helpers import fill_state_dict
    enc = PcdObjEncoder(None, freeze=True)
    enc.load_state_dict(fill_state_dict(enc.state_dict(), 4))
    enc = enc.cuda().eval()
    batch = synth_batch(17, 4, O=15, P=1024, n_valid=[15, 0, 7, 8], device="cuda")
    fts, masks = batch["obj_fts"], batch["obj_masks"]
    assert masks.sum(1).tolist() == [15, 0, 7, 8]
    full = enc.embed(fts)
    enc.skip_padded = True
    got = enc.embed(fts, masks)
    assert torch.equal(got, full)
    # interleaved pattern (padding between real objects) on real data only in the valid slots
    m2 = masks.clone()
    m2[0, ::2] = False
    f2 = fts.clone()
    f2[0, ::2] = 1.0
    enc.skip_padded = False
    want = enc.embed(f2)
    enc.skip_padded = True
    assert torch.equal(enc.embed(f2, m2), want)
    # without masks the flag changes nothing
    assert torch.equal(enc.embed(fts), full)


@pytest.mark.parametrize("mma", ["split", "f32"])
@pytest.mark.parametrize("variant,seed", [("transform", 0), ("transform", 1), ("anchor", 0), ("anchor", 1)])
def test_every_level_against_the_references_own_level_outputs(variant, seed, mma):
    """Level by level against what the REFERENCE's PointnetSAModule chain produced (tests/golden/make_golden.py::
    capture_encoder_internals: FPS picks and ball-query rows of both sampled levels for every object, the level
    outputs of the first two objects, the encoder output): the indices inside the fused launches bit for bit, the
    features of every level -- split (bf16 x 3) and f32-input kernels alike -- within the fp32 tolerance."""
    from msr3d_amd.pointnet2 import fused
    g = load_golden(variant, seed)
    model = build_prompter(variant, seed, device="cuda")
    net = model.obj_encoder.pcd_net
    fts = torch.from_numpy(g["obj_fts"]).cuda()
    pts = fts.reshape(-1, fts.shape[2], fts.shape[3]).contiguous()
    prev = fused.set_sa_mma(mma)
    try:
        with torch.no_grad():
            _, dbg = fused.forward(net, pts, return_internals=True)
    finally:
        fused.set_sa_mma(prev)
    b = pts.shape[0]
    assert np.array_equal(dbg["idx1"].cpu().numpy(), g["sa0_fps_idx"].reshape(b, -1))
    assert np.array_equal(dbg["ball1"].cpu().numpy().reshape(b, -1), g["sa0_ball_idx"].reshape(b, -1))
    assert np.array_equal(dbg["idx2"].cpu().numpy(), g["sa1_fps_idx"].reshape(b, -1))
    assert np.array_equal(dbg["ball2"].cpu().numpy().reshape(b, -1), g["sa1_ball_idx"].reshape(b, -1))
    # level outputs: the reference's are channel-major (2, C, npoint), the kernels' token-major (b, npoint, C)
    assert rel_l2(dbg["feat1"][:2].permute(0, 2, 1).cpu().numpy(), g["sa0_out_first2"]) < TOL
    assert rel_l2(dbg["feat2"][:2].permute(0, 2, 1).cpu().numpy(), g["sa1_out_first2"]) < TOL
    assert rel_l2(dbg["pooled"][:2].cpu().numpy(), g["sa2_out_first2"].reshape(2, -1)) < TOL
