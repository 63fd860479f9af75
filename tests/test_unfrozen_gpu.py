"""Unfrozen-backbone training (`freeze: False`): the composite path --
the nine HIP ops + their deterministic backward under torch's conv/BN(train) -- forward and
backward against (1) the same ops expressed in torch on the same device and (2) the same modules
on the CPU with the oracle standing in for the ops.  Tolerances are stated in the test."""
import copy

import pytest
import torch

from oracle import pn2
from tests.helpers import fill_state_dict, rel_l2

pytestmark = pytest.mark.gpu


class _TorchScatterExt:
    """The op surface with the three *_grad entries (and group/gather) re-expressed in torch on the
    same device; index ops come from the HIP library.  Same device => identical BN / ReLU / max
    selections, so any difference isolates the hand-written forward/backward kernels."""

    def __init__(self, hip_ext):
        self._hip = hip_ext

    def __getattr__(self, name):
        return getattr(self._hip, name)

    @staticmethod
    def group_points(points, idx):
        b, c, n = points.shape
        flat = idx.long().reshape(b, 1, -1).expand(b, c, -1)
        return points.gather(2, flat).reshape(b, c, idx.shape[1], idx.shape[2])

    @staticmethod
    def group_points_grad(grad_out, idx, n):
        b, c = grad_out.shape[:2]
        out = torch.zeros(b, c, n, device=grad_out.device, dtype=torch.float64)
        flat = idx.long().reshape(b, 1, -1).expand(b, c, -1)
        return out.scatter_add_(2, flat, grad_out.double().reshape(b, c, -1)).float()

    @staticmethod
    def gather_points(points, idx):
        return points.gather(2, idx.long().unsqueeze(1).expand(-1, points.shape[1], -1))

    @staticmethod
    def gather_points_grad(grad_out, idx, n):
        b, c = grad_out.shape[:2]
        out = torch.zeros(b, c, n, device=grad_out.device, dtype=torch.float64)
        return out.scatter_add_(2, idx.long().unsqueeze(1).expand(-1, c, -1), grad_out.double()).float()


def _run(model, ext, fts, w, monkeypatch):
    from msr3d_amd.pointnet2 import pointnet2_utils
    monkeypatch.setattr(pointnet2_utils, "_ext", ext)
    model.zero_grad(set_to_none=True)
    out, _ = model(fts)
    (out * w).sum().backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    stats = {n: b.detach().clone() for n, b in model.named_buffers() if "running" in n}
    return out.detach(), grads, stats


def test_unfrozen_encoder_forward_backward(monkeypatch):
    """(1) HIP ops vs the same ops written in torch on the GPU: tight (1e-5), ReLU masks and max
    selections being identical; (2) vs the CPU with the oracle: forward 1e-4; gradients only to
    3e-2, because a pre-activation within rounding of zero (a handful among ~6 M in train-mode BN)
    flips its ReLU mask between devices and moves the fp32 gradient by ~1e-3 -- a property of the
    network, observed on about half of the seeds."""
    from msr3d_amd.modules.vision.pcd_pointnet_encoder import PcdObjEncoder
    from msr3d_amd.pointnet2 import _ext as hip_ext
    from msr3d_amd.synth import synth_batch
    kw = dict(sa_n_points=[32, 16, None], sa_n_samples=[32, 32, None], sa_radii=[0.2, 0.4, None],
              sa_mlps=[[3, 64, 64, 128], [128, 128, 128, 256], [256, 256, 512, 768]], dropout=0.0,
              freeze=False)
    cpu = PcdObjEncoder(None, **kw)
    cpu.load_state_dict(fill_state_dict(cpu.state_dict(), 11))
    cpu.train()
    fts = synth_batch(5, 1, O=12, P=1024)["obj_fts"]
    w = torch.randn(1, 12, 768)

    # (the SharedMLPs run on the token GEMMs in training mode: ordered split-K, so that "identical
    # inputs" still means "identical bits" for the level-3 layers, whose few rows make them split)
    from msr3d_amd import hipops
    was = hipops.set_deterministic(True)
    try:
        gpu = copy.deepcopy(cpu).cuda()
        out_h, g_h, s_h = _run(gpu, hip_ext, fts.cuda(), w.cuda(), monkeypatch)
        gpu2 = copy.deepcopy(cpu).cuda()
        out_t, g_t, s_t = _run(gpu2, _TorchScatterExt(hip_ext), fts.cuda(), w.cuda(), monkeypatch)
    finally:
        hipops.set_deterministic(was)
    assert torch.equal(out_h, out_t)                              # forward ops are exact copies
    assert len(g_h) >= 28
    for n in g_h:
        if g_t[n].abs().max() < 1e-6:        # conv bias in front of BN(train): mathematically zero
            continue
        assert rel_l2(g_h[n].cpu().numpy(), g_t[n].cpu().numpy()) < 1e-5, n

    out_c, g_c, s_c = _run(cpu, pn2.ext_module(), fts, w, monkeypatch)
    assert rel_l2(out_h.cpu().numpy(), out_c.numpy()) < 1e-4
    assert sorted(g_c) == sorted(g_h)
    for n in g_c:
        if g_c[n].abs().max() < 1e-6:
            assert g_h[n].abs().max() < 1e-3, n
            continue
        assert rel_l2(g_h[n].cpu().numpy(), g_c[n].numpy()) < 3e-2, n
    for n in s_c:                                                 # running statistics updated alike
        assert rel_l2(s_h[n].cpu().numpy(), s_c[n].numpy()) < 1e-4, n


def test_unfrozen_backward_is_run_to_run_reproducible():
    """With the ordered-sum *_grad kernels the gradient reaching the input features through
    group_points / gather_points is bit-identical across runs."""
    from msr3d_amd.pointnet2 import pointnet2_utils as pu
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(8, 16, 1024, generator=g).cuda().requires_grad_()
    xyz = torch.rand(8, 1024, 3, generator=g).cuda()
    idx = pu.furthest_point_sample(xyz, 32)
    new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    ball = pu.ball_query(0.3, 32, xyz, new_xyz)
    wgt = torch.randn(8, 16, 32, 32, generator=g).cuda()
    grads = []
    for _ in range(3):
        feats.grad = None
        (pu.grouping_operation(feats, ball) * wgt).sum().backward()
        grads.append(feats.grad.clone())
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
    assert grads[0].abs().sum() > 0


# ---------------------------------------------------------------------------------------
# SharedMLP in training mode on this build's kernels (hipops.shared_mlp_train: token GEMMs +
# csrc/bn_train.hip) against torch's conv / batch_norm / relu / amax on the same device.
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,C", [(5000, 64), (4096, 128), (8193, 256), (37, 4), (700, 768), (513, 1024)])
def test_bn_relu_train_matches_float64(R, C):
    """Outputs, gradients and the running statistics of the fused BatchNorm(train) + ReLU against a
    float64 evaluation: 1e-5 rel-L2 (fp32 data, statistics accumulated in double)."""
    from msr3d_amd import hipops
    torch.manual_seed(R + C)
    x = (torch.randn(R, C, device="cuda") * 2.0 + 0.7).requires_grad_()
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.5)
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
    ref = copy.deepcopy(bn).double()
    gy = torch.randn(R, C, device="cuda")

    y = hipops._BNReLUTrain.apply(x, bn.weight, bn.bias, bn)
    y.backward(gy)

    xd = x.detach().double().requires_grad_()
    yd = torch.relu(torch.nn.functional.batch_norm(
        xd, ref.running_mean, ref.running_var, ref.weight, ref.bias, training=True, momentum=0.1, eps=bn.eps))
    yd.backward(gy.double())
    assert rel_l2(y.detach().cpu().numpy(), yd.detach().cpu().numpy()) < 1e-5
    assert rel_l2(x.grad.cpu().numpy(), xd.grad.cpu().numpy()) < 2e-5
    assert rel_l2(bn.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy()) < 2e-5
    assert rel_l2(bn.running_var.cpu().numpy(), ref.running_var.cpu().numpy()) < 1e-5
    assert rel_l2(bn.bias.grad.cpu().numpy(), ref.bias.grad.cpu().numpy()) < 2e-5
    assert rel_l2(bn.running_mean.cpu().numpy(), ref.running_mean.cpu().numpy()) < 1e-5
    assert int(bn.num_batches_tracked) == 1
    # ordered reductions: a second evaluation is bit-identical
    bn2 = copy.deepcopy(bn)
    y2 = hipops._BNReLUTrain.apply(x.detach(), bn.weight.detach(), bn.bias.detach(), bn2)
    assert torch.equal(y2, y.detach())
    with pytest.raises(ValueError, match="more than 1 value per channel"):      # torch's own refusal
        hipops._BNReLUTrain.apply(x.detach()[:1], bn.weight.detach(), bn.bias.detach(), bn2)


@pytest.mark.parametrize("spec,shape", [([6, 64, 64, 128], (3, 6, 32, 32)), ([131, 128, 128, 256], (2, 131, 16, 32)),
                                        ([259, 256, 512, 768], (5, 259, 1, 16))])
def test_shared_mlp_train_matches_torch_modules(spec, shape):
    from msr3d_amd import hipops
    from msr3d_amd.pointnet2.pytorch_utils import SharedMLP
    torch.manual_seed(sum(spec))
    mlp = SharedMLP(list(spec), bn=True).cuda().train()
    ref = copy.deepcopy(mlp).double()
    x = torch.randn(*shape, device="cuda", requires_grad=True)
    assert hipops.shared_mlp_train_supported(mlp, x)
    g = torch.randn(shape[0], spec[-1], shape[2], device="cuda")
    out = hipops.shared_mlp_train(mlp, x)
    (out * g).sum().backward()
    xd = x.detach().double().requires_grad_()
    outd = torch.amax(ref(xd), dim=3)
    (outd * g.double()).sum().backward()
    assert out.shape == outd.shape
    assert rel_l2(out.detach().cpu().numpy(), outd.detach().cpu().numpy()) < 2e-5
    assert rel_l2(x.grad.cpu().numpy(), xd.grad.cpu().numpy()) < 1e-4
    for (n, p), (_, q) in zip(mlp.named_parameters(), ref.named_parameters()):
        assert rel_l2(p.grad.cpu().numpy(), q.grad.cpu().numpy()) < 1e-4, n
    for (n, a), (_, b) in zip(mlp.named_buffers(), ref.named_buffers()):
        if "running" in n:
            assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-5, n
        else:
            assert int(a) == int(b) == 1, n
    # eval mode, CPU tensors, or the switch turned off keep the module path
    mlp.eval()
    assert not hipops.shared_mlp_train_supported(mlp, x)
    mlp.train()
    mlp.use_hip_train = False
    assert not hipops.shared_mlp_train_supported(mlp, x)


@pytest.mark.parametrize("n,m,ns,C,radius", [(1024, 32, 32, 3, 0.2), (32, 16, 32, 128, 0.4), (50, 7, 5, 0, 0.5)])
def test_group_rows_is_query_and_group_token_major(n, m, ns, C, radius):
    """msr3d_group_rows = QueryAndGroup's output permuted to (b, point, sample, channel), bit for bit;
    its backward = the deterministic group_points_grad, bit for bit (same ascending-entry sums)."""
    from msr3d_amd import hipops
    from msr3d_amd.pointnet2 import pointnet2_utils as pu
    torch.manual_seed(n + C)
    b = 5
    xyz = torch.rand(b, n, 3, device="cuda")
    new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), pu.furthest_point_sample(xyz, m)
                                  ).transpose(1, 2).contiguous()
    feats = torch.randn(b, C, n, device="cuda").requires_grad_() if C else None
    assert hipops.group_rows_supported(xyz, new_xyz, feats, ns)
    idx = pu.ball_query(radius, ns, xyz, new_xyz)
    KP = (3 + C + 3) // 4 * 4
    rows = hipops._GroupRows.apply(xyz, new_xyz, feats, idx, KP)
    grouped = pu.QueryAndGroup(radius, ns, use_xyz=True)(xyz, new_xyz, feats)       # (b, 3+C, m, ns)
    want = grouped.permute(0, 2, 3, 1).reshape(b * m * ns, 3 + C)
    assert torch.equal(rows[:, :3 + C], want.detach())
    assert int(rows[:, 3 + C:].abs().sum()) == 0
    if C:
        g = torch.randn_like(rows)
        rows.backward(g)
        got = feats.grad.clone()
        feats.grad = None
        grouped.backward(g[:, :3 + C].reshape(b, m, ns, 3 + C).permute(0, 3, 1, 2).contiguous())
        assert torch.equal(got, feats.grad)
    # coordinates that need a gradient keep the composite path
    assert not hipops.group_rows_supported(xyz.clone().requires_grad_(), new_xyz, feats, ns)


def test_sa_module_train_rows_path_matches_grouped_path():
    """A PointnetSAModule in training mode: the token-major rows path against the same kernels fed
    by the channel-major grouper (forced by coordinates that require a gradient)."""
    from msr3d_amd.pointnet2.pointnet2_modules import PointnetSAModule
    torch.manual_seed(4)
    sa = PointnetSAModule(mlp=[128, 128, 128, 256], npoint=16, radius=0.4, nsample=32).cuda().train()
    sb = copy.deepcopy(sa)
    xyz = torch.rand(6, 32, 3, device="cuda")
    f1 = torch.randn(6, 128, 32, device="cuda").requires_grad_()
    f2 = f1.detach().clone().requires_grad_()
    w = torch.randn(6, 256, 16, device="cuda")
    _, out1 = sa(xyz, f1)
    (out1 * w).sum().backward()
    _, out2 = sb(xyz.clone().requires_grad_(), f2)          # composite grouper, same SharedMLP kernels
    (out2 * w).sum().backward()
    assert rel_l2(out1.detach().cpu().numpy(), out2.detach().cpu().numpy()) < 1e-6
    assert rel_l2(f1.grad.cpu().numpy(), f2.grad.cpu().numpy()) < 1e-5
    for (n, p), (_, q) in zip(sa.named_parameters(), sb.named_parameters()):
        assert rel_l2(p.grad.cpu().numpy(), q.grad.cpu().numpy()) < 1e-5, n
    for (n, a), (_, c) in zip(sa.named_buffers(), sb.named_buffers()):
        assert rel_l2(a.float().cpu().numpy(), c.float().cpu().numpy()) < 1e-6, n


@pytest.mark.parametrize("G,ns,C", [(300, 32, 128), (64, 16, 768), (7, 5, 4), (1000, 32, 64)])
def test_bn_relu_maxpool_train_matches_max_pool2d_float64(G, ns, C):
    """The last SharedMLP layer fused with the neighbourhood max-pool against float64
    batch_norm -> relu -> F.max_pool2d([1, nsample]) (the reference's pooling, first maximum wins):
    pooled values, dx, dgamma, dbeta, running statistics; duplicated rows (what ball_query's padding
    produces) exercise the tie rule."""
    import torch.nn.functional as F
    from msr3d_amd import hipops
    torch.manual_seed(G + ns + C)
    x = torch.randn(G, ns, C, device="cuda") * 1.5 + 0.3
    x[::3, ns - 1] = x[::3, 0]                       # exact duplicates: the maximum may be tied
    x[1::3, ns // 2] = x[1::3, 1]
    x = x.reshape(G * ns, C).requires_grad_()
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.5)
    ref = copy.deepcopy(bn).double()
    gp = torch.randn(G, C, device="cuda")

    pooled = hipops._BNReLUMaxPoolTrain.apply(x, bn.weight, bn.bias, bn, ns)
    pooled.backward(gp)

    xd = x.detach().double().requires_grad_()
    yd = torch.relu(F.batch_norm(xd, ref.running_mean, ref.running_var, ref.weight, ref.bias, training=True,
                                 momentum=0.1, eps=bn.eps))
    pd = F.max_pool2d(yd.view(G, ns, C).permute(2, 0, 1)[None], kernel_size=[1, ns])[0, :, :, 0].t()   # (G, C)
    pd.backward(gp.double())
    assert rel_l2(pooled.detach().cpu().numpy(), pd.detach().cpu().numpy()) < 1e-5
    assert rel_l2(x.grad.cpu().numpy(), xd.grad.cpu().numpy()) < 5e-5
    assert rel_l2(bn.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy()) < 5e-5
    assert rel_l2(bn.bias.grad.cpu().numpy(), ref.bias.grad.cpu().numpy()) < 5e-5
    assert rel_l2(bn.running_var.cpu().numpy(), ref.running_var.cpu().numpy()) < 1e-5
    assert int(bn.num_batches_tracked) == 1


def test_hot_path_model_trains_an_unfrozen_backbone_in_a_plain_loop():
    """`freeze: False` end to end: MSR3DHotPath forward / backward / AdamW in an ordinary torch loop --
    backbone, situated encoder and projector all receive gradients, the loss goes down, BN running
    statistics move; the train step takes this configuration only with the encoder's parameters in its
    gradient engine (tests/test_unfrozen_step_gpu.py covers the step itself)."""
    import msr3d_amd.model  # noqa: F401
    import msr3d_amd.modules  # noqa: F401
    from msr3d_amd.config import AttrDict, default_prompter_cfg
    from msr3d_amd.dp import FlatGradAllReduce
    from msr3d_amd.model import build_model
    from msr3d_amd.synth import synth_batch
    from msr3d_amd.train_step import HotPathTrainStep
    torch.manual_seed(0)
    cfg = AttrDict({"prompter": default_prompter_cfg(freeze=False, dropout=0.0), "llm_hidden_size": 64,
                    "model": {"name": "MSR3DHotPath"}})
    model = build_model(cfg).cuda().train()
    enc = model.visual_prompter.obj_encoder
    assert not enc.freeze and all(p.requires_grad for p in enc.pcd_net.parameters())
    batch = synth_batch(9, 2, O=12, P=1024, device="cuda")
    rm0 = enc.pcd_net.encoder[0].mlps[0].layer0.bn.bn.running_mean.clone()
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    target = torch.randn(2, 12, 64, device="cuda")
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        out = model(dict(batch))["scene_embeds"]
        loss = (out - target).square().mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    w = enc.pcd_net.encoder[0].mlps[0].layer0.conv.weight
    assert w.grad is not None and float(w.grad.abs().sum()) > 0
    assert model.llm_proj.weight.grad is not None
    assert losses[-1] < losses[0]
    assert not torch.equal(rm0, enc.pcd_net.encoder[0].mlps[0].layer0.bn.bn.running_mean)
    dp = FlatGradAllReduce([p for n, p in model.named_parameters() if p.requires_grad and "obj_encoder" not in n])
    with pytest.raises(ValueError, match="owned by the gradient engine"):
        HotPathTrainStep(model, opt, dp, lambda o: o["scene_embeds"].sum(), batch)
