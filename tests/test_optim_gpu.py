"""GPU: the flat-buffer clip + AdamW kernels against torch.optim.AdamW + clip_grad_norm_
(the reference's optimiser stack) over several steps, with and without the LR schedule."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def make(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(37, 64), torch.nn.LayerNorm(64), torch.nn.Linear(64, 5)).cuda()


@pytest.mark.parametrize("sched", ["constant", "warmup_cosine_instructblip"])
def test_flat_adamw_matches_torch(sched):
    from msr3d_amd.dp import FlatGradAllReduce
    from msr3d_amd.optim import FlatAdamW
    a, b = make(0), make(0)
    dp = FlatGradAllReduce(a.parameters())
    warm, total = 3, 10
    opt_a = FlatAdamW(dp, lr=1e-2, weight_decay=0.05, max_grad_norm=0.5, schedule=sched,
                      warmup_steps=warm, total_steps=total)
    opt_b = torch.optim.AdamW(b.parameters(), lr=1e-2, betas=(0.9, 0.999), weight_decay=0.05)

    def lam(step):
        if sched == "constant":
            return 1.0
        if step <= warm:
            return 1e-3 + step / warm * (1 - 1e-3)
        return 0.5 * (1 + math.cos((step - warm) / (total - warm) * math.pi))
    sch = torch.optim.lr_scheduler.LambdaLR(opt_b, lam)
    for it in range(7):
        x = torch.randn(16, 37, device="cuda", generator=torch.Generator("cuda").manual_seed(it))
        dp.zero_grad()
        (a(x).pow(2).sum() * 10).backward()
        dp.finish()
        opt_a.step()
        opt_b.zero_grad()
        (b(x).pow(2).sum() * 10).backward()
        torch.nn.utils.clip_grad_norm_(b.parameters(), 0.5)
        opt_b.step()
        sch.step()
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.allclose(pa, pb, rtol=2e-5, atol=2e-6), it
    assert int(opt_a.step_ctr.item()) == 7
    # parameters are views of the flat buffer and the module still works / saves
    assert all(p.data_ptr() >= opt_a.flat_p.data_ptr() for p in a.parameters())
    assert set(a.state_dict().keys()) == set(b.state_dict().keys())


def test_dot_is_exact_enough_and_bit_reproducible():
    from msr3d_amd import hipops
    g = torch.Generator().manual_seed(3)
    for n in (4, 1024, 960 * 4096, 1000 * 1000):
        a = torch.randn(n, generator=g).cuda()
        b = torch.randn(n, generator=g).cuda()
        want = float((a.double() * b.double()).sum())
        got = [float(hipops.dot(a, b)) for _ in range(3)]
        assert got[0] == got[1] == got[2]
        assert abs(got[0] - want) <= 1e-5 * float((a.double() * b.double()).abs().sum()) + 1e-6


@pytest.mark.parametrize("world", [2, 3, 8])
def test_grad_scale_in_the_optimiser_is_the_separate_averaging_pass_bit_for_bit(world):
    """msr3d_adamw_flat_scaled(grad_scale = 1 / world) on the all-reduced SUM == `grads *= 1 / world` followed
    by the unscaled call: same clip coefficient, same moments, same weights, for a world that is not a power
    of two as well (/root/reference/trainer/leo_trainer.py:50-52: DDP averages, then accelerate clips)."""
    from msr3d_amd.dp import FlatGradAllReduce
    from msr3d_amd.optim import FlatAdamW
    a, b = make(1), make(1)
    dpa, dpb = FlatGradAllReduce(a.parameters()), FlatGradAllReduce(b.parameters())
    oa = FlatAdamW(dpa, lr=1e-2, weight_decay=0.05, max_grad_norm=0.5)
    ob = FlatAdamW(dpb, lr=1e-2, weight_decay=0.05, max_grad_norm=0.5)
    dpa.world, dpa.scale_in_optimizer = world, True
    for it in range(4):
        x = torch.randn(16, 37, device="cuda", generator=torch.Generator("cuda").manual_seed(it))
        for m, dp in ((a, dpa), (b, dpb)):
            dp.zero_grad()
            (m(x).pow(2).sum() * 10 * world).backward()       # "the sum over ranks"
        dpb.flat.mul_(1.0 / world)
        oa.step(zero_grad=True)
        ob.step()
        assert torch.equal(oa.flat_p, ob.flat_p) and torch.equal(oa.exp_avg_sq, ob.exp_avg_sq), it
        assert float(dpa.flat.abs().max()) == 0.0
