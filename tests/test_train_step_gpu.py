"""GPU: the HIP-graph-captured training step equals the eagerly issued one (dropout 0 so
both are deterministic), over several steps including the optimizer state."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(use_graph, steps=3):
    import msr3d_amd.model  # noqa: F401
    import msr3d_amd.modules  # noqa: F401
    from msr3d_amd.config import AttrDict, default_prompter_cfg
    from msr3d_amd.dp import FlatGradAllReduce
    from msr3d_amd.model import build_model
    from msr3d_amd.synth import synth_batch
    from msr3d_amd.train_step import HotPathTrainStep
    torch.manual_seed(0)
    cfg = AttrDict({"prompter": default_prompter_cfg(dropout=0.0), "llm_hidden_size": 128,
                    "model": {"name": "MSR3DHotPath"}})
    model = build_model(cfg).cuda().train()
    params = [p for p in model.parameters() if p.requires_grad]
    dp = FlatGradAllReduce(params)
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0.05, capturable=use_graph, foreach=True)
    batches = [synth_batch(50 + i, 2, O=10, P=1024, device="cuda") for i in range(2)]
    w = torch.randn(2, 10, 128, generator=torch.Generator().manual_seed(1)).cuda()
    step = HotPathTrainStep(model, opt, dp, lambda o: (o["scene_embeds"] * w).mean(), batches[0],
                            use_graph=use_graph)
    if use_graph:
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        step.capture(batches[0], warmup=2)          # warm-up steps mutate weights/optimizer:
        model.load_state_dict(sd)                   # restore both before comparing
        for st in opt.state.values():
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()
    losses = [float(step(batches[i % 2])) for i in range(steps)]
    torch.cuda.synchronize()
    return losses, {k: v.detach().clone() for k, v in model.named_parameters() if v.requires_grad}


def test_graph_replay_equals_eager():
    le, pe = _run(False)
    lg, pg = _run(True)
    # split-K partial sums meet by atomicAdd: summation order (hence the last bits) differs from
    # run to run, and AdamW's normalisation turns a noise-only gradient (w_ks.bias: mathematically
    # zero) into O(lr) steps of random sign -- exclude it, compare the rest at 1e-3 / 3 lr.
    assert le == pytest.approx(lg, rel=1e-4, abs=1e-6)
    for k in pe:
        if k.endswith("w_ks.bias"):
            continue
        assert torch.allclose(pe[k], pg[k], rtol=1e-3, atol=3e-3 * 1e-1), k


def _grads(packed):
    import msr3d_amd.model  # noqa: F401
    import msr3d_amd.modules  # noqa: F401
    from msr3d_amd import hipops
    from msr3d_amd.config import AttrDict, default_prompter_cfg
    from msr3d_amd.dp import FlatGradAllReduce
    from msr3d_amd.model import build_model
    from msr3d_amd.optim import FlatAdamW
    from msr3d_amd.synth import synth_batch
    torch.manual_seed(0)
    cfg = AttrDict({"prompter": default_prompter_cfg(dropout=0.0), "llm_hidden_size": 128,
                    "model": {"name": "MSR3DHotPath"}})
    model = build_model(cfg).cuda().train()
    params = [p for p in model.parameters() if p.requires_grad]
    dp = FlatGradAllReduce(params, pack_groups=hipops.collect_pack_groups(model) if packed else None)
    opt = FlatAdamW(dp, lr=1e-3)
    n = hipops.attach_packed_views(model, dp, opt) if packed else 0
    batch = synth_batch(77, 2, O=10, P=1024, device="cuda")
    dp.zero_grad()
    out = model(dict(batch))
    out["scene_embeds"].pow(2).mean().backward()
    dp.finish()
    torch.cuda.synchronize()
    return n, {k: v.grad.detach().clone() for k, v in model.named_parameters() if v.requires_grad}, model, opt


def test_packed_projection_views_match_separate_linears():
    n0, g0, _, _ = _grads(False)
    n1, g1, model, opt = _grads(True)
    assert n0 == 0 and n1 == 3                      # three attention blocks packed
    for k in g0:
        if k.endswith("w_ks.bias"):                 # zero gradient up to rounding noise
            continue
        ref = g0[k].double()
        assert float((g1[k].double() - ref).norm()) <= 2e-5 * float(ref.norm()) + 1e-9, k
    # the packed weight view aliases the four parameters' storage (zero copy)
    attn = model.visual_prompter.spatial_encoder[0].self_attn
    wv = attn._packed[0]
    assert wv.shape == (816, 256) and wv.data_ptr() == attn.w_qs.weight.data_ptr()
    assert torch.equal(wv[256:512], attn.w_ks.weight) and torch.equal(wv[768:], attn.lang_cond_fc.weight)


def test_prefetched_encoder_gives_identical_steps():
    """step(batch, next_batch) (encoder of the next batch on the side stream) == sequential."""
    import msr3d_amd.model  # noqa: F401
    import msr3d_amd.modules  # noqa: F401
    from msr3d_amd.config import AttrDict, default_prompter_cfg
    from msr3d_amd.dp import FlatGradAllReduce
    from msr3d_amd.model import build_model
    from msr3d_amd.optim import FlatAdamW
    from msr3d_amd.synth import synth_batch
    from msr3d_amd.train_step import HotPathTrainStep

    def run(pipelined):
        torch.manual_seed(0)
        cfg = AttrDict({"prompter": default_prompter_cfg(dropout=0.0), "llm_hidden_size": 64,
                        "model": {"name": "MSR3DHotPath"}})
        model = build_model(cfg).cuda().train()
        dp = FlatGradAllReduce([p for p in model.parameters() if p.requires_grad])
        opt = FlatAdamW(dp, lr=1e-3)
        batches = [synth_batch(200 + i, 2, O=8, P=1024, device="cuda") for i in range(3)]
        w = torch.randn(2, 8, 64, generator=torch.Generator().manual_seed(1)).cuda()
        step = HotPathTrainStep(model, opt, dp, lambda o: (o["scene_embeds"] * w).mean(), batches[0],
                                use_graph=False)
        losses = []
        for i in range(5):
            nb = batches[(i + 1) % 3] if pipelined else None
            losses.append(float(step(batches[i % 3], nb)))
        torch.cuda.synchronize()
        return losses

    assert run(True) == pytest.approx(run(False), rel=1e-4, abs=1e-6)
