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


def _accum_run(accum, use_graph, explicit_grad):
    """One optimiser step over the SAME 4 scenes, either as one batch or as `accum` micro-batches."""
    import msr3d_amd.model  # noqa: F401
    import msr3d_amd.modules  # noqa: F401
    from msr3d_amd import hipops
    from msr3d_amd.config import AttrDict, default_prompter_cfg
    from msr3d_amd.dp import FlatGradAllReduce
    from msr3d_amd.model import build_model
    from msr3d_amd.optim import FlatAdamW
    from msr3d_amd.synth import synth_batch
    from msr3d_amd.train_step import HotPathTrainStep
    torch.manual_seed(0)
    cfg = AttrDict({"prompter": default_prompter_cfg(dropout=0.0), "llm_hidden_size": 128,
                    "model": {"name": "MSR3DHotPath"}})
    model = build_model(cfg).cuda().train()
    params = [p for p in model.parameters() if p.requires_grad]
    dp = FlatGradAllReduce(params, pack_groups=hipops.collect_pack_groups(model))
    opt = FlatAdamW(dp, lr=1e-3, weight_decay=0.05, max_grad_norm=5.0)
    hipops.attach_packed_views(model, dp, opt)
    full = synth_batch(91, 4, O=10, P=1024, device="cuda")
    per = 4 // accum
    micro = [{k: v[i * per:(i + 1) * per].contiguous() for k, v in full.items()} for i in range(accum)]
    w = torch.randn(4, 10, 128, generator=torch.Generator().manual_seed(1)).cuda()
    # the loss weights of the current micro-batch live in STATIC tensors (a captured graph replays
    # fixed addresses; Python-side indexing would be frozen at capture time)
    wi = torch.empty_like(w[:per])
    gi = torch.empty_like(w[:per])

    def select(i):
        wi.copy_(w[i * per:(i + 1) * per])
        gi.copy_(wi / wi.numel())

    def loss_fn(o):                     # mean over the micro-batch's own scenes (as the reference's loss)
        y = o["scene_embeds"]
        if explicit_grad:
            return (y.detach() * wi).mean(), y, gi
        return (y * wi).mean()

    select(0)

    step = HotPathTrainStep(model, opt, dp, loss_fn, micro[0], use_graph=use_graph, accum_steps=accum)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    if use_graph:
        osd = opt.state_dict()
        step.capture(micro[0], warmup=1)            # warm-up steps mutate weights / optimiser state
        model.load_state_dict(sd)
        opt.load_state_dict(osd)
    for i in range(accum):
        select(i)
        step(micro[i])
    torch.cuda.synchronize()
    weights = {k: v.detach().clone() for k, v in model.named_parameters() if v.requires_grad}
    weights["__flat_grad__"] = dp.flat.detach().clone()       # what the optimiser consumed
    return weights, sd


@pytest.mark.parametrize("explicit_grad", [False, True])
def test_gradient_accumulation_equals_the_full_batch_step(explicit_grad):
    """accum_steps = 2 / 4 micro-batches == the same scenes as one batch (mean loss, scale
    1/accum per micro-batch as accelerate does; no cross-sample coupling: BN frozen), eager."""
    ref, sd = _accum_run(1, False, explicit_grad)
    ref.pop("__flat_grad__")
    moved = max(float((ref[k] - sd[k]).abs().max()) for k in ref)
    assert moved > 1e-4                                            # the step did something
    for accum in (2, 4):
        got, _ = _accum_run(accum, False, explicit_grad)
        for k in ref:
            if k.endswith("w_ks.bias"):
                continue
            assert torch.allclose(got[k], ref[k], rtol=1e-4, atol=2e-5), (accum, k)


def test_gradient_accumulation_with_the_captured_graph():
    """The graph then holds ONE micro-batch's forward/backward; zeroing, exchange and optimiser run
    eagerly around accum_steps replays."""
    ref, _ = _accum_run(2, False, True)
    got, _ = _accum_run(2, True, True)
    # compare the accumulated gradient (Adam's first update is lr * sign(g): entries whose gradient
    # is rounding noise flip sign between two runs of the atomically-summed GEMMs)
    g0, g1 = ref.pop("__flat_grad__").double(), got.pop("__flat_grad__").double()
    assert float((g1 - g0).norm() / g0.norm()) < 1e-5
    close = [torch.allclose(got[k], ref[k], rtol=1e-4, atol=2e-5) for k in ref]
    assert sum(close) >= len(close) - 3


def _det_run(steps=3, use_graph=True):
    import msr3d_amd.model  # noqa: F401
    import msr3d_amd.modules  # noqa: F401
    from msr3d_amd import hipops
    from msr3d_amd.config import AttrDict, default_prompter_cfg
    from msr3d_amd.dp import FlatGradAllReduce
    from msr3d_amd.model import build_model
    from msr3d_amd.optim import FlatAdamW
    from msr3d_amd.synth import synth_batch
    from msr3d_amd.train_step import HotPathTrainStep
    torch.manual_seed(0)
    hipops._seed_words.clear()                  # dropout seed word re-derived from torch's seed
    hipops._salt_counter[0] = 0
    cfg = AttrDict({"prompter": default_prompter_cfg(dropout=0.1), "llm_hidden_size": 128,
                    "model": {"name": "MSR3DHotPath"}})
    model = build_model(cfg).cuda().train()
    params = [p for p in model.parameters() if p.requires_grad]
    dp = FlatGradAllReduce(params, pack_groups=hipops.collect_pack_groups(model))
    opt = FlatAdamW(dp, lr=1e-3, weight_decay=0.05, max_grad_norm=5.0)
    hipops.attach_packed_views(model, dp, opt)
    batches = [synth_batch(60 + i, 4, O=12, P=1024, device="cuda") for i in range(2)]
    w = torch.randn(4, 12, 128, generator=torch.Generator().manual_seed(1)).cuda()
    step = HotPathTrainStep(model, opt, dp, lambda o: (o["scene_embeds"] * w).mean(), batches[0],
                            use_graph=use_graph)
    if use_graph:
        step.capture(batches[0], warmup=1)
    for i in range(steps):
        step(batches[i % 2])
    torch.cuda.synchronize()
    return opt.flat_p.detach().clone(), dp.flat.detach().clone()


@pytest.mark.parametrize("use_graph", [False, True])
def test_deterministic_mode_gives_bit_identical_weights(monkeypatch, use_graph):
    """hipops.set_deterministic(True) (MSR3D_DETERMINISTIC=1): ordered split-K, ordered LayerNorm
    gamma/beta and bias reductions -- two runs from the same seeds, dropout ON, end in the same bits.
    (The default mode lets split-K partial sums and those reductions meet by float atomics.)"""
    from msr3d_amd import hipops
    monkeypatch.setattr(hipops, "_deterministic", [True])
    p1, g1 = _det_run(use_graph=use_graph)
    p2, g2 = _det_run(use_graph=use_graph)
    assert torch.equal(g1, g2), int((g1 != g2).sum())
    assert torch.equal(p1, p2), int((p1 != p2).sum())
    assert torch.isfinite(p1).all()


def test_direct_staging_still_copies_the_batch_tensors_the_loss_reads():
    """in the schedule's direct-staging mode (`_load` lets the one-launch prologue read the
    batch's geometry where it is) every OTHER tensor of the batch must still reach the static buffers the
    captured step sees -- here a per-batch weighting tensor consumed by loss_fn from the scene dict."""
    import msr3d_amd.model  # noqa: F401
    import msr3d_amd.modules  # noqa: F401
    from msr3d_amd import hipops
    from msr3d_amd.config import AttrDict, default_prompter_cfg
    from msr3d_amd.dp import FlatGradAllReduce
    from msr3d_amd.model import build_model
    from msr3d_amd.optim import FlatAdamW
    from msr3d_amd.synth import synth_batch
    from msr3d_amd.train_step import HotPathTrainStep
    losses = []
    for use_graph in (True, False):
        torch.manual_seed(0)
        cfg = AttrDict({"prompter": default_prompter_cfg(dropout=0.0), "llm_hidden_size": 128,
                        "model": {"name": "MSR3DHotPath"}})
        model = build_model(cfg).cuda().train()
        params = [p for p in model.parameters() if p.requires_grad]
        dp = FlatGradAllReduce(params, pack_groups=hipops.collect_pack_groups(model))
        opt = FlatAdamW(dp, lr=0.0, weight_decay=0.0)            # lr 0: the steps are independent
        hipops.attach_packed_views(model, dp, opt)
        batches = []
        for i in range(3):
            b = synth_batch(90 + i, 2, O=10, P=1024, device="cuda")
            b["target_w"] = torch.full((2, 10, 128), float(i + 1), device="cuda")
            batches.append(b)

        def loss_fn(out):
            return (out["scene_embeds"] * out["target_w"]).mean()
        step = HotPathTrainStep(model, opt, dp, loss_fn, batches[0], use_graph=use_graph)
        step.capture(batches[0])
        ls = [float(step(b)) for b in batches]
        assert step._sched_direct
        losses.append(ls)
        # same scene, different weighting tensor: the loss must scale with it
        b0 = dict(batches[0]); b0["target_w"] = batches[0]["target_w"] * 4.0
        assert float(step(b0)) == pytest.approx(4.0 * ls[0], rel=1e-5)
    assert losses[0] == pytest.approx(losses[1], rel=1e-5)


def test_accumulation_window_encoded_in_one_pass_gives_the_same_step():
    """with the frozen encoder, encode_window() runs ONE encoder pass over all
    micro-batches of an optimiser step.  The features are bit-identical to per-micro-batch encoding (every
    object's feature is independent of the launch it is in), so the accumulated gradients and the updated
    weights are the same (bit-identical up to the arrival order of the schedule's few float atomics)."""
    import msr3d_amd.model  # noqa: F401
    import msr3d_amd.modules  # noqa: F401
    from msr3d_amd import hipops
    from msr3d_amd.config import AttrDict, default_prompter_cfg
    from msr3d_amd.dp import FlatGradAllReduce
    from msr3d_amd.model import build_model
    from msr3d_amd.optim import FlatAdamW
    from msr3d_amd.synth import synth_batch
    from msr3d_amd.train_step import HotPathTrainStep
    accum, outs = 3, []
    for window in (False, True, "slices"):
        torch.manual_seed(0)
        cfg = AttrDict({"prompter": default_prompter_cfg(dropout=0.0), "llm_hidden_size": 256,
                        "model": {"name": "MSR3DHotPath"}})
        model = build_model(cfg).cuda().train()
        params = [p for p in model.parameters() if p.requires_grad]
        dp = FlatGradAllReduce(params, pack_groups=hipops.collect_pack_groups(model))
        opt = FlatAdamW(dp, lr=1e-3, weight_decay=0.0)
        hipops.attach_packed_views(model, dp, opt)
        batches = [synth_batch(300 + i, 2, O=24, P=1024, device="cuda") for i in range(accum)]
        w = torch.linspace(-1, 1, 2 * 24 * 256, device="cuda").view(2, 24, 256)
        step = HotPathTrainStep(model, opt, dp, lambda o: (o["scene_embeds"] * w).mean(), batches[0],
                                use_graph=True, accum_steps=accum)
        step.capture(batches[0])
        feats = []
        for rep in range(2):                       # two optimiser steps
            if window:
                step.encode_window(batches)
            for b in batches:
                step(b)
                feats.append(step.static["obj_embeds"].clone())
        torch.cuda.synchronize()
        outs.append((feats, {k: v.detach().clone() for k, v in model.named_parameters() if v.requires_grad}))
    (f0, p0) = outs[0]
    for f1, p1 in outs[1:]:
        _same_window(f0, p0, f1, p1)


def _same_window(f0, p0, f1, p1):
    for a, b in zip(f0, f1):
        assert torch.equal(a, b)                   # encoder features: bit-identical
    # Two Adam steps from zero moments move every weight by ~lr * sign(g) each: the positional encoders' parameter
    # gradients still meet by float atomics, whose arrival order perturbs the LAST bits of step 1's update, and where a
    # gradient of step 2 is rounding noise around zero that can flip its sign -- bounded by 2 lr per step, and rare
    for k in p0:
        assert torch.allclose(p0[k], p1[k], rtol=0, atol=4.2e-3), k
        assert float(((p0[k] - p1[k]).abs() > 1e-5).float().mean()) < 0.02 or k.endswith("w_ks.bias"), k


@pytest.mark.parametrize("explicit_grad", [False, True])
def test_window_step_equals_the_accumulated_micro_steps(explicit_grad):
    """micro_batches = m: the whole accumulation window in ONE pass (encoder, trainable part and backward once over
    m x B scenes, loss per micro-batch slice scaled 1/m) == m accumulated micro-steps: same gradient buffer
    (fp32 rounding of a different summation order), same weights after the optimiser step, captured graph."""
    import msr3d_amd.model  # noqa: F401
    import msr3d_amd.modules  # noqa: F401
    from msr3d_amd import hipops
    from msr3d_amd.config import AttrDict, default_prompter_cfg
    from msr3d_amd.dp import FlatGradAllReduce
    from msr3d_amd.model import build_model
    from msr3d_amd.optim import FlatAdamW
    from msr3d_amd.synth import synth_batch
    from msr3d_amd.train_step import HotPathTrainStep
    m, B, O, E = 3, 2, 24, 256
    w = torch.linspace(-1, 1, B * O * E, device="cuda").view(B, O, E)
    g = w / w.numel()

    def loss_fn(o):
        y = o["scene_embeds"]
        assert y.shape[0] == B                      # always ONE micro-batch's scenes
        if explicit_grad:
            return (y.detach() * w).mean(), y, g
        return (y * w).mean()

    outs = []
    for window in (False, True):
        torch.manual_seed(0)
        cfg = AttrDict({"prompter": default_prompter_cfg(dropout=0.0), "llm_hidden_size": E,
                        "model": {"name": "MSR3DHotPath"}})
        model = build_model(cfg).cuda().train()
        dp = FlatGradAllReduce([p for p in model.parameters() if p.requires_grad],
                               pack_groups=hipops.collect_pack_groups(model))
        opt = FlatAdamW(dp, lr=1e-3, weight_decay=0.0)
        hipops.attach_packed_views(model, dp, opt)
        micro = [synth_batch(300 + i, B, O=O, P=1024, device="cuda") for i in range(m)]
        if window:
            whole = {k: torch.cat([b[k] for b in micro], 0) for k in micro[0]}
            step = HotPathTrainStep(model, opt, dp, loss_fn, whole, use_graph=True, micro_batches=m)
            step.capture(whole)
            losses = [float(step(whole))]
        else:
            step = HotPathTrainStep(model, opt, dp, loss_fn, micro[0], use_graph=True, accum_steps=m)
            step.capture(micro[0])
            losses = [float(step(b)) for b in micro]
        torch.cuda.synchronize()
        outs.append((sum(losses) / len(losses), dp.flat.detach().clone(),
                     {k: v.detach().clone() for k, v in model.named_parameters() if v.requires_grad}))
    (l0, g0, p0), (l1, g1, p1) = outs
    assert l0 == pytest.approx(l1, rel=1e-5)
    assert float(g0.abs().max()) > 0
    assert float((g0 - g1).norm() / g0.norm()) < 2e-5
    # (the first Adam step moves every weight by ~lr * sign(g): where g is rounding noise around zero the two
    # summation orders may disagree on the sign -- bounded by 2 lr, and rare)
    for k in p0:
        assert torch.allclose(p0[k], p1[k], rtol=0, atol=2.1e-3), k
        assert float(((p0[k] - p1[k]).abs() > 1e-5).float().mean()) < 0.02 or k.endswith("w_ks.bias"), k


def test_window_step_rejects_what_it_cannot_slice():
    import msr3d_amd.model  # noqa: F401
    from msr3d_amd.train_step import HotPathTrainStep
    from msr3d_amd.synth import synth_batch
    b = synth_batch(1, 3, O=4, P=64, device="cuda")
    with pytest.raises(ValueError):
        HotPathTrainStep(None, None, None, None, b, micro_batches=2)           # 3 scenes, 2 micro-batches
    with pytest.raises(ValueError):
        HotPathTrainStep(None, None, None, None, b, micro_batches=3, accum_steps=2)
