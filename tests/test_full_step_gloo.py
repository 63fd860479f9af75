"""N > 1 path of the FULL step's gradient engine on CPU (gloo, world 2): the joint parameter set of MSR3DFullStep --
prompter + llm_proj + every LoRA pair -- in ONE flat buffer, buckets leaving from the backward hooks in the order
backward produces them, identical averaged gradients on both ranks, unused parameters untouched.

The language-model KERNELS are GPU-only (LoRALinear raises on CPU tensors; tests/test_full_step_gpu.py runs the real
thing with two ranks on the test GPU), so on CPU the consumer of the LoRA matrices is a torch restatement of the same
formula written here -- y = x W^T + (alpha / r) (x A^T) B^T per projection, msr3d.py:103-112 -- chained through the
module's own parameters.  What is under test is the engine (msr3d_amd/dp.py as msr3d_amd/full_step.py drives it)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def build(seed=0):
    import msr3d_amd.model  # noqa: F401
    import msr3d_amd.modules  # noqa: F401
    from msr3d_amd.config import AttrDict, default_prompter_cfg
    from msr3d_amd.model import build_model
    torch.manual_seed(seed)
    cfg = AttrDict({"prompter": default_prompter_cfg(dropout=0.0), "llm_hidden_size": 128,
                    "llm": {"num_layers": 2, "hidden_size": 128, "num_heads": 2, "intermediate_size": 128,
                            "vocab_size": 256, "lora": {"rank": 16, "alpha": 16}},
                    "scene_sp_token": 250, "model": {"name": "MSR3DFullStep"}})
    model = build_model(cfg).train()
    g = torch.Generator().manual_seed(1)          # frozen weights: the same on every rank (a checkpoint); trainable: per seed
    with torch.no_grad():
        for layer in model.llm_model.layers:
            for grp in (layer.self_attn, layer.mlp):
                for m in grp.values():
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) / m.in_features ** 0.5)
                    m.lora_B.weight.copy_(torch.randn(m.lora_B.weight.shape, generator=g) * 0.05)
        model.llm_model.lm_head.weight.copy_(torch.randn(256, 128, generator=g) / 128 ** 0.5)
        model.embed_tokens.copy_(torch.randn(256, 128, generator=g) * 0.5)
    return model


def lora(m, x):
    return F.linear(x, m.weight.float()) + m.scaling * F.linear(F.linear(x, m.lora_A.weight), m.lora_B.weight)


def forward_cpu(model, batch):
    """MSR3DFullStep.forward with the CPU stand-ins: torch scatter, torch LoRA chain, the loss's CPU formulation."""
    from msr3d_amd.llm.losses import seq_mean_cross_entropy
    from msr3d_amd.model.msr3d_full import build_targets
    from msr3d_amd.model.scene_embeds import MSR3DHotPath, scatter_scene_embeds
    d = MSR3DHotPath.forward(model, dict(batch))
    ids = torch.cat([batch["input_ids"], batch["output_ids"]], 1)
    am = torch.cat([batch["attention_mask"], batch["output_mask"]], 1)
    emb = F.embedding(ids, model.embed_tokens).float()
    emb, am = scatter_scene_embeds(emb, am, ids, d["scene_embeds"], d["obj_masks"], model.scene_sp_token)
    x = emb * am.unsqueeze(-1).to(emb.dtype)
    for layer in model.llm_model.layers:
        a, m = layer.self_attn, layer.mlp
        # (token mixing by a causal running mean of v: the answer positions must see the scene tokens)
        steps = torch.arange(1, x.shape[1] + 1, dtype=x.dtype).view(1, -1, 1)
        ctx = torch.cumsum(lora(a["v_proj"], x), 1) / steps
        x = x + 0.1 * torch.tanh(lora(a["o_proj"], lora(a["q_proj"], x) * torch.sigmoid(lora(a["k_proj"], x)) + ctx))
        x = x + 0.1 * lora(m["down_proj"], F.silu(lora(m["gate_proj"], x)) * lora(m["up_proj"], x))
    logits = F.linear(x, model.llm_model.lm_head.weight.float())
    targets = build_targets(batch["input_ids"].shape[1], batch["output_ids"], batch["output_mask"])
    return seq_mean_cross_entropy(logits, targets)


def make_batch(seed, B=2, O=8):
    from msr3d_amd.synth import synth_batch, synth_text
    b = synth_batch(seed, B, O=O, P=16)
    b.pop("obj_fts")
    b["obj_embeds"] = torch.randn(B, O, 768, generator=torch.Generator().manual_seed(seed))   # (frozen encoder's output)
    b.update(synth_text(seed + 1, B, L=O, T_in=40, T_out=24, vocab=250, scene_token=250))
    return b


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from msr3d_amd import hipops
        from msr3d_amd.dp import FlatGradAllReduce
        model = build(seed=rank)                    # rank-dependent init: the engine broadcasts rank 0's
        params = model.get_opt_params()
        eng = FlatGradAllReduce(params, bucket_bytes=64 << 10, overlap=True, pack_groups=hipops.collect_pack_groups(model))
        _, spread = eng.replica_checksum()
        assert spread == 0.0 and eng.world == world and len(eng.buckets) > 8
        # bucket order = backward order: last decoder layer's LoRA first, the prompter last
        assert any(eng.order[0] is p for p in model.llm_model.layers[-1].parameters())
        assert any(eng.order[-1] is p for p in model.visual_prompter.parameters())
        sent = []
        launch = eng._launch

        def spy(b):
            if not eng._launched[b]:
                sent.append(b)
            launch(b)
        eng._launch = spy
        grads = []
        for step in range(2):
            eng.zero_grad()
            eng.begin_micro(last=True)
            loss = forward_cpu(model, make_batch(100 + 10 * rank + step)).mean()
            loss.backward()
            during = list(sent)
            eng.finish()
            grads.append(eng.flat.numpy().copy())
            assert sorted(sent) == list(range(len(eng.buckets))) and len(during) >= len(eng.buckets) // 2, (during, sent)
            assert during[0] == 0 and sent[-1] == len(eng.buckets) - 1
            sent.clear()
        unused = [n for n, p in model.named_parameters() if p.requires_grad and float(p.grad.abs().max()) == 0.0]
        q.put((rank, "ok", grads, unused, [float(loss.detach())]))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:            # noqa: BLE001
        import traceback
        q.put((rank, "error: " + repr(e) + "\n" + traceback.format_exc(), None, None, None))


def test_joint_engine_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, status, grads, unused, loss = q.get(timeout=300)
        assert status == "ok", status
        res[r] = (grads, unused, loss)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for s in range(2):
        assert np.array_equal(res[0][0][s], res[1][0][s]), "ranks hold different averaged gradients"
    assert "visual_prompter.anchor_feat" in res[0][1]            # no gradient in this configuration: stays zero
    # the exchanged buffer is the AVERAGE of the two ranks' local gradients: recompute both locally
    from msr3d_amd import hipops
    from msr3d_amd.dp import FlatGradAllReduce
    model = build(seed=0)
    eng = FlatGradAllReduce(model.get_opt_params(), bucket_bytes=64 << 10, pack_groups=hipops.collect_pack_groups(model))
    acc = np.zeros_like(res[0][0][0])
    for rank in range(2):
        eng.zero_grad()
        forward_cpu(model, make_batch(100 + 10 * rank)).mean().backward()
        acc += eng.flat.numpy()
    got, want = res[0][0][0], acc / 2
    assert np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max()) and np.count_nonzero(want) > 1000
    # every LoRA matrix and llm_proj received a gradient through the language-model side
    off = eng.offset
    for p in model.llm_model.lora_parameters() + list(model.llm_proj.parameters()):
        assert np.abs(want[off[id(p)]:off[id(p)] + p.numel()]).max() > 0
