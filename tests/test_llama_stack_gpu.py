"""The LoRA-Llama stack (msr3d_amd/llm/stack.py: decoder layers + final RMSNorm + frozen head + the per-sequence mean
cross-entropy) against the reference stack's own numbers: tests/golden/llama_stack_seed0.npz was produced by
transformers' LlamaForCausalLM called as /root/reference/model/msr3d/msr3d.py:409-415 calls it (inputs_embeds +
left-padding attention_mask), peft's LoRA formula on all seven projections of both layers, and the loss statement of
msr3d.py:426-441 (tests/golden/make_golden_llama_stack.py).  The HIP stack keeps every activation in bf16 (the
reference runs under bf16 autocast): loss within 1e-2 relative, gradients within 3e-2 rel-L2.  Plus: a training step
of the stack on the flat-buffer data-parallel engine (the LoRA matrices are what a rank exchanges)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _stack(g):
    from msr3d_amd.llm import LoRALlamaStack
    from tests.helpers import llama_stack_weights
    layers, hidden, heads, inter, vocab, r, alpha, B, T = (int(v) for v in g["cfg"])
    w = llama_stack_weights(int(g["seed"]), layers, hidden, inter, r, vocab)
    net = LoRALlamaStack(layers, hidden, heads, inter, vocab, r=r, lora_alpha=alpha, rms_eps=float(g["eps"]),
                         rope_theta=float(g["theta"]), device="cuda")
    with torch.no_grad():
        for i, layer in enumerate(net.layers):
            lw = w["layers"][i]
            for n in NAMES:
                m = (layer.self_attn if n in layer.self_attn else layer.mlp)[n]
                m.load_base_weight(torch.from_numpy(lw[n]).cuda())
                m.lora_A.weight.copy_(torch.from_numpy(lw[n + ".A"]))
                m.lora_B.weight.copy_(torch.from_numpy(lw[n + ".B"]))
            layer.input_layernorm_weight.copy_(torch.from_numpy(lw["ln1"]))
            layer.post_attention_layernorm_weight.copy_(torch.from_numpy(lw["ln2"]))
        net.norm_weight.copy_(torch.from_numpy(w["norm"]))
        net.lm_head.load_weight(torch.from_numpy(w["head"]).cuda())
    return net


def test_stack_loss_and_gradients_match_the_transformers_fixture():
    g = dict(np.load(os.path.join(GOLD, "llama_stack_seed0.npz")))
    net = _stack(g)
    assert all(p.dtype == torch.float32 for p in net.lora_parameters()) and len(net.lora_parameters()) == 2 * 7 * 2
    x = torch.from_numpy(g["x"]).cuda().to(torch.bfloat16).requires_grad_(True)
    keep = torch.from_numpy(g["keep"]).cuda()
    targets = torch.from_numpy(g["targets"]).cuda()
    lg = net(x, attention_mask=keep)
    assert lg.dtype == torch.bfloat16 and lg.shape == (x.shape[0], x.shape[1], int(g["cfg"][4]))
    assert rel(lg[0, 60:64].float(), g["logits_first"]) < 2e-2
    loss = net(x, attention_mask=keep, targets=targets)
    assert loss.shape == (x.shape[0],)
    assert np.allclose(loss.detach().cpu().numpy(), g["loss"], rtol=1e-2)
    loss.backward(torch.from_numpy(g["grad_loss"]).cuda())
    rows = keep.bool()
    assert rel(x.grad.float()[rows], torch.from_numpy(g["dx"]).cuda()[rows]) < 3e-2
    for i, layer in enumerate(net.layers):
        for n in NAMES:
            m = (layer.self_attn if n in layer.self_attn else layer.mlp)[n]
            assert rel(m.lora_A.weight.grad, g[f"dA/{i}/{n}"]) < 3e-2, (i, n)
            assert rel(m.lora_B.weight.grad, g[f"dB/{i}/{n}"]) < 3e-2, (i, n)


def test_head_and_loss_over_the_answer_span_only_are_the_same_loss_and_gradients():
    """supervised_from = p: final norm + head + cross-entropy over positions p - 1 .. T - 1 only, given targets[:, :p] < 0.
    The loss is the same number (the same logit rows enter it); every gradient is the same sum of non-zero terms (the rows
    left out had d logits == 0 exactly)."""
    g = dict(np.load(os.path.join(GOLD, "llama_stack_seed0.npz")))
    keep = torch.from_numpy(g["keep"]).cuda()
    T = keep.shape[1]
    p = T - 40
    targets = torch.from_numpy(g["targets"]).cuda().clone()
    targets[:, :p] = -100
    targets[:, p + 3:p + 30] = torch.randint(0, int(g["cfg"][4]), (targets.shape[0], 27), device="cuda")
    outs = []
    for span in (None, p, 1):
        net = _stack(g)
        x = torch.from_numpy(g["x"]).cuda().to(torch.bfloat16).requires_grad_(True)
        loss = net(x, attention_mask=keep, targets=targets, supervised_from=span)
        loss.sum().backward()
        outs.append((loss.detach(), x.grad.clone(), [q.grad.clone() for q in net.lora_parameters()]))
    (l0, dx0, g0) = outs[0]
    for l1, dx1, g1 in outs[1:]:
        # the same logit rows enter the loss: the same number; the gradients sum the same non-zero terms, the last layer's
        # token-local half over fewer rows (the left-out rows contributed exact zeros, but the 32-token groups of the
        # weight-gradient products fall differently): equal to fp32 rounding of a reordered sum
        assert torch.equal(l0, l1)
        assert rel(dx1.float(), dx0.float()) < 1e-2          # (bf16 storage: the members of an input group add their d input
        #                                                      in bf16, in place above 128 rows, through autograd below)
        for a, b in zip(g0, g1):
            assert rel(b, a) < 2e-2 or float(a.abs().max()) == 0.0      # (downstream of those bf16 sums)


def test_stack_trains_on_the_flat_gradient_engine():
    """The LoRA matrices as ONE flat gradient buffer (what a data-parallel rank all-reduces: msr3d_amd/dp.py) and the
    fused clip + AdamW over it: the loss of a fixed batch goes down, the frozen weights do not move."""
    from msr3d_amd.dp import FlatGradAllReduce
    from msr3d_amd.optim import FlatAdamW
    g = dict(np.load(os.path.join(GOLD, "llama_stack_seed0.npz")))
    net = _stack(g)
    dp = FlatGradAllReduce(net.lora_parameters())
    opt = FlatAdamW(dp, lr=2e-3, weight_decay=0.0, max_grad_norm=1.0)
    x = torch.from_numpy(g["x"]).cuda().to(torch.bfloat16)
    keep = torch.from_numpy(g["keep"]).cuda()
    targets = torch.from_numpy(g["targets"]).cuda()
    head0 = net.lm_head.weight.clone()
    w0 = net.layers[0].mlp["up_proj"].weight.clone()
    losses = []
    for _ in range(6):
        dp.zero_grad()
        loss = net(x, attention_mask=keep, targets=targets).mean()
        loss.backward()
        dp.finish()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0] - 0.05, losses
    assert torch.equal(net.lm_head.weight, head0) and torch.equal(net.layers[0].mlp["up_proj"].weight, w0)
    assert dp.numel == sum(p.numel() for p in net.lora_parameters())


def test_frozen_linear_refuses_cpu_and_follows_weight_writes():
    from msr3d_amd.llm import FrozenLinear
    lin = FrozenLinear(128, 256, device="cuda")
    with torch.no_grad():
        lin.weight.copy_(torch.randn(256, 128, device="cuda"))
    x = torch.randn(64, 128, device="cuda").bfloat16().requires_grad_(True)
    y = lin(x)
    y.backward(torch.ones_like(y))
    assert rel(y.float(), x.float() @ lin.weight.float().T) < 1e-2
    assert rel(x.grad.float(), torch.ones(64, 256, device="cuda") @ lin.weight.float()) < 1e-2
    with pytest.raises(RuntimeError):
        lin(torch.randn(4, 128))


def test_frozen_head_backward_with_few_rows_cuts_the_long_reduction():
    """FrozenLinear's d input at few rows against a long reduction (the head over the answer span) runs as a batch of
    reduction chunks with fp32 partials: against float64 on the same bf16 operands, and against the one-launch product."""
    from msr3d_amd.llm.stack import FrozenLinear
    torch.manual_seed(3)
    K, V = 512, 16000
    head = FrozenLinear(K, V, device="cuda")
    head.load_weight((torch.randn(V, K, device="cuda") / K ** 0.5))
    for M in (260, 1500):                               # 260: the chunked path; 1500: one launch
        x = torch.randn(M, K, device="cuda").bfloat16().requires_grad_(True)
        gy = torch.randn(M, V, device="cuda").bfloat16()
        head(x).backward(gy)
        want = gy.double() @ head.weight.double()
        assert rel(x.grad, want) < 4e-3, M
