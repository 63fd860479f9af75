"""Debug aid: run the trainable-part schedule in 'strips' (round 2) and 'blocks' (round 3) mode on the same
inputs and print the relative difference of every intermediate buffer both modes produce."""
import sys

import torch

sys.path.insert(0, ".")
from msr3d_amd import fused_model, hipops  # noqa: E402
from tests.test_fused_model_gpu import _setup  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def run(model, dp, batch, mode):
    fused_model.set_mode(mode)
    model._schedule.enabled = True
    seed = hipops.seed_word(torch.device("cuda", torch.cuda.current_device()))
    seed.fill_(12345)
    hipops._salt_counter[0] = 500
    dp.zero_grad()
    out = model(dict(batch))
    y = out["scene_embeds"]
    w = torch.linspace(-1, 1, y.numel(), device="cuda").view_as(y)
    (y * w).sum().backward()
    dp.finish()
    torch.cuda.synchronize()
    a = model._schedule.arena
    bufs = {k: v.detach().clone() for k, v in a.views.items()}
    grads = {k: v.grad.detach().clone() for k, v in model.named_parameters() if v.requires_grad}
    return bufs, grads


def main():
    dropout = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
    B, O = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3, 20)
    E = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    model, dp, batch = _setup(dropout, B=B, O=O, E=E)
    sb, sg = run(model, dp, batch, "strips")
    bb, bg = run(model, dp, batch, "blocks")
    alias = {"fc": "fcacc", "d_xin": "d_xacc"}
    nl = 3
    order = ["x0", "pos"]
    for i in range(nl):
        order += [f"xin{i}", f"qkvc{i}", f"probs{i}", f"ctx{i}", f"fc{i}", f"s1_{i}", f"s2_{i}", f"t{i}", f"pre{i}",
                  f"h{i}", f"ffn{i}"]
    order += ["tok", "scene"]
    print("--- forward / last-layer backward buffers (strips vs blocks)")
    for k in order:
        kb = k
        for a_, b_ in alias.items():
            if k.startswith(a_) and k[len(a_):].isdigit():
                kb = b_ + k[len(a_):]
        if k in sb and kb in bb:
            print(f"{k:10s} {rel(bb[kb], sb[k]):.3e}   |ref| {float(sb[k].norm()):.3e}")
    # strips mode reuses its backward temporaries: only layer 0's survive
    for k, kb in (("d_ffn", "d_ffn0"), ("d_pre", "d_pre0"), ("d_t", "d_t0"), ("d_fc", "d_fc0"), ("d_qkvc", "d_qkvc0"),
                  ("d_xin2", "d_xacc2"), ("d_xin1", "d_xacc1"), ("d_xin0", "d_xacc0"), ("d_la", "d_la"), ("d_lb", "d_lb")):
        print(f"{k:10s} {rel(bb[kb], sb[k]):.3e}   |ref| {float(sb[k].norm()):.3e}")
    print("--- parameter gradients")
    worst = 0.0
    for k in sg:
        n = float(sg[k].norm())
        r = rel(bg[k], sg[k]) if n > 0 else float(bg[k].abs().max())
        if k.endswith("w_ks.bias"):
            continue
        worst = max(worst, r)
        if r > 2e-5:
            print(f"{k:60s} {r:.3e}  |ref| {n:.3e}")
    print("worst gradient rel", worst)


if __name__ == "__main__":
    main()
