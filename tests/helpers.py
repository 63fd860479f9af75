"""Shared test helpers: golden loading, deterministic weights (same generator as
tests/golden/make_golden.py), model construction."""
import importlib.util
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
_mg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mg)
fill_state_dict = _mg.fill_state_dict

VARIANTS = {"transform": "as_transform_for_objects", "anchor": "as_object"}


def load_golden(variant, seed):
    return dict(np.load(os.path.join(GOLDEN, f"prompter_{variant}_seed{seed}.npz"), allow_pickle=False))


def build_prompter(variant, seed, device="cpu"):
    import msr3d_amd.model  # noqa: F401
    import msr3d_amd.modules  # noqa: F401
    from msr3d_amd.config import default_prompter_cfg
    from msr3d_amd.model import build_model
    model = build_model(default_prompter_cfg(situation_type=VARIANTS[variant], freeze=True)).eval()
    model.load_state_dict(fill_state_dict(model.state_dict(), seed), strict=True)
    return model.to(device)


def golden_inputs(g, device="cpu"):
    keys = ["obj_fts", "obj_masks", "obj_locs", "anchor_locs", "anchor_orientation"]
    return {k: torch.from_numpy(g[k]).to(device) for k in keys}


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _bf16_round(a):
    """fp32 numpy array rounded to bf16 (nearest even), returned as fp32."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16
    return r.view(np.float32).reshape(a.shape)


def llama_layer_weights(seed, hidden, inter, r):
    """Deterministic bf16-representable weights of one LoRA-Llama decoder layer (numpy generator: stable across
    torch versions), shared by tests/golden/make_golden_llama_layer.py and tests/test_llama_layer_gpu.py."""
    rng = np.random.default_rng(seed)
    shapes = {"q_proj": (hidden, hidden), "k_proj": (hidden, hidden), "v_proj": (hidden, hidden),
              "o_proj": (hidden, hidden), "gate_proj": (inter, hidden), "up_proj": (inter, hidden),
              "down_proj": (hidden, inter)}
    w = {}
    for n, (o, i) in shapes.items():
        w[n] = _bf16_round(rng.standard_normal((o, i)).astype(np.float32) / np.sqrt(i))
        w[n + ".A"] = _bf16_round(rng.standard_normal((r, i)).astype(np.float32) / np.sqrt(i))
        w[n + ".B"] = _bf16_round(rng.standard_normal((o, r)).astype(np.float32) * 0.05)
    w["ln1"] = _bf16_round(1.0 + 0.1 * rng.standard_normal(hidden).astype(np.float32))
    w["ln2"] = _bf16_round(1.0 + 0.1 * rng.standard_normal(hidden).astype(np.float32))
    return w


def llama_stack_weights(seed, layers, hidden, inter, r, vocab):
    """Weights of a LoRA-Llama stack: per-layer sets (llama_layer_weights, seeds seed + 10 i), the final norm and the
    head; shared by tests/golden/make_golden_llama_stack.py and tests/test_llama_stack_gpu.py."""
    rng = np.random.default_rng(seed + 7777)
    return {"layers": [llama_layer_weights(seed + 10 * i, hidden, inter, r) for i in range(layers)],
            "norm": _bf16_round(1.0 + 0.1 * rng.standard_normal(hidden).astype(np.float32)),
            "head": _bf16_round(rng.standard_normal((vocab, hidden)).astype(np.float32) / np.sqrt(hidden))}
