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
