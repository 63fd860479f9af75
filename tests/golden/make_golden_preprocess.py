"""Golden vectors for the sample-construction row, made by RUNNING the
reference's own functions in the build container:

    python tests/golden/make_golden_preprocess.py         (needs /root/reference; CPU only)

`data/datasets/msr3d.py` cannot be imported here (jsonlines, nltk, cv2, open3d ... are absent),
so the three functions on this row -- `MSR3DBase.preprocess_pcd`, `MSR3DBase._get_scene_encoder_input`
(data/datasets/msr3d.py:181-241, 267-298) and `build_rotate_mat` (data/data_utils.py:175-189) --
are compiled from the reference's source files where they lie (ast -> code object; nothing is
written to the repo) and executed against seeded synthetic scans.  The random draws they make
(`np.random.choice`, `random.choice`, `random.shuffle`) are recorded so the oracle and the HIP
kernels can be fed the same ones.  The dataset wrapper's padding (dataset_wrapper.py:141-158)
is applied with the reference's formulas on torch tensors, as there.

Output: tests/golden/preprocess_seed{0,1,2}.npz (inputs + expected outputs; data only).
"""
import ast
import os
import random
import sys
import types

import numpy as np
import torch
from scipy.spatial.transform import Rotation as R

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from msr3d_amd.synth import synth_scan  # noqa: E402  (seeded synthetic scan in the on-disk layout)


def _extract(path, cls, names):
    tree = ast.parse(open(path).read())
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    fns = [n for n in body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(fns) == len(names), (path, names)
    for f in fns:
        f.decorator_list = []
    mod = ast.Module(body=fns, type_ignores=[])
    return compile(ast.fix_missing_locations(mod), path, "exec")


def reference_functions():
    ns = {"np": np, "torch": torch, "random": random, "R": R}
    exec(_extract(os.path.join(REF, "data/data_utils.py"), None, ["build_rotate_mat"]), ns)
    exec(_extract(os.path.join(REF, "data/datasets/msr3d.py"), "MSR3DBase",
                  ["preprocess_pcd", "_get_scene_encoder_input"]), ns)
    return ns


def scan_obj_pcds(points, colors, labels):
    """scannet_base.py:57-67 + scan_data_loader.py:191-192, the reference's formulas."""
    colors = colors / 127.5 - 1
    pcds = np.concatenate([points, colors], 1)
    obj_pcds = []
    for i in range(labels.max() + 1):
        obj_pcds.append(pcds[labels == i])
    return {idx: obj_pcds[idx] for idx in range(len(obj_pcds))}


def pad_tensors(tensors, lens=None, pad=0):          # dataset_wrapper.py:141-149 semantics
    if tensors.shape[0] == lens:
        return tensors
    shape = list(tensors.shape)
    shape[0] = lens - shape[0]
    return torch.cat((tensors, torch.ones(shape, dtype=tensors.dtype) * pad), dim=0)


def make(seed, n_inst, n_points, max_obj_len, num_points, split="train"):
    rng = np.random.default_rng(seed)
    points, colors, labels = synth_scan(rng, n_inst, n_points)
    obj_pcds = scan_obj_pcds(points, colors, labels)
    obj_pcds = {k: v for k, v in obj_pcds.items() if len(v) > 0}      # a real scan has no empty instance
    ns = reference_functions()

    rec = {"choice": [], "rot": [], "shuffle_in": [], "shuffle_out": []}
    real_choice, real_shuffle, real_brm = np.random.choice, random.shuffle, ns["build_rotate_mat"]

    def rec_choice(a, size=None, replace=True, p=None):
        out = real_choice(a, size=size, replace=replace, p=p)
        rec["choice"].append(np.asarray(out).copy())
        return out

    def rec_brm(*a, **k):
        m = real_brm(*a, **k)
        rec["rot"].append(None if m is None else m.copy())
        return m

    def rec_shuffle(x):
        # the reference shuffles lists of arrays or of ids; record the permutation
        tagged = list(range(len(x)))
        real_shuffle(tagged)
        x[:] = [x[t] for t in tagged]
        rec["shuffle_out"].append(np.asarray(tagged))

    np.random.seed(seed)
    random.seed(seed + 1)
    fake_random = types.SimpleNamespace(choice=random.choice, shuffle=rec_shuffle)
    ns["random"] = fake_random
    ns["build_rotate_mat"] = rec_brm
    np.random.choice = rec_choice
    try:
        theta_probe = None
        self = types.SimpleNamespace(split=split, num_points=num_points, max_obj_len=max_obj_len,
                                     use_rotate=True)
        self.preprocess_pcd = types.MethodType(ns["preprocess_pcd"], self)
        scan_insts = [int(i) for i in rng.choice(sorted(obj_pcds), size=5, replace=False)]
        situation = (rng.uniform(-3, 3, 3), R.from_euler("xyz", [0, 0, rng.uniform(0, 6.28)]).as_quat())
        # private copies: the reference rotates the cached arrays in place (msr3d.py:189-190)
        scan_data = {"obj_pcds": {k: v.copy() for k, v in obj_pcds.items()}}
        out = ns["_get_scene_encoder_input"](self, scan_data, scan_insts, situation=situation)
    finally:
        np.random.choice = real_choice

    obj_fts = pad_tensors(out["obj_fts"], lens=max_obj_len, pad=1.0).float()
    obj_masks = torch.arange(max_obj_len) < len(out["obj_locs"])
    obj_locs = pad_tensors(out["obj_locs"], lens=max_obj_len, pad=0.0).float()
    rot = rec["rot"][0]
    g = {
        "points": points, "colors": colors, "instance_labels": labels,
        "inst_ids": np.asarray(sorted(obj_pcds), np.int64), "scan_insts": np.asarray(scan_insts, np.int64),
        "max_obj_len": np.int64(max_obj_len), "num_points": np.int64(num_points),
        "rot_is_none": np.bool_(rot is None),
        "rot_matrix": np.eye(3, dtype=np.float32) if rot is None else rot,
        "pcd_idxs": np.stack(rec["choice"]).astype(np.int32),
        "shuffle_perm": rec["shuffle_out"][0] if rec["shuffle_out"] else np.zeros((0,), np.int64),
        "situation_pos": situation[0], "situation_ori": situation[1],
        "situation_pos_out": np.asarray(out["situation"][0]), "situation_ori_out": np.asarray(out["situation"][1]),
        "obj_fts": obj_fts.numpy(), "obj_locs": obj_locs.numpy(), "obj_masks": obj_masks.numpy(),
        "obj_locs_f64": out["obj_locs"].numpy(),
    }
    return g


if __name__ == "__main__":
    # (seed, instances, points, max_obj_len, P): fewer objects than the cap; more (selection +
    # shuffle path); small P for a compact third fixture with many with-replacement draws
    # the third one on the 'val' split: no rotation (build_rotate_mat returns None)
    for seed, n_inst, n_points, cap, P, split in [(0, 23, 30000, 60, 1024, "train"),
                                                  (1, 75, 40000, 60, 1024, "train"),
                                                  (2, 40, 6000, 16, 256, "val")]:
        g = make(seed, n_inst, n_points, cap, P, split)
        path = os.path.join(HERE, f"preprocess_seed{seed}.npz")
        np.savez_compressed(path, **g)
        print(path, {k: v.shape for k, v in g.items() if hasattr(v, "shape") and v.ndim},
              "rot none" if g["rot_is_none"] else "rotated", os.path.getsize(path) // 1024, "KiB")
