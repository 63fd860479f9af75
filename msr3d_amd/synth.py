"""Seeded synthetic MSQA scenes in the dataset's conventions.

What the reference's host pipeline hands the model, restated as a generator (there
is no dataset on the box):
  * per object `n_raw ~ U{200..5000}` points in a box of size U(0.1,2.0)^3 m somewhere in an
    8x8x3 m room; subsampled to P points, WITH replacement iff n_raw < P; centred on the mean
    and scaled to max-norm 1 (data/datasets/msr3d.py:199-209); colours U(-1,1)
    (data/datasets/scannet_base.py:61);
  * `n_valid ~ U{20..O}` real objects, the rest padded with the constant 1.0 and mask False,
    padded obj_locs rows 0 (data/datasets/dataset_wrapper.py:155-158);
  * obj_locs = [centre xyz, size xyz] in scene metres; anchor position U(room), anchor
    orientation a yaw quaternion in scipy (x,y,z,w) order (data/data_utils.py:544-552).
Generated with numpy on the host (identical on every machine), then moved to `device`.
"""
import numpy as np
import torch

ROOM = np.array([8.0, 8.0, 3.0])


def synth_scene(rng, O=60, P=1024, n_valid=None, dense=False):
    """dense=True: the WORST case of the distinct-row set-abstraction kernels, not a dataset-like scene -- every slot a
    real object (no padding clouds), P different points inside a ball of radius 0.095 (not rescaled to max-norm 1), so
    that every ball query of both levels (radii 0.2 / 0.4, configs/msr3d.yaml:155) finds at least its 32 samples and
    no neighbourhood row is a copy of another."""
    if n_valid is None:
        n_valid = O if dense else int(rng.integers(min(20, O), O + 1))
    fts = np.ones((O, P, 6), np.float32)
    locs = np.zeros((O, 6), np.float32)
    mask = np.zeros((O,), bool)
    for o in range(n_valid):
        size = rng.uniform(0.1, 2.0, 3)
        centre = rng.uniform(0, 1, 3) * ROOM
        n_raw = int(rng.integers(200, 5001))
        raw = (rng.uniform(-0.5, 0.5, (n_raw, 3)) * size + centre).astype(np.float32)
        sel = rng.choice(n_raw, P, replace=n_raw < P)
        pts = raw[sel]
        box_c = (raw.max(0) + raw.min(0)) / 2
        box_s = raw.max(0) - raw.min(0)
        pts = pts - pts.mean(0)
        pts = pts / max(float(np.sqrt((pts ** 2).sum(1)).max()), 1e-6)
        if dense:
            d = rng.normal(size=(P, 3))
            d /= np.sqrt((d ** 2).sum(1, keepdims=True))
            pts = (d * (0.095 * rng.uniform(0, 1, (P, 1)) ** (1.0 / 3.0))).astype(np.float32)
        fts[o, :, :3] = pts
        fts[o, :, 3:] = rng.uniform(-1, 1, (P, 3))
        locs[o, :3] = box_c
        locs[o, 3:] = box_s
        mask[o] = True
    anchor = (rng.uniform(0, 1, 3) * ROOM).astype(np.float32)
    yaw = rng.uniform(-np.pi, np.pi)
    quat = np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)], np.float32)
    return fts, mask, locs, anchor, quat


def synth_batch(seed, B, O=60, P=1024, n_valid=None, device="cpu", dense=False):
    """-> dict of torch tensors: obj_fts (B,O,P,6) f32, obj_masks (B,O) bool, obj_locs (B,O,6),
    anchor_locs (B,3), anchor_orientation (B,4).  `n_valid`: int or per-sample list.  dense: synth_scene."""
    rng = np.random.default_rng(seed)
    cols = [[], [], [], [], []]
    for i in range(B):
        nv = n_valid[i] if isinstance(n_valid, (list, tuple)) else n_valid
        for c, v in zip(cols, synth_scene(rng, O, P, nv, dense=dense)):
            c.append(v)
    names = ["obj_fts", "obj_masks", "obj_locs", "anchor_locs", "anchor_orientation"]
    return {n: torch.from_numpy(np.stack(c)).to(device) for n, c in zip(names, cols)}


def synth_scan(rng, n_inst, n_points):
    """A scan in the on-disk layout of scan_data/pcd_with_global_alignment/<scan>.pth:
    points f32 (N,3), colors u8 (N,3), instance_labels i64 (N,) with -100 = unlabelled."""
    centres = rng.uniform([-4, -4, 0], [4, 4, 2.5], (n_inst, 3))
    sizes = rng.uniform(0.1, 2.0, (n_inst, 3))
    weights = rng.uniform(0.2, 5.0, n_inst)
    weights[rng.integers(0, n_inst)] = 0.02          # one tiny object (fewer points than P)
    labels = rng.choice(n_inst, size=n_points, p=weights / weights.sum()).astype(np.int64)
    pts = centres[labels] + (rng.random((n_points, 3)) - 0.5) * sizes[labels]
    unl = rng.random(n_points) < 0.1
    labels[unl] = -100
    # one degenerate object: all its points coincide (max_dist < 1e-6 branch)
    deg = labels == 3
    pts[deg] = centres[3]
    return pts.astype(np.float32), rng.integers(0, 256, (n_points, 3)).astype(np.uint8), labels


def synth_text(seed, B, L=60, T_in=544, T_out=32, vocab=32000, scene_token=31495, pad_id=0, device="cpu"):
    """Token side of a synthetic MSQA sample, in the shapes `MSR3D.forward` builds them
    (/root/reference/model/msr3d/msr3d.py:203-206 left-padded prompt ids + mask, :368-376 right-padded answer + eos):
    input_ids / attention_mask (B, T_in) with exactly L scene placeholders per row (contiguous, as
    build_text_prompt writes them, msr3d.py:308-309), output_ids / output_mask (B, T_out).  Random ids stand in for
    text (there is no tokenizer on the box); none of them equals the placeholder id."""
    rng = np.random.default_rng(seed)
    ids = np.full((B, T_in), pad_id, np.int64)
    am = np.zeros((B, T_in), np.int64)
    out = np.full((B, T_out), pad_id, np.int64)
    om = np.zeros((B, T_out), np.int64)

    def words(n):
        w = rng.integers(3, vocab, n)
        w[w == scene_token] = 3
        return w
    for b in range(B):
        n_pad = int(rng.integers(0, max(1, min(40, T_in - L - 8))))
        n = T_in - n_pad                                 # real prompt tokens, left-padded
        before = int(rng.integers(4, n - L - 3))
        row = np.concatenate([[1], words(before - 1), np.full(L, scene_token), words(n - before - L)])
        ids[b, n_pad:] = row
        am[b, n_pad:] = 1
        n_ans = int(rng.integers(3, T_out + 1))
        out[b, :n_ans] = np.concatenate([[1], words(n_ans - 2), [2]])        # bos ... eos
        om[b, :n_ans] = 1
    return {"input_ids": torch.from_numpy(ids).to(device), "attention_mask": torch.from_numpy(am).to(device),
            "output_ids": torch.from_numpy(out).to(device), "output_mask": torch.from_numpy(om).to(device)}
