"""Scene-encoder inputs for a batch, built on the device.

Host-side mirror of `MSR3DBase._get_scene_encoder_input` + `preprocess_pcd`
(/root/reference/data/datasets/msr3d.py:181-241,267-298), `build_rotate_mat`
(data/data_utils.py:175-189) and the wrapper's padding / masks
(data/datasets/dataset_wrapper.py:151-158): the cheap, branchy parts (which objects, which
rotation, the agent pose) stay in Python with the reference's use of `random`; everything that
touches points is one `msr3d_preprocess_pcd` launch over the HBM-resident SceneStore.
"""
import ctypes
import random

import numpy as np
import torch

from .. import _lib
from .scene_store import _p

ROTATE_ANGLES = [0, np.pi / 2, np.pi, np.pi * 3 / 2]


def build_rotate_mat(split, rot_aug=True, rand_angle="axis"):
    """data_utils.py:175-189: a float32 z-rotation by a random multiple of 90 degrees on the
    training split, else None."""
    if rand_angle == "random":
        theta = np.random.rand() * np.pi * 2
    else:
        theta = random.choice(ROTATE_ANGLES)
    if rot_aug and split == "train" and theta is not None and theta != 0:
        return np.array([[np.cos(theta), -np.sin(theta), 0],
                         [np.sin(theta), np.cos(theta), 0],
                         [0, 0, 1]], dtype=np.float32)
    return None


def _quat_to_matrix(q):
    """Rotation matrix of a quaternion (x, y, z, w), normalised first (what
    scipy.spatial.transform.Rotation.from_quat(q).as_matrix() returns)."""
    x, y, z, w = np.asarray(q, np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _matrix_to_quat(m):
    """Quaternion (x, y, z, w) of a rotation matrix with the branch / sign convention of
    scipy's Rotation.from_matrix(m).as_quat() (Markley's method: the largest of the diagonal
    entries and the trace picks the component that is computed directly and comes out positive),
    which the reference uses at msr3d.py:235-238 -- the sign matters downstream, the quaternion's
    Fourier features feed `orientation_encoder`.  tests/test_sample_input_cpu.py checks it
    against scipy."""
    m = np.asarray(m, np.float64)
    trace = m[0, 0] + m[1, 1] + m[2, 2]
    decision = [m[0, 0], m[1, 1], m[2, 2], trace]
    choice = int(np.argmax(decision))
    q = np.empty(4)
    if choice != 3:
        i = choice
        j = (i + 1) % 3
        k = (j + 1) % 3
        q[i] = 1 - trace + 2 * m[i, i]
        q[j] = m[j, i] + m[i, j]
        q[k] = m[k, i] + m[i, k]
        q[3] = m[k, j] - m[j, k]
    else:
        q[0] = m[2, 1] - m[1, 2]
        q[1] = m[0, 2] - m[2, 0]
        q[2] = m[1, 0] - m[0, 1]
        q[3] = 1 + trace
    return q / np.linalg.norm(q)


def rotate_situation(situation, rot_matrix):
    """msr3d.py:224-240: agent position and orientation quaternion (x, y, z, w) follow the scene
    rotation."""
    if rot_matrix is None:
        return situation
    pos, ori = situation
    pos_new = (np.array(pos).reshape(1, 3) @ rot_matrix.transpose()).reshape(-1)
    ori_new = _matrix_to_quat(rot_matrix @ _quat_to_matrix(np.array(ori)))
    return pos_new, ori_new


class SceneInputBuilder:
    def __init__(self, store, max_obj_len=60, num_points=1024, split="train", use_rotate=True, seed=0):
        self.store = store
        self.max_obj_len = int(max_obj_len)
        self.num_points = int(num_points)
        self.split = split
        self.use_rotate = use_rotate
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.step = 0

    # ------------------------------------------------------------------ msr3d.py:267-294
    def select_objects(self, scan_id, scan_insts):
        inst_ids = list(self.store.inst_ids(scan_id))
        if len(inst_ids) <= self.max_obj_len:
            return inst_ids
        present = set(inst_ids)
        selected = [i for i in scan_insts if i in present]
        if len(selected) >= self.max_obj_len:
            random.shuffle(selected)
            return selected[:self.max_obj_len]
        remained = [i for i in inst_ids if i not in scan_insts]
        random.shuffle(remained)
        selected += remained[:self.max_obj_len - len(selected)]
        assert len(selected) == self.max_obj_len
        return selected

    # ------------------------------------------------------------------
    def build(self, samples, pcd_idxs=None, rot_matrices=None, selections=None, out=None,
              return_indices=False):
        """samples: sequence of dicts with 'scan_id', 'insts' (the instance ids the question is
        about, msr3d.py:423) and optionally 'situation' = (position (3,), quaternion (4,)).
        pcd_idxs (B,O,P) int32 / rot_matrices (list of 3x3 or None) / selections (list of id
        lists) override the random draws (parity tests).  Returns a dict with the keys the
        collated batch of the reference carries: obj_fts (B,O,P,6), obj_locs (B,O,6), obj_masks
        (B,O) bool, anchor_locs (B,3), anchor_orientation (B,4)."""
        lib = _lib.load()
        st, dev = self.store, self.store.device
        B, O, P = len(samples), self.max_obj_len, self.num_points
        begin = np.zeros((B, O), np.int64)
        count = np.zeros((B, O), np.int32)
        rot = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (B, 1))
        any_rot = False
        anchor_locs = np.zeros((B, 3), np.float32)
        anchor_ori = np.zeros((B, 4), np.float32)
        anchor_ori[:, 3] = 1.0                                     # check_output_and_fill_dummy, msr3d.py:112-114
        for b, smp in enumerate(samples):
            scan = st.scans[smp["scan_id"]]
            sel = selections[b] if selections is not None else self.select_objects(smp["scan_id"],
                                                                                   smp.get("insts", []))
            if len(sel) > O:
                raise ValueError("more objects selected than max_obj_len")
            k = len(sel)
            begin[b, :k] = [scan["begin"][i] for i in sel]
            count[b, :k] = [scan["count"][i] for i in sel]
            m = rot_matrices[b] if rot_matrices is not None else build_rotate_mat(self.split,
                                                                                  rot_aug=self.use_rotate)
            if m is not None:
                rot[b] = np.asarray(m, np.float32).reshape(9)
                any_rot = True
            if smp.get("situation") is not None:
                pos, ori = rotate_situation(smp["situation"], m)
                anchor_locs[b] = np.asarray(pos, np.float32)
                anchor_ori[b] = np.asarray(ori, np.float32)

        if out is None:
            out = {
                "obj_fts": torch.empty((B, O, P, 6), dtype=torch.float32, device=dev),
                "obj_locs": torch.empty((B, O, 6), dtype=torch.float32, device=dev),
                "obj_masks": torch.empty((B, O), dtype=torch.bool, device=dev),
                "anchor_locs": torch.empty((B, 3), dtype=torch.float32, device=dev),
                "anchor_orientation": torch.empty((B, 4), dtype=torch.float32, device=dev),
            }
        d_begin = torch.from_numpy(begin).to(dev, non_blocking=True)
        d_count = torch.from_numpy(count).to(dev, non_blocking=True)
        d_rot = torch.from_numpy(rot).to(dev, non_blocking=True) if any_rot else None
        d_idx = None
        if pcd_idxs is not None:
            d_idx = torch.as_tensor(np.ascontiguousarray(pcd_idxs, dtype=np.int32)).to(dev)
            if tuple(d_idx.shape) != (B, O, P):
                raise ValueError("pcd_idxs must be (B, max_obj_len, num_points)")
        idx_out = torch.empty((B, O, P), dtype=torch.int32, device=dev) if return_indices else None
        out["anchor_locs"].copy_(torch.from_numpy(anchor_locs), non_blocking=True)
        out["anchor_orientation"].copy_(torch.from_numpy(anchor_ori), non_blocking=True)
        seed = (self.seed + 0x9E3779B97F4A7C15 * self.step) & 0xFFFFFFFFFFFFFFFF
        self.step += 1
        with torch.cuda.device(dev):
            rc = lib.msr3d_preprocess_pcd(B, O, P, _p(st.points), _p(st.colors), _p(d_begin), _p(d_count),
                                          _p(d_rot), _p(d_idx), ctypes.c_ulonglong(seed), _p(out["obj_fts"]),
                                          _p(out["obj_locs"]), _p(out["obj_masks"]), _p(idx_out),
                                          _lib.current_stream_ptr(dev))
        _lib.check(rc, "msr3d_preprocess_pcd")
        out["last_seed"] = seed
        if return_indices:
            out["pcd_idxs"] = idx_out
        return out
