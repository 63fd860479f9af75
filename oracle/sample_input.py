"""CPU restatement (numpy) of the reference's scene-sample construction: the row SURVEY.md
§8(f) rank 2 names -- per-scan object segmentation, object selection, `preprocess_pcd`
(rotate, box, subsample, normalise) and the dataset wrapper's padding.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and tools/bench_preprocess.py's cpu leg).
Nothing under msr3d_amd/ may import this.

Pinned by tests/golden/preprocess_seed*.npz, produced by tests/golden/make_golden_preprocess.py,
which EXECUTES the reference's own `MSR3DBase.preprocess_pcd`, `_get_scene_encoder_input` and
`build_rotate_mat` (function bodies compiled from /root/reference at generation time; the
module itself is not importable here: jsonlines / nltk / cv2 / open3d are absent) on seeded
synthetic scans, recording the random draws they made.

Reference lines restated:
  data/datasets/scannet_base.py:57-67     colours u8 -> f64 `c / 127.5 - 1`, `[points, colors]` concat
                                          (=> every later quantity is float64), per-instance masks
  data/datasets/scan_data_loader.py:83-94,191-192   {inst_id: pcds[mask]} dictionaries
  data/datasets/msr3d.py:181-241          preprocess_pcd
  data/datasets/msr3d.py:267-298          _get_scene_encoder_input (selection when > max_obj_len)
  data/datasets/dataset_wrapper.py:141-158   pad_tensors(pad=1.0 / 0.0), obj_masks, `.float()`
  data/data_utils.py:175-189              build_rotate_mat ('axis' angles, float32 matrix)

Not restated: the in-place write `obj_pcd[:, :3] = matmul(...)` at msr3d.py:189-190 mutates the
arrays held by the global scan cache (`.copy()` at :268 is a shallow dict copy), so in the
reference rotations accumulate across samples of the same scan.  That is state leaking between
samples, not part of the per-sample function; oracle and kernels rotate a private copy.
"""
import numpy as np


# ---------------------------------------------------------------- scan -> objects
def scan_to_pcds(points, colors_u8):
    """scannet_base.py:60-62: float64 (N, 6) = [xyz, rgb/127.5 - 1]."""
    colors = colors_u8 / 127.5 - 1
    return np.concatenate([points, colors], 1)


def segment_instances(instance_labels, inst_ids):
    """For every id in `inst_ids` (in that order) the ascending point indices with that label
    (`pcds[instance_labels == i]`, scannet_base.py:65-67 / scan_data_loader.py:91-93).
    Returns (order (sum n_i,) int32, offsets (len+1,) int64)."""
    order, offsets = [], [0]
    for i in inst_ids:
        idx = np.nonzero(instance_labels == i)[0]
        order.append(idx.astype(np.int32))
        offsets.append(offsets[-1] + len(idx))
    order = np.concatenate(order) if order else np.zeros((0,), np.int32)
    return order, np.asarray(offsets, np.int64)


# ---------------------------------------------------------------- rotation draw
ROTATE_ANGLES = [0, np.pi / 2, np.pi, np.pi * 3 / 2]


def rotate_mat(theta):
    """data_utils.py:181-188 (float32 entries; None for theta == 0 / no augmentation)."""
    if theta is None or theta == 0:
        return None
    return np.array([[np.cos(theta), -np.sin(theta), 0],
                     [np.sin(theta), np.cos(theta), 0],
                     [0, 0, 1]], dtype=np.float32)


# ---------------------------------------------------------------- preprocess_pcd
def preprocess_pcd(obj_pcds, pcd_idxs, rot_matrix=None):
    """msr3d.py:181-216 with the random draws passed in.
    obj_pcds: list of float64 (n_i, 6); pcd_idxs: list of int (P,) per object;
    returns obj_fts float64 (n_obj, P, 6), obj_locs float64 (n_obj, 6)."""
    obj_fts, obj_locs = [], []
    for obj_pcd, idxs in zip(obj_pcds, pcd_idxs):
        obj_pcd = np.array(obj_pcd, dtype=np.float64, copy=True)
        if rot_matrix is not None:
            obj_pcd[:, :3] = np.matmul(obj_pcd[:, :3], rot_matrix.transpose())       # :189-190
        obj_center = obj_pcd[:, :3].mean(0)                                           # :192
        obj_size = obj_pcd[:, :3].max(0) - obj_pcd[:, :3].min(0)                      # :193
        obj_locs.append(np.concatenate([obj_center, obj_size], 0))
        obj_pcd = obj_pcd[np.asarray(idxs)]                                           # :200-202
        obj_pcd[:, :3] = obj_pcd[:, :3] - obj_pcd[:, :3].mean(0)                      # :205
        max_dist = np.sqrt((obj_pcd[:, :3] ** 2).sum(1)).max()                        # :206
        if max_dist < 1e-6:                                                           # :207-208
            max_dist = 1
        obj_pcd[:, :3] = obj_pcd[:, :3] / max_dist                                    # :209
        obj_fts.append(obj_pcd)
    return np.stack(obj_fts, 0), np.array(obj_locs)


def rotate_situation(situation, rot_matrix):
    """msr3d.py:228-240: agent position / orientation quaternion (x,y,z,w) under the scene
    rotation."""
    from scipy.spatial.transform import Rotation as R
    if rot_matrix is None:
        return situation
    pos, ori = situation
    pos_new = (np.array(pos).reshape(1, 3) @ rot_matrix.transpose()).reshape(-1)
    ori_new = R.from_matrix(rot_matrix @ R.from_quat(np.array(ori)).as_matrix()).as_quat().reshape(-1)
    return pos_new, ori_new


def pad_sample(obj_fts, obj_locs, max_obj_len):
    """dataset_wrapper.py:141-158: pad objects with 1.0, locations with 0.0, mask = arange < n,
    everything `.float()`."""
    n, P = obj_fts.shape[0], obj_fts.shape[1]
    fts = np.ones((max_obj_len, P, 6), np.float32)
    locs = np.zeros((max_obj_len, 6), np.float32)
    fts[:n] = obj_fts.astype(np.float32)
    locs[:n] = obj_locs.astype(np.float32)
    return fts, locs, np.arange(max_obj_len) < n


# ---------------------------------------------------------------- object selection
def select_objects(inst_ids, scan_insts, max_obj_len, shuffle):
    """msr3d.py:267-294 on instance ids (the reference shuffles the arrays; the permutation
    is the same).  `shuffle(list)` is random.shuffle or a recording stand-in."""
    inst_ids = list(inst_ids)
    if len(inst_ids) <= max_obj_len:
        return inst_ids
    present = set(inst_ids)
    selected = [i for i in scan_insts if i in present]
    if len(selected) >= max_obj_len:
        shuffle(selected)
        return selected[:max_obj_len]
    remained = [i for i in inst_ids if i not in scan_insts]
    shuffle(remained)
    selected += remained[:max_obj_len - len(selected)]
    assert len(selected) == max_obj_len
    return selected


# ---------------------------------------------------------------- device index generator
# The kernels can draw the subsample themselves (no index traffic from the host).  Integer
# work, restated here bit for bit: msr3d_amd/csrc/preprocess.hip `draw_index`.
_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x):
    """'lowbias32' finaliser on uint64 lanes masked to 32 bits."""
    x = x & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def object_key(seed, b, o):
    """Per-(sample, object-slot) 32-bit key from the 64-bit step seed."""
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    k = _mix32(lo ^ np.uint64(0x9E3779B9))
    k = _mix32(k ^ hi)
    k = _mix32(k ^ np.uint64((b * 0x85EBCA6B) & 0xFFFFFFFF))
    k = _mix32(k ^ np.uint64((o * 0xC2B2AE35) & 0xFFFFFFFF))
    return k


def _feistel(x, key, bits):
    """6-round alternating (unbalanced) Feistel permutation of [0, 2^bits): the low `bits - bits//2`
    bits and the high `bits//2` bits take turns being whitened by a hash of the other half."""
    lb = bits // 2
    rb = bits - lb
    lmask, rmask = np.uint64((1 << lb) - 1), np.uint64((1 << rb) - 1)
    left, right = (x >> np.uint64(rb)) & lmask, x & rmask
    for r in range(6):
        c = np.uint64((r * 0x9E3779B1) & 0xFFFFFFFF)
        if r % 2 == 0:
            left = left ^ (_mix32(right ^ key ^ c) & lmask)
        else:
            right = right ^ (_mix32(left ^ key ^ c) & rmask)
    return (left << np.uint64(rb)) | right


def draw_indices(seed, b, o, n, P):
    """The subsample of np.random.choice(n, P, replace=n < P) (msr3d.py:200-201), as the device
    draws it: n >= P -> the first P images of a keyed permutation of [0, n) (Feistel network on
    the enclosing power of two, cycle-walked back into range: distinct by construction);
    n < P -> P independent uniform draws (32-bit multiply-high)."""
    key = object_key(seed, b, o)
    j = np.arange(P, dtype=np.uint64)
    if n < P:
        u = _mix32(_mix32(j ^ key) + np.uint64(0x68E31DA4))
        return ((u * np.uint64(n)) >> np.uint64(32)).astype(np.int32)
    bits = 1
    while (1 << bits) < n:
        bits += 1
    y = _feistel(j, key, bits)
    while True:
        out = y >= np.uint64(n)
        if not out.any():
            break
        y = np.where(out, _feistel(y, key, bits), y)
    return y.astype(np.int32)
