"""The sample-construction oracle (oracle/sample_input.py) against the vectors produced by
running the reference's own preprocess_pcd / _get_scene_encoder_input / build_rotate_mat
(tests/golden/make_golden_preprocess.py), plus the integer properties of the device index
generator's restatement."""
import os

import numpy as np
import pytest

from oracle import sample_input as si

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEEDS = [0, 1, 2]


def load(seed):
    return dict(np.load(os.path.join(GOLDEN, f"preprocess_seed{seed}.npz"), allow_pickle=False))


def replay_shuffle(perm):
    def shuffle(x):
        assert len(x) == len(perm)
        x[:] = [x[t] for t in perm]
    return shuffle


def oracle_sample(g):
    pcds = si.scan_to_pcds(g["points"], g["colors"])
    order, offsets = si.segment_instances(g["instance_labels"], g["inst_ids"])
    sel = si.select_objects(g["inst_ids"].tolist(), g["scan_insts"].tolist(), int(g["max_obj_len"]),
                            replay_shuffle(g["shuffle_perm"].tolist()))
    slot = {int(i): k for k, i in enumerate(g["inst_ids"])}
    objs = [pcds[order[offsets[slot[i]]:offsets[slot[i] + 1]]] for i in sel]
    rot = None if g["rot_is_none"] else g["rot_matrix"]
    fts, locs = si.preprocess_pcd(objs, list(g["pcd_idxs"]), rot)
    return sel, fts, locs, rot


@pytest.mark.parametrize("seed", SEEDS)
def test_oracle_matches_reference_run(seed):
    g = load(seed)
    sel, fts, locs, rot = oracle_sample(g)
    assert len(sel) == int(g["obj_masks"].sum())
    assert np.array_equal(locs, g["obj_locs_f64"])               # same numpy formulas: bit-exact
    pf, pl, pm = si.pad_sample(fts, locs, int(g["max_obj_len"]))
    assert np.array_equal(pf, g["obj_fts"])
    assert np.array_equal(pl, g["obj_locs"])
    assert np.array_equal(pm, g["obj_masks"])
    pos, ori = si.rotate_situation((g["situation_pos"], g["situation_ori"]), rot)
    assert np.allclose(pos, g["situation_pos_out"], rtol=0, atol=1e-12)
    assert np.allclose(ori, g["situation_ori_out"], rtol=0, atol=1e-12)


def test_rotate_mat_matches_recorded_matrix():
    hits = 0
    for seed in SEEDS:
        g = load(seed)
        if g["rot_is_none"]:
            continue
        assert any(np.array_equal(si.rotate_mat(t), g["rot_matrix"]) for t in si.ROTATE_ANGLES[1:])
        hits += 1
    assert hits >= 1 and si.rotate_mat(0) is None


def test_degenerate_object_keeps_unit_scale():
    # all points coincide: max_dist < 1e-6 -> divide by 1 (msr3d.py:207-208)
    obj = np.tile(np.array([[1.5, -2.0, 0.25, 0.1, 0.2, 0.3]]), (7, 1))
    fts, locs = si.preprocess_pcd([obj], [np.arange(16) % 7])
    assert np.array_equal(fts[0, :, :3], np.zeros((16, 3)))
    assert np.array_equal(locs[0], [1.5, -2.0, 0.25, 0, 0, 0])


def test_select_objects_paths():
    ids = list(range(100, 110))
    assert si.select_objects(ids, [105], 60, None) == ids                      # under the cap: all, in order
    rev = lambda x: x.reverse()                                                # noqa: E731
    got = si.select_objects(ids, [103, 999, 108], 4, rev)                      # relevant first, then shuffled rest
    assert got[:2] == [103, 108] and got[2:] == [109, 107]
    got = si.select_objects(ids, [101, 102, 103, 104, 105], 3, rev)            # too many relevant: shuffle + cut
    assert got == [105, 104, 103]


@pytest.mark.parametrize("n,P", [(1024, 1024), (1025, 1024), (5000, 1024), (70000, 1024), (300000, 2048),
                                 (17, 16), (4, 4)])
def test_device_draw_without_replacement_is_a_partial_permutation(n, P):
    idx = si.draw_indices(0x1234567890ABCDEF, 3, 7, n, P)
    assert idx.shape == (P,) and idx.dtype == np.int32
    assert idx.min() >= 0 and idx.max() < n
    assert len(np.unique(idx)) == P
    other = si.draw_indices(0x1234567890ABCDEF, 3, 8, n, P)
    assert not np.array_equal(idx, other)                                       # keyed per object slot


def test_device_draw_full_permutation_and_uniformity():
    n = 1000
    perm = si.draw_indices(42, 0, 0, n, n)
    assert np.array_equal(np.sort(perm), np.arange(n))
    # with replacement (n < P): all in range, roughly uniform
    idx = si.draw_indices(42, 1, 2, 50, 100000)
    assert idx.min() == 0 and idx.max() == 49
    counts = np.bincount(idx, minlength=50)
    assert abs(counts - 2000).max() < 250
    # without replacement: every element equally likely to be picked (over many keys)
    hits = np.zeros(64)
    for o in range(400):
        hits[si.draw_indices(7, 0, o, 64, 16)] += 1
    assert abs(hits - 100).max() < 45


def test_host_mirror_situation_rotation_matches_scipy_and_the_reference_run():
    """msr3d_amd.data.rotate_situation restates scipy's from_quat/as_matrix/from_matrix/as_quat chain
    (msr3d.py:228-240) in numpy, signs included."""
    from scipy.spatial.transform import Rotation as R
    from msr3d_amd.data.scene_input import ROTATE_ANGLES, _matrix_to_quat, _quat_to_matrix, rotate_situation
    rng = np.random.default_rng(0)
    for _ in range(300):
        q = rng.standard_normal(4)
        if rng.random() < 0.3:                      # the dataset's orientations are yaw-only
            q[:2] = 0
        for th in ROTATE_ANGLES[1:]:
            m = si.rotate_mat(th)
            assert np.allclose(_quat_to_matrix(q), R.from_quat(q).as_matrix(), atol=1e-14)
            want = R.from_matrix(m @ R.from_quat(q).as_matrix()).as_quat()
            got = _matrix_to_quat(m @ _quat_to_matrix(q))
            assert np.allclose(got, want, atol=1e-12), (q, th)
    for seed in SEEDS:
        g = load(seed)
        rot = None if g["rot_is_none"] else g["rot_matrix"]
        pos, ori = rotate_situation((g["situation_pos"], g["situation_ori"]), rot)
        assert np.allclose(pos, g["situation_pos_out"], atol=1e-12)
        assert np.allclose(ori, g["situation_ori_out"], atol=1e-12)


def test_host_mirror_build_rotate_mat_draws_like_the_reference():
    import random
    from msr3d_amd.data import build_rotate_mat
    random.seed(5)
    thetas = [random.choice(si.ROTATE_ANGLES) for _ in range(20)]
    random.seed(5)
    for th in thetas:
        m = build_rotate_mat("train")
        want = si.rotate_mat(th)
        assert (m is None and want is None) or np.array_equal(m, want)
    assert build_rotate_mat("val") is None and build_rotate_mat("train", rot_aug=False) is None
