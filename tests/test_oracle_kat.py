"""Known-answer tests that pin oracle/pn2_oracle.c.

The reference's native ops cannot run here (CUDA only), and the reference ships
exactly one test input for them (pointnet2_test.py:25-27).  These cases follow
from the kernel text; each cites the lines it exercises.
"""
import numpy as np
import pytest

from oracle import pn2


def test_opt_n_threads_powers_of_two():
    # cuda_utils.h:13-19
    for n, want in [(1, 1), (2, 2), (3, 2), (16, 16), (32, 32), (33, 32), (1000, 512),
                    (1024, 512), (2048, 512), (511, 256), (64, 64), (100, 64)]:
        assert pn2.opt_n_threads(n) == want, n
    assert pn2.opt_block_config(32, 3) == (32, 2)
    assert pn2.opt_block_config(16, 128) == (16, 32)
    assert pn2.opt_block_config(1024, 64) == (512, 1)


def test_fps_all_ones_padding_object():
    # dataset_wrapper.py:156 pads with 1.0: every distance is 0, 0 > -1 only for k=0's
    # thread first -> tree keeps idx of tid 0 on ties -> all zeros
    xyz = np.ones((2, 1024, 3), np.float32)
    assert (pn2.furthest_point_sampling(xyz, 32) == 0).all()


def test_fps_all_points_skipped():
    # sampling_gpu.cu:100-101: |p|^2 <= 1e-3 never selected; best=-1,besti=0 everywhere -> 0
    rng = np.random.default_rng(0)
    xyz = (rng.standard_normal((3, 256, 3)) * 1e-3).astype(np.float32)
    assert (pn2.furthest_point_sampling(xyz, 16) == 0).all()


@pytest.mark.parametrize("n,a,b,want", [
    (1024, 5, 517, 5),      # same thread (tid 5), lower k wins (strict >)
    (1024, 3, 514, 514),    # tid 3 vs 2: lowest differing bit 0, tid 2 has it clear
    (1024, 1, 2, 2),
    (1024, 1, 256, 256),
    (32, 1, 2, 2),
])
def test_fps_tie_break(n, a, b, want):
    # two points at the same (maximal) distance from point 0; everything else at point 0's
    # position.  sampling_gpu.cu:59-65 keeps idx1 on ties in the halving tree.
    xyz = np.full((1, n, 3), 0.5, np.float32)
    xyz[0, a] = (0.5, 0.5, 0.9)
    xyz[0, b] = (0.5, 0.5, 0.9)
    idx = pn2.furthest_point_sampling(xyz, 2)
    assert idx[0, 0] == 0 and idx[0, 1] == want


def _exact_mag_point(target):
    """(x, y, z) f32 with fma(z,z, fma(x,x, y*y)) == target exactly.  x, z are powers of two
    whose squares are whole multiples of ulp(target), so every step is exact."""
    ulp = np.float32(2.0 ** -33)            # ulp of floats in [2^-10, 2^-9)
    assert np.nextafter(target, np.float32(1)) - target == ulp
    y0 = np.float32(np.sqrt(np.float64(target)))
    cands = [y0]
    lo = hi = y0
    for _ in range(64):
        lo = np.nextafter(lo, np.float32(0)); hi = np.nextafter(hi, np.float32(1))
        cands += [lo, hi]
    for x in (np.float32(0), np.float32(2.0 ** -16), np.float32(2.0 ** -15)):
        for z in (np.float32(0), np.float32(2.0 ** -16), np.float32(2.0 ** -15)):
            for y in cands:
                yy = np.float32(y * y)
                if np.float64(yy) + np.float64(x) * x + np.float64(z) * z == np.float64(target):
                    return x, y, z
    raise AssertionError("no exact representation found")


def test_fps_skip_threshold_is_double_compare():
    # sampling_gpu.cu:100-101 `mag <= 1e-3` compares in double: float(1e-3f) =
    # 0.001000000047... is NOT skipped, the next float below is.
    t = np.float32(1e-3)
    below = np.nextafter(t, np.float32(0))
    for mag, skipped in [(t, False), (below, True)]:
        x, y, z = _exact_mag_point(mag)
        xyz = np.zeros((1, 4, 3), np.float32)
        xyz[0, 0] = (0.5, 0.0, 0.0)
        xyz[0, 1] = (0.5, 0.0, 0.0)
        xyz[0, 2] = (x, y, z)
        xyz[0, 3] = (0.5, 0.0, 0.0)
        idx = pn2.furthest_point_sampling(xyz, 2)
        assert idx[0, 1] == (0 if skipped else 2), (mag, skipped)


def test_ball_query_zero_and_single_hit():
    # ball_query_gpu.cu:30-39: no hit -> zeros (host zero-init); first hit pre-fills the row
    xyz = np.zeros((1, 8, 3), np.float32)
    xyz[0, :, 0] = np.arange(8) * 10.0
    new_xyz = np.array([[[1000.0, 0, 0], [30.0, 0, 0], [0.05, 0, 0]]], np.float32)
    idx = pn2.ball_query(new_xyz, xyz, 0.2, 4)
    assert (idx[0, 0] == 0).all()
    assert (idx[0, 1] == 3).all()
    assert (idx[0, 2] == 0).all()


def test_ball_query_padding_object_rows():
    # all-ones object: every point is within radius of every centre -> first nsample indices
    xyz = np.ones((1, 1024, 3), np.float32)
    new_xyz = np.ones((1, 32, 3), np.float32)
    idx = pn2.ball_query(new_xyz, xyz, 0.2, 32)
    assert (idx[0] == np.arange(32)[None, :]).all()


def test_ball_query_strict_less_and_order():
    xyz = np.zeros((1, 6, 3), np.float32)
    xyz[0, :, 0] = [0.0, 0.5, 0.25, 0.1, 0.5, 0.3]
    new_xyz = np.zeros((1, 1, 3), np.float32)
    idx = pn2.ball_query(new_xyz, xyz, 0.5, 4)   # d2 < 0.25 strictly: k=0,2,3,5
    assert idx[0, 0].tolist() == [0, 2, 3, 5]
    idx = pn2.ball_query(new_xyz, xyz, 0.5, 8)   # fewer hits than nsample: tail = first hit
    assert idx[0, 0].tolist() == [0, 2, 3, 5, 0, 0, 0, 0]


def test_three_interpolate_reference_test_input():
    # the reference's only test input: pointnet2_test.py:25-27
    feats = np.arange(8, dtype=np.float32).reshape(1, 2, 4) + 1
    idx = np.array([[[0, 1, 2], [1, 2, 3]]], np.int32)
    w = np.array([[[1, 1, 1], [2, 2, 2]]], np.float32)
    out = pn2.three_interpolate(feats, idx, w)
    want = np.array([[[1 + 2 + 3, 2 * (2 + 3 + 4)], [5 + 6 + 7, 2 * (6 + 7 + 8)]]], np.float32)
    assert (out == want).all()
    g = np.ones((1, 2, 2), np.float32)
    gp = pn2.three_interpolate_grad(g, idx, w, 4)
    assert gp[0, 0].tolist() == [1, 3, 3, 2]


def test_three_nn_fewer_than_three_known():
    # interpolate_gpu.cu:27-28: bests start at 1e40 (-> inf in f32), indices at 0
    unknown = np.zeros((1, 2, 3), np.float32)
    known = np.array([[[1, 0, 0], [0, 2, 0]]], np.float32)
    d2, idx = pn2.three_nn(unknown, known)
    assert d2[0, 0, 0] == 1 and d2[0, 0, 1] == 4 and np.isinf(d2[0, 0, 2])
    assert idx[0, 0].tolist() == [0, 1, 0]


def test_three_nn_ties_keep_earlier():
    unknown = np.zeros((1, 1, 3), np.float32)
    known = np.array([[[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0]]], np.float32)
    _, idx = pn2.three_nn(unknown, known)
    assert idx[0, 0].tolist() == [0, 1, 2]


def test_gather_group_exact_copy():
    rng = np.random.default_rng(1)
    pts = rng.standard_normal((2, 5, 40)).astype(np.float32)
    idx = rng.integers(0, 40, (2, 7)).astype(np.int32)
    out = pn2.gather_points(pts, idx)
    for b in range(2):
        assert (out[b] == pts[b][:, idx[b]]).all()
    gidx = rng.integers(0, 40, (2, 7, 3)).astype(np.int32)
    g = pn2.group_points(pts, gidx)
    for b in range(2):
        assert (g[b] == pts[b][:, gidx[b]]).all()
    # grads: scatter-add == dense one-hot matmul
    go = rng.standard_normal((2, 5, 7, 3)).astype(np.float32)
    gg = pn2.group_points_grad(go, gidx, 40)
    want = np.zeros((2, 5, 40), np.float64)
    for b in range(2):
        for j in range(7):
            for k in range(3):
                want[b, :, gidx[b, j, k]] += go[b, :, j, k]
    assert np.allclose(gg, want, atol=1e-5)
