"""GPU parity: the nine HIP ops (through the C ABI via msr3d_amd.pointnet2._ext)
against the CPU oracle on identical seeded inputs.  Index ops are bit-exact;
fp gathers/interpolation are bit-exact too (same fma chain); only the atomic
scatter-adds carry a tolerance (summation order)."""
import numpy as np
import pytest
import torch

from oracle import pn2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ext():
    from msr3d_amd.pointnet2 import _ext
    return _ext


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def unit_ball_cloud(rng, b, n, dup_frac=0.0):
    """Objects as the dataset makes them: centred, max-norm 1 (data/datasets/msr3d.py:199-209),
    optionally sampled with replacement (duplicates -> exact distance ties)."""
    out = np.empty((b, n, 3), np.float32)
    for i in range(b):
        size = rng.uniform(0.1, 2.0, 3)
        n_raw = n if dup_frac == 0 else max(8, int(n * (1 - dup_frac)))
        raw = rng.uniform(-0.5, 0.5, (n_raw, 3)) * size
        pts = raw[rng.integers(0, n_raw, n)] if n_raw < n else raw
        pts = pts - pts.mean(0)
        pts = pts / max(np.max(np.linalg.norm(pts, axis=1)), 1e-12)
        out[i] = pts.astype(np.float32)
    return out


FPS_CASES = [
    # (b, n, m, kind)
    (7, 1024, 32, "ball"), (5, 1024, 32, "dup"), (3, 1024, 32, "ones"), (3, 1024, 32, "tiny"),
    (4, 1024, 32, "grid"), (6, 32, 16, "ball"), (6, 32, 16, "dup"), (3, 2048, 64, "ball"),
    (3, 2048, 64, "dup"), (2, 1000, 40, "ball"), (2, 513, 33, "grid"), (2, 100, 100, "dup"),
    (2, 64, 8, "ball"), (2, 65, 8, "ball"), (3, 1, 1, "ball"), (2, 3, 3, "ball"), (2, 17, 20, "dup"),
    (2, 4096, 128, "ball"), (1, 5000, 64, "dup"), (1, 9000, 40, "ball"), (1, 20000, 24, "grid"),
    (2, 1024, 1, "ball"), (2, 255, 12, "mixed"), (4, 1024, 32, "mixed"),
]


def make_cloud(rng, b, n, kind):
    if kind == "ball":
        return unit_ball_cloud(rng, b, n)
    if kind == "dup":
        return unit_ball_cloud(rng, b, n, dup_frac=0.7)
    if kind == "ones":
        return np.ones((b, n, 3), np.float32)           # dataset_wrapper.py:156 padding object
    if kind == "tiny":
        return (rng.standard_normal((b, n, 3)) * 5e-3).astype(np.float32)  # mostly skipped
    if kind == "grid":                                   # lattice: massive exact ties
        g = rng.integers(-3, 4, (b, n, 3)).astype(np.float32) * 0.25
        return g
    if kind == "mixed":                                  # some skipped, some dup, some far
        x = unit_ball_cloud(rng, b, n, dup_frac=0.5)
        x[:, ::3] *= 0.02
        return x
    raise ValueError(kind)


@pytest.mark.parametrize("b,n,m,kind", FPS_CASES)
def test_fps_bit_exact(ext, b, n, m, kind):
    rng = np.random.default_rng(hash((b, n, m, kind)) % (2 ** 32))
    xyz = make_cloud(rng, b, n, kind)
    want = pn2.furthest_point_sampling(xyz, m)
    got = ext.furthest_point_sampling(dev(xyz), m).cpu().numpy()
    assert got.dtype == np.int32 and got.shape == (b, m)
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]


def test_fps_tie_rule_kat(ext):
    # same closed-form cases as tests/test_oracle_kat.py::test_fps_tie_break
    for n, a, bb, want in [(1024, 5, 517, 5), (1024, 3, 514, 514), (1024, 1, 2, 2),
                           (1024, 1, 256, 256), (32, 1, 2, 2)]:
        xyz = np.full((1, n, 3), 0.5, np.float32)
        xyz[0, a] = (0.5, 0.5, 0.9)
        xyz[0, bb] = (0.5, 0.5, 0.9)
        got = ext.furthest_point_sampling(dev(xyz), 2).cpu().numpy()
        assert got[0, 1] == want


BQ_CASES = [
    (7, 1024, 32, 0.2, 32, "ball"), (5, 1024, 32, 0.2, 32, "dup"), (3, 1024, 32, 0.2, 32, "ones"),
    (6, 32, 16, 0.4, 32, "ball"), (2, 2048, 64, 0.2, 64, "ball"), (2, 1000, 7, 0.05, 5, "ball"),
    (2, 100, 9, 10.0, 16, "ball"), (2, 100, 9, 1e-4, 16, "ball"), (1, 5000, 130, 0.3, 48, "grid"),
    (3, 70, 3, 0.4, 1, "ball"), (1, 9000, 16, 0.1, 100, "ball"),
]


@pytest.mark.parametrize("b,n,m,r,ns,kind", BQ_CASES)
def test_ball_query_bit_exact(ext, b, n, m, r, ns, kind):
    rng = np.random.default_rng(hash((b, n, m, ns, kind)) % (2 ** 32))
    xyz = make_cloud(rng, b, n, kind)
    cidx = rng.integers(0, n, (b, m))
    new_xyz = np.take_along_axis(xyz, cidx[..., None].repeat(3, -1), 1).copy()
    if kind == "ball":
        new_xyz[:, -1] += 100.0   # a centre with no neighbour at all -> zero row
    want = pn2.ball_query(new_xyz, xyz, r, ns)
    got = ext.ball_query(dev(new_xyz), dev(xyz), r, ns).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("b,c,n,npoint,ns", [(5, 3, 1024, 32, 32), (4, 128, 32, 16, 32),
                                             (2, 7, 50, 5, 3), (1, 259, 16, 1, 16)])
def test_group_gather_exact(ext, b, c, n, npoint, ns):
    rng = np.random.default_rng(5)
    pts = rng.standard_normal((b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, (b, npoint, ns)).astype(np.int32)
    got = ext.group_points(dev(pts), dev(idx)).cpu().numpy()
    assert np.array_equal(got, pn2.group_points(pts, idx))
    gidx = rng.integers(0, n, (b, npoint)).astype(np.int32)
    got = ext.gather_points(dev(pts), dev(gidx)).cpu().numpy()
    assert np.array_equal(got, pn2.gather_points(pts, gidx))
    # grads: the kernels sum every destination's contributions in ascending source order, the
    # order of the sequential oracle -> bit-exact (the reference's atomicAdd order is arbitrary)
    go = rng.standard_normal((b, c, npoint, ns)).astype(np.float32)
    got = ext.group_points_grad(dev(go), dev(idx), n).cpu().numpy()
    assert np.array_equal(got, pn2.group_points_grad(go, idx, n))
    go = rng.standard_normal((b, c, npoint)).astype(np.float32)
    got = ext.gather_points_grad(dev(go), dev(gidx), n).cpu().numpy()
    assert np.array_equal(got, pn2.gather_points_grad(go, gidx, n))


@pytest.mark.parametrize("b,n,m,c", [(3, 500, 64, 16), (2, 64, 2, 4), (2, 1300, 1100, 3), (1, 5, 1, 2)])
def test_three_nn_interpolate(ext, b, n, m, c):
    rng = np.random.default_rng(9)
    unknown = rng.standard_normal((b, n, 3)).astype(np.float32)
    known = rng.standard_normal((b, m, 3)).astype(np.float32)
    if m > 8:
        known[:, 5] = known[:, 2]     # duplicates: ties keep the earlier index
    d2w, iw = pn2.three_nn(unknown, known)
    d2, i = ext.three_nn(dev(unknown), dev(known))
    assert np.array_equal(i.cpu().numpy(), iw)
    assert np.array_equal(d2.cpu().numpy(), d2w)
    feats = rng.standard_normal((b, c, m)).astype(np.float32)
    w = rng.uniform(0, 1, (b, n, 3)).astype(np.float32)
    got = ext.three_interpolate(dev(feats), dev(iw), dev(w)).cpu().numpy()
    assert np.array_equal(got, pn2.three_interpolate(feats, iw, w))
    go = rng.standard_normal((b, c, n)).astype(np.float32)
    got = ext.three_interpolate_grad(dev(go), dev(iw), dev(w), m).cpu().numpy()
    assert np.array_equal(got, pn2.three_interpolate_grad(go, iw, w, m))       # ordered sums: bit-exact


def test_grads_heavy_collisions_and_oversize_fallback(ext):
    """All sources hitting a handful of destinations (long ordered sums, bit-exact), repeated runs
    identical; and a problem whose inverted index does not fit in LDS (atomic path, tolerance)."""
    rng = np.random.default_rng(21)
    b, c, n, npoint, ns = 3, 40, 1024, 32, 32
    idx = rng.integers(0, 4, (b, npoint, ns)).astype(np.int32)        # 1024 entries -> 4 destinations
    go = rng.standard_normal((b, c, npoint, ns)).astype(np.float32)
    want = pn2.group_points_grad(go, idx, n)
    first = ext.group_points_grad(dev(go), dev(idx), n)
    assert np.array_equal(first.cpu().numpy(), want)
    for _ in range(3):
        assert torch.equal(ext.group_points_grad(dev(go), dev(idx), n), first)
    n_big = 30000                                                      # 2n+1+E > 36 K ints
    idx = rng.integers(0, n_big, (1, 64, 32)).astype(np.int32)
    go = rng.standard_normal((1, 5, 64, 32)).astype(np.float32)
    got = ext.group_points_grad(dev(go), dev(idx), n_big).cpu().numpy()
    assert np.allclose(got, pn2.group_points_grad(go, idx, n_big), rtol=1e-5, atol=1e-5)


def test_reference_interpolate_test_input(ext):
    # pointnet2_test.py:25-27 (the reference's only test input for this extension)
    feats = np.arange(8, dtype=np.float32).reshape(1, 2, 4) + 1
    idx = np.array([[[0, 1, 2], [1, 2, 3]]], np.int32)
    w = np.array([[[1, 1, 1], [2, 2, 2]]], np.float32)
    got = ext.three_interpolate(dev(feats), dev(idx), dev(w)).cpu().numpy()
    assert got.tolist() == [[[6.0, 18.0], [18.0, 42.0]]]


def test_argument_errors(ext):
    x = torch.zeros(2, 8, 3, device="cuda")
    with pytest.raises(RuntimeError, match="CPU not supported"):
        ext.furthest_point_sampling(x.cpu(), 4)
    with pytest.raises(RuntimeError, match="contiguous"):
        ext.furthest_point_sampling(torch.zeros(2, 3, 8, device="cuda").transpose(1, 2), 4)
    with pytest.raises(RuntimeError, match="float"):
        ext.furthest_point_sampling(x.double(), 4)
    with pytest.raises(RuntimeError, match="int"):
        ext.gather_points(torch.zeros(2, 3, 8, device="cuda"), torch.zeros(2, 4, device="cuda").long())


def test_runs_on_current_stream_async(ext):
    # the op must be ordered on torch's current stream (no default-stream launch)
    s = torch.cuda.Stream()
    rng = np.random.default_rng(0)
    xyz = unit_ball_cloud(rng, 4, 1024)
    want = pn2.furthest_point_sampling(xyz, 32)
    with torch.cuda.stream(s):
        d = dev(xyz) * 1.0          # produced on s
        got = ext.furthest_point_sampling(d, 32)
    s.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)
