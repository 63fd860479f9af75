"""Distinct-row set abstraction (round 5; csrc/sa_split.hip::sa2_rows_kernel and its siblings): ball_query pads a
neighbourhood with copies of its first hit (ball_query_gpu.cu:35-39), the SharedMLP acts row by row and max is
idempotent, so multiplying only the distinct rows must give THE SAME BITS as multiplying all nsample rows.  Every
comparison here is `torch.equal` against the all-rows kernels of rounds 2-4 (which the goldens pin), on the benchmark's
960 clouds and on adversarial hit counts: 0, 1, 15, 16, 17, 32 and more hits per centre, every centre empty, all points
coincident, clouds flagged constant."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(seed):
    from msr3d_amd.modules.layers.pointnet import PointNetPP
    from tests.helpers import fill_state_dict
    net = PointNetPP(sa_n_points=[32, 16, None], sa_n_samples=[32, 32, None], sa_radii=[0.2, 0.4, None],
                     sa_mlps=[[3, 64, 64, 128], [128, 128, 128, 256], [256, 256, 512, 768]])
    net.load_state_dict(fill_state_dict(net.state_dict(), seed))
    return net.cuda().eval()


def _both(net, pts, valid=None):
    from msr3d_amd.pointnet2 import fused
    out = {}
    for rows in (False, True):
        prev = fused.set_sa_rows(rows)
        try:
            with torch.no_grad():
                out[rows] = fused.forward(net, pts, return_internals=True, valid=valid)
        finally:
            fused.set_sa_rows(prev)
    return out[False], out[True]


def _same(a, b, valid=None):
    (ya, da), (yb, db) = a, b
    keys = ("idx1", "idx2", "ball1", "ball2", "new_xyz1", "new_xyz2", "feat1", "feat2", "pooled")
    for k in keys:
        x, y = da[k], db[k]
        if valid is not None:
            x, y = x[valid], y[valid]
        assert torch.equal(x, y), k
    assert torch.equal(ya, yb)


def test_bench_batch_is_bit_identical():
    """The benchmark's own 16 scenes x 60 objects (a third of them padding clouds)."""
    from msr3d_amd.synth import synth_batch
    net = _net(3)
    batch = synth_batch(10000, 16, device="cuda")
    pts = batch["obj_fts"].reshape(-1, 1024, 6).contiguous()
    old, new = _both(net, pts)
    _same(old, new)
    # the sampling launch reports exactly the padding slots as constant clouds
    assert torch.equal(new[1]["constant"].bool(), ~batch["obj_masks"].reshape(-1))


@pytest.mark.parametrize("points", [300, 512, 1000])
def test_other_cloud_sizes_are_bit_identical(points):
    """Clouds that are not a multiple of the ball query's 256-point round (its clamped last round) and the plans written
    inside the sampling launch for them."""
    from msr3d_amd.synth import synth_batch
    net = _net(9)
    pts = synth_batch(points, 1, O=24, P=points, n_valid=19, device="cuda")["obj_fts"][0].contiguous()
    _same(*_both(net, pts))


def test_constant_flag_is_bitwise():
    """One repeated point -> 1; a single differing bit anywhere (last point's last channel; -0.0 against +0.0) -> 0."""
    from msr3d_amd.pointnet2 import fused
    net = _net(3)
    pts = torch.full((5, 1024, 6), 0.5, device="cuda")
    pts[1, 1023, 5] = torch.nextafter(torch.tensor(0.5), torch.tensor(1.0)).item()
    pts[2] = 0.0
    pts[3] = 0.0
    pts[3, 500, 1] = -0.0
    pts[4, 0, 0] = 0.25                                   # the FIRST point is the odd one
    with torch.no_grad():
        _, dbg = fused.forward(net, pts, return_internals=True)
    assert dbg["constant"].tolist() == [1, 0, 1, 0, 0]
    _same(*_both(net, pts))


@pytest.mark.parametrize("objects,keep", [(70, 0.6), (7, 0.5), (200, 1.0)])
def test_valid_mask_and_ragged_counts(objects, keep):
    from msr3d_amd.synth import synth_batch
    net = _net(11)
    scenes = (objects + 59) // 60
    pts = synth_batch(5, scenes, device="cuda")["obj_fts"].reshape(-1, 1024, 6)[:objects].contiguous()
    valid = None
    if keep < 1.0:
        g = torch.Generator().manual_seed(objects)
        valid = (torch.rand(objects, generator=g) < keep).cuda()
        valid[0] = False
        valid[-1] = False
        valid[objects // 2] = True
    _same(*_both(net, pts, valid), valid=valid)


def _level2_direct(net, xyz, feat, new_xyz, radius, rows, constant=None):
    """One level-2 launch through the C ABI on hand-made inputs -> (out (b, m, 256), ball (b, m, 32))."""
    from msr3d_amd import _lib
    from msr3d_amd.pointnet2 import fused
    S = fused.get_plan(net)["split2"]
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    out = torch.full((b, m, 256), float("nan"), device="cuda")
    ball = torch.full((b, m, 32), -1, dtype=torch.int32, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
    lib = _lib.load()
    st = _lib.current_stream_ptr(xyz.device)
    if rows:
        ws = torch.empty((int(lib.msr3d_sa_level2_rows_ws_bytes(b)),), dtype=torch.uint8, device="cuda")
        rc = lib.msr3d_sa_level2_rows(b, n, m, ctypes.c_float(radius), p(xyz), p(feat), p(new_xyz), p(S[0][0]), p(S[0][1]),
                                      p(S[1][0]), p(S[1][1]), p(S[2][0]), p(S[2][1]), p(out), p(ball), p(None), p(constant),
                                      p(ws), 0, st)
    else:
        rc = lib.msr3d_sa_level_split(2, b, n, m, ctypes.c_float(radius), p(xyz), p(feat), p(new_xyz), p(S[0][0]), p(S[0][1]),
                                      p(S[1][0]), p(S[1][1]), p(S[2][0]), p(S[2][1]), p(out), p(ball), p(None), st)
    _lib.check(rc, "level 2")
    torch.cuda.synchronize()
    return out, ball


def _adversarial_clouds(n, m, seed):
    """Objects whose centres have prescribed hit counts: points on a line, spacing d; a centre at position x with radius r
    catches the points within r of it.  Returns xyz (b, n, 3), new_xyz (b, m, 3), radius, expected hit counts (b, m)."""
    rng = np.random.default_rng(seed)
    radius = 0.4
    wants = [0, 1, 15, 16, 17, 31, 32, min(n, 40)]
    objs_xyz, objs_ctr, objs_cnt = [], [], []
    for k in range(6):
        # clusters of sizes drawn from `wants`, each cluster tight (within 0.01) and clusters 2.0 apart
        sizes = []
        while sum(sizes) < n:
            sizes.append(int(rng.choice(wants[1:])))
        sizes[-1] -= sum(sizes) - n
        sizes = [s for s in sizes if s > 0]
        pts, starts = [], []
        for ci, s in enumerate(sizes):
            starts.append(len(pts))
            base = np.array([2.0 * ci, 0.0, 0.0])
            for _ in range(s):
                pts.append(base + rng.uniform(-0.01, 0.01, 3))
        pts = np.array(pts, np.float32)
        perm = rng.permutation(n) if k % 2 else np.arange(n)
        pts = pts[perm]
        ctr, cnt = [], []
        for j in range(m):
            if j % 5 == 4:                           # a centre nobody is near: zero hits
                ctr.append([0.0, 50.0 + j, 0.0])
                cnt.append(0)
            else:
                ci = int(rng.integers(len(sizes)))
                ctr.append([2.0 * ci, 0.0, 0.0])
                cnt.append(sizes[ci])
        objs_xyz.append(pts)
        objs_ctr.append(np.array(ctr, np.float32))
        objs_cnt.append(cnt)
    return (torch.from_numpy(np.stack(objs_xyz)).cuda(), torch.from_numpy(np.stack(objs_ctr)).cuda(), radius,
            np.array(objs_cnt))


@pytest.mark.parametrize("n,m", [(32, 16), (64, 16), (20, 7), (33, 1)])
def test_level2_adversarial_hit_counts(n, m):
    net = _net(5)
    xyz, ctr, radius, cnt = _adversarial_clouds(n, m, seed=n * 100 + m)
    b = xyz.shape[0]
    g = torch.Generator().manual_seed(n + m)
    feat = torch.randn(b, n, 128, generator=g).cuda()
    feat[1] = feat[1].abs() * 1e-20                       # tiny activations: ReLU zeros everywhere
    want, ball_w = _level2_direct(net, xyz, feat, ctr, radius, rows=False)
    got, ball_g = _level2_direct(net, xyz, feat, ctr, radius, rows=True)
    assert torch.equal(ball_w, ball_g)
    # the prescribed hit counts are what ball_query found (the case list is what it claims to be)
    bw = ball_w.cpu().numpy()
    distinct = np.array([[len(set(bw[i, j].tolist())) for j in range(m)] for i in range(b)])
    assert (distinct == np.clip(cnt, 1, 32)).all()
    if n >= 32 and m >= 16:
        assert set(cnt.ravel().tolist()) >= {0, 1, 15, 16, 17, 32}, sorted(set(cnt.ravel().tolist()))
    assert torch.isfinite(got).all()
    assert torch.equal(want, got)


def test_level2_every_centre_empty_and_all_points_coincident():
    net = _net(6)
    g = torch.Generator().manual_seed(1)
    xyz = torch.rand(3, 32, 3, generator=g).cuda()
    xyz[1] = 0.25                                          # one repeated point: every centre catches all 32
    ctr = xyz[:, :16].clone().contiguous()
    ctr[0] += 100.0                                        # object 0: no centre has a neighbour
    feat = torch.randn(3, 32, 128, generator=g).cuda()
    feat[1] = feat[1, :1]                                  # ... with one repeated feature row: a constant cloud
    want, _ = _level2_direct(net, xyz, feat, ctr, 0.4, rows=False)
    got, _ = _level2_direct(net, xyz, feat, ctr, 0.4, rows=True)
    assert torch.equal(want, got)
    flags = torch.tensor([0, 1, 0], dtype=torch.uint8).cuda()
    got_c, _ = _level2_direct(net, xyz, feat, ctr, 0.4, rows=True, constant=flags)
    assert torch.equal(want, got_c)


def test_level2_rows_refuses_shapes_it_does_not_take():
    from msr3d_amd import _lib
    net = _net(6)
    xyz = torch.rand(1, 32, 3).cuda()
    feat = torch.randn(1, 32, 128).cuda()
    with pytest.raises(RuntimeError):
        _level2_direct(net, xyz, feat, torch.rand(1, 17, 3).cuda(), 0.4, rows=True)      # m > 16
    with pytest.raises(RuntimeError):
        _level2_direct(net, torch.rand(1, 65, 3).cuda(), torch.randn(1, 65, 128).cuda(), torch.rand(1, 4, 3).cuda(), 0.4, rows=True)


def test_one_planning_launch_for_both_levels_is_the_two_planning_launches():
    """msr3d_sa_plan12 (default) against each level planning in its own call: same bits, debug lists included."""
    from msr3d_amd.pointnet2 import fused
    from msr3d_amd.synth import synth_batch
    net = _net(4)
    pts = synth_batch(77, 3, device="cuda")["obj_fts"].reshape(-1, 1024, 6).contiguous()
    outs = {}
    for merged in (False, True):
        prev, fused._PLAN12 = fused._PLAN12, merged
        try:
            with torch.no_grad():
                outs[merged] = fused.forward(net, pts, return_internals=True)
        finally:
            fused._PLAN12 = prev
    _same(outs[False], outs[True])


@pytest.mark.parametrize("scenes,masked", [(16, False), (3, False), (2, True)])
def test_plans_written_by_the_sampling_launch_are_the_planning_launch(scenes, masked):
    """msr3d_sa_fps2_query_plan (default: the level-1 task list and the level-2 row lists written inside the sampling
    launch) against msr3d_sa_fps2_query_flags + msr3d_sa_plan12: every internal and the output, same bits; the path
    taken is checked through the entries that ran."""
    from msr3d_amd import _lib
    from msr3d_amd.pointnet2 import fused
    from msr3d_amd.synth import synth_batch
    net = _net(4)
    batch = synth_batch(78, scenes, device="cuda")
    pts = batch["obj_fts"].reshape(-1, 1024, 6).contiguous()
    valid = batch["obj_masks"].reshape(-1) if masked else None
    outs, ran = {}, {}
    for inside in (False, True):
        prev, fused._PLAN_IN_SAMPLING = fused._PLAN_IN_SAMPLING, inside
        sink = {}
        _lib.set_timing_sink(sink, census=True)
        try:
            with torch.no_grad():
                outs[inside] = fused.forward(net, pts, return_internals=True, valid=valid)
            torch.cuda.synchronize()
        finally:
            _lib.set_timing_sink(None)
            fused._PLAN_IN_SAMPLING = prev
        ran[inside] = set(sink)
    assert "msr3d_sa_fps2_query_plan" in ran[True] and "msr3d_sa_plan12" not in ran[True]
    assert "msr3d_sa_plan12" in ran[False] and "msr3d_sa_fps2_query_plan" not in ran[False]
    _same(outs[False], outs[True], valid=valid)
    for rep in range(3):                                   # the task list's order differs run to run; the results do not
        with torch.no_grad():
            again = fused.forward(net, pts, return_internals=True, valid=valid)
        _same(outs[True], again, valid=valid)


def test_planned_call_without_a_plan_is_refused():
    from msr3d_amd import _lib
    from msr3d_amd.pointnet2 import fused
    net = _net(6)
    S = fused.get_plan(net)["split2"]
    lib = _lib.load()
    b, n, m = 2, 32, 16
    xyz, feat, ctr = torch.rand(b, n, 3).cuda(), torch.randn(b, n, 128).cuda(), torch.rand(b, m, 3).cuda()
    out = torch.empty(b, m, 256, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
    S1 = fused.get_plan(net)["split1"]
    pts = torch.rand(b, 64, 6).cuda()
    ball = torch.zeros(b, n, 32, dtype=torch.int32, device="cuda")
    out1 = torch.empty(b, n, 128, device="cuda")
    ws1 = torch.empty((int(lib.msr3d_sa_level1_rows_ws_bytes(b, n)),), dtype=torch.uint8, device="cuda")
    st = _lib.current_stream_ptr(xyz.device)
    # a fresh stream has no plan: planned = 1 must not run on whatever the workspace holds
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        rc = lib.msr3d_sa_level1_rows(b, 64, n, p(pts), p(xyz), p(ball), p(S1[0][0]), p(S1[0][1]), p(S1[1][0]), p(S1[1][1]),
                                      p(S1[2][0]), p(S1[2][1]), p(out1), p(None), p(None), p(ws1), 1,
                                      _lib.current_stream_ptr(xyz.device))
    assert rc == -22
    torch.cuda.synchronize()
    del st, S, feat, ctr, out


@pytest.mark.parametrize("b,const_frac,valid_frac", [(960, 0.33, None), (960, 0.0, None), (960, 1.0, None), (5, 0.4, None),
                                                     (61, 0.5, 0.7), (770, 0.03, None), (4100, 0.5, None), (1, 0.0, None),
                                                     (1, 1.0, None), (49, 1.0, 0.5)])
def test_level3_tiles_against_four_object_tiles(b, const_frac, valid_frac):
    """msr3d_sa_level3_tiles (three real objects a workgroup, constant objects one row each, chosen on the device from
    the flags) against msr3d_sa_level_split(level 3): the same bits -- listed and fall-back forms (960 real objects need
    more workgroups than the launch has; 4100 objects are more than the device-side ranking takes), flags that LIE
    are not tested: `constant` is what the sampling launch reports, a constant object's sixteen rows are one row."""
    from msr3d_amd import _lib
    from msr3d_amd.pointnet2 import fused
    net = _net(5)
    lib = _lib.load()
    S = fused.get_plan(net)["split3"]
    g = torch.Generator().manual_seed(b * 7 + int(const_frac * 100))
    xyz = torch.rand(b, 16, 3, generator=g).cuda()
    feat = torch.randn(b, 16, 256, generator=g).cuda()
    const = (torch.rand(b, generator=g) < const_frac) if 0.0 < const_frac < 1.0 else torch.full((b,), const_frac >= 1.0)
    const = const.cuda()
    xyz[const] = xyz[const][:, :1]                          # a constant object: sixteen identical rows
    feat[const] = feat[const][:, :1]
    valid = None
    if valid_frac is not None:
        valid = (torch.rand(b, generator=g) < valid_frac).cuda()
        valid[0] = True
    p = lambda t: None if t is None else t.data_ptr()       # noqa: E731
    vm = None if valid is None else valid.view(torch.uint8)
    cm = const.view(torch.uint8).contiguous()
    st = _lib.current_stream_ptr()
    ref = torch.full((b, 768), float("nan"), device="cuda")
    _lib.check(lib.msr3d_sa_level_split(3, b, 16, 1, ctypes.c_float(0.0), p(xyz), p(feat), None, p(S[0][0]), p(S[0][1]),
                                        p(S[1][0]), p(S[1][1]), p(S[2][0]), p(S[2][1]), p(ref), None, p(vm), st), "level_split(3)")
    for flags in (cm, None):                                # without flags: every valid object is a real one
        out = torch.full((b, 768), float("nan"), device="cuda")
        _lib.check(lib.msr3d_sa_level3_tiles(b, p(xyz), p(feat), p(S[0][0]), p(S[0][1]), p(S[1][0]), p(S[1][1]), p(S[2][0]),
                                             p(S[2][1]), p(out), p(vm), p(flags), st), "level3_tiles")
        keep = torch.ones(b, dtype=torch.bool, device="cuda") if valid is None else valid
        assert torch.equal(out[keep], ref[keep])
        assert not torch.isnan(out[keep]).any()


@pytest.mark.parametrize("n,m1,m2,radius1", [(1024, 20, 7, 0.2), (512, 64, 16, 0.15), (768, 33, 16, 0.3), (300, 48, 9, 0.5),
                                             (1024, 32, 16, 1e-3)])
def test_in_launch_plans_for_other_centre_counts_feed_the_rows_kernels(n, m1, m2, radius1):
    """The plans msr3d_sa_fps2_query_plan writes, CONSUMED: msr3d_sa_level1_rows / msr3d_sa_level2_rows with planned = 1
    behind it against the same two calls planning for themselves behind msr3d_sa_fps2_query_flags -- centre counts other
    than the encoder's 32 / 16 (a lane a centre up to 64; level 2's quad per centre up to 16), clouds with a clamped last
    query round, a radius that catches nothing (every centre one row), a constant cloud and a skipped one."""
    from msr3d_amd import _lib
    from msr3d_amd.pointnet2 import fused
    net = _net(8)
    lib = _lib.load()
    S1, S2 = fused.get_plan(net)["split1"], fused.get_plan(net)["split2"]
    b = 7
    g = torch.Generator().manual_seed(n + m1)
    pts = (torch.rand(b, n, 6, generator=g) - 0.5).cuda()
    pts[:, :, :3] += 0.7
    pts[3] = pts[3, :1]                                                       # a constant cloud
    valid = torch.tensor([1, 1, 0, 1, 1, 1, 1], dtype=torch.bool, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)     # noqa: E731
    st = _lib.current_stream_ptr()
    i32 = dict(dtype=torch.int32, device="cuda")
    r2 = 0.4
    res = {}
    for inside in (True, False):
        new1, new2 = torch.zeros(b, m1, 3, device="cuda"), torch.zeros(b, m2, 3, device="cuda")
        ball1, ball2 = torch.zeros(b, m1, 32, **i32), torch.zeros(b, m2, 32, **i32)
        const = torch.zeros(b, dtype=torch.uint8, device="cuda")
        feat1 = torch.zeros(b, m1, 128, device="cuda")
        feat2 = torch.zeros(b, m2, 256, device="cuda")
        ws1 = torch.empty((int(lib.msr3d_sa_level1_rows_ws_bytes(b, m1)),), dtype=torch.uint8, device="cuda")
        ws2 = torch.empty((int(lib.msr3d_sa_level2_rows_ws_bytes(b)),), dtype=torch.uint8, device="cuda")
        vm = valid.view(torch.uint8)
        if inside:
            _lib.check(lib.msr3d_sa_fps2_query_plan(b, n, 6, m1, m2, p(pts), None, p(new1), None, p(new2), p(vm),
                                                    ctypes.c_float(radius1), 32, p(ball1), p(const), p(ws1), ctypes.c_float(r2),
                                                    p(feat2), p(ball2), p(ws2), st), "fps2_query_plan")
        else:
            _lib.check(lib.msr3d_sa_fps2_query_flags(b, n, 6, m1, m2, p(pts), None, p(new1), None, p(new2), p(vm),
                                                     ctypes.c_float(radius1), 32, p(ball1), p(const), st), "fps2_query_flags")
        planned = 1 if inside else 0
        _lib.check(lib.msr3d_sa_level1_rows(b, n, m1, p(pts), p(new1), p(ball1), p(S1[0][0]), p(S1[0][1]), p(S1[1][0]), p(S1[1][1]),
                                            p(S1[2][0]), p(S1[2][1]), p(feat1), p(vm), p(const), p(ws1), planned, st), "level1_rows")
        _lib.check(lib.msr3d_sa_level2_rows(b, m1, m2, ctypes.c_float(r2), p(new1), p(feat1), p(new2), p(S2[0][0]), p(S2[0][1]),
                                            p(S2[1][0]), p(S2[1][1]), p(S2[2][0]), p(S2[2][1]), p(feat2), p(ball2), p(vm),
                                            p(const), p(ws2), planned, st), "level2_rows")
        torch.cuda.synchronize()
        res[inside] = dict(new1=new1, new2=new2, ball1=ball1, ball2=ball2, const=const, feat1=feat1, feat2=feat2)
    for k in res[True]:
        assert torch.equal(res[True][k][valid], res[False][k][valid]), k
    assert res[True]["const"].tolist()[3] == 1 and float(res[True]["feat2"][valid].abs().max()) > 0
