"""The sampling launch on shapes the encoder tests do not reach (they all run 1024 points of 6 floats, 32 / 16 centres, 32
slots): msr3d_sa_fps2_query_flags -- FPS of two levels and level 1's ball query in one launch, csrc/pn2_device.h -- and
msr3d_sa_fps2_query_plan against the oracle (oracle/pn2: sampling_gpu.cu:69-173 and ball_query_gpu.cu:9-44 restated),
bit for bit: clouds whose size is not a multiple of the query's 256-point round, packed xyz / odd point strides (the
staging's scalar path), few and many centres, rows shorter and longer than a wave, radii that catch nothing / everything,
constant clouds, a valid mask."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import pn2

pytestmark = pytest.mark.gpu


def _clouds(b, n, ps, seed, spread=1.0):
    g = torch.Generator().manual_seed(seed)
    pts = (torch.rand(b, n, ps, generator=g) - 0.5) * spread
    pts[:, :, :3] += 0.7                                   # (|p|^2 > 1e-3: no point is skipped by sampling_gpu.cu:100-101)
    return pts


def _run(entry, pts, m1, m2, radius, nsample, valid=None, plan=False):
    from msr3d_amd import _lib
    lib = _lib.load()
    b, n, ps = pts.shape
    i32 = dict(dtype=torch.int32, device="cuda")
    idx1, idx2 = torch.full((b, m1), -1, **i32), torch.full((b, max(m2, 1)), -1, **i32)
    xyz1 = torch.full((b, m1, 3), float("nan"), device="cuda")
    xyz2 = torch.full((b, max(m2, 1), 3), float("nan"), device="cuda")
    ball = torch.full((b, m1, nsample), -1, **i32)
    const = torch.full((b,), 7, dtype=torch.uint8, device="cuda")
    p = lambda t: None if t is None else t.data_ptr()     # noqa: E731
    vm = None if valid is None else valid.view(torch.uint8)
    st = _lib.current_stream_ptr()
    if plan:
        ws1 = torch.empty((int(lib.msr3d_sa_level1_rows_ws_bytes(b, m1)),), dtype=torch.uint8, device="cuda")
        ws2 = torch.empty((int(lib.msr3d_sa_level2_rows_ws_bytes(b)),), dtype=torch.uint8, device="cuda")
        out2 = torch.empty((b, max(m2, 1), 256), device="cuda")
        rc = lib.msr3d_sa_fps2_query_plan(b, n, ps, m1, m2, p(pts), p(idx1), p(xyz1), p(idx2), p(xyz2), p(vm),
                                          ctypes.c_float(radius), nsample, p(ball), p(const), p(ws1), ctypes.c_float(0.4),
                                          p(out2), None, p(ws2), st)
    elif entry == "fps":                                   # the sampling alone (any cloud size: msr3d_sa_fps2_flags)
        rc = lib.msr3d_sa_fps2_flags(b, n, ps, m1, m2, p(pts), p(idx1), p(xyz1), p(idx2), p(xyz2), p(vm), p(const), st)
    else:
        rc = lib.msr3d_sa_fps2_query_flags(b, n, ps, m1, m2, p(pts), p(idx1), p(xyz1), p(idx2), p(xyz2), p(vm),
                                           ctypes.c_float(radius), nsample, p(ball), p(const), st)
    torch.cuda.synchronize()
    return rc, dict(idx1=idx1, idx2=idx2, xyz1=xyz1, xyz2=xyz2, ball=ball, const=const)


def _oracle(pts, m1, m2, radius, nsample):
    xyz = pts[..., :3].contiguous().cpu().numpy()
    i1 = pn2.furthest_point_sampling(xyz, m1)
    nx1 = np.take_along_axis(xyz, i1[..., None].astype(np.int64).repeat(3, -1), 1)
    ball = pn2.ball_query(nx1, xyz, radius, nsample)
    i2 = nx2 = None
    if m2 > 0:
        i2 = pn2.furthest_point_sampling(nx1, m2)
        nx2 = np.take_along_axis(nx1, i2[..., None].astype(np.int64).repeat(3, -1), 1)
    return i1, nx1, ball, i2, nx2


@pytest.mark.parametrize("n,ps,m1,m2,nsample,radius", [
    (1024, 6, 32, 16, 32, 0.2), (300, 6, 32, 16, 32, 0.25), (513, 3, 7, 5, 8, 0.3), (777, 7, 64, 16, 40, 0.2),
    (1000, 4, 33, 0, 100, 0.5), (512, 3, 32, 16, 64, 5.0), (1024, 3, 32, 16, 32, 1e-4), (260, 6, 16, 16, 32, 0.15),
    (768, 12, 48, 9, 32, 0.3), (1024, 6, 48, 40, 32, 0.2), (640, 3, 63, 63, 16, 0.2)])
def test_sampling_launch_against_the_oracle(n, ps, m1, m2, nsample, radius):
    b = 6
    pts = _clouds(b, n, ps, seed=n + ps + m1).cuda()
    pts[4] = pts[4, :1]                                    # a constant cloud
    pts[5, :, :3] = pts[5, :1, :3]                         # constant coordinates, other channels not: NOT a constant cloud
    if ps == 3:
        pts[5, -1, 2] += 0.125
    rc, got = _run("flags", pts, m1, m2, radius, nsample)
    assert rc == 0
    i1, nx1, ball, i2, nx2 = _oracle(pts, m1, m2, radius, nsample)
    assert np.array_equal(got["idx1"].cpu().numpy(), i1)
    assert np.array_equal(got["xyz1"].cpu().numpy(), nx1)
    assert np.array_equal(got["ball"].cpu().numpy(), ball)
    if m2 > 0:
        assert np.array_equal(got["idx2"].cpu().numpy(), i2)
        assert np.array_equal(got["xyz2"].cpu().numpy(), nx2)
    flags = got["const"].tolist()
    assert flags[:5] == [0, 0, 0, 0, 1] and flags[5] == 0


@pytest.mark.parametrize("n,ps,m1,m2", [(1024, 6, 7, 5), (200, 3, 30, 29), (3000, 3, 64, 40), (64, 3, 48, 33), (5000, 6, 17, 17)])
def test_second_level_of_the_sampling_with_more_centres_than_its_block(n, ps, m1, m2):
    """The second level is the reference's launch over the m1 winners: a block of 2^floor(log2 m1) threads, so two points a
    thread when m1 is not a power of two (sampling_gpu.cu:69-173 with cuda_utils.h:13-19).  Until round 6 the fused kernels
    ranked only the first 2^floor(log2 m1) winners at that level -- invisible at the shipped 32 -> 16 (and wherever
    m2 <= that block: FPS over an FPS prefix re-picks the prefix), wrong beyond: found by this file's first case."""
    pts = _clouds(3, n, ps, seed=n + m1).cuda()
    rc, got = _run("fps", pts, m1, m2, 0.2, 32)
    assert rc == 0
    i1, nx1, _, i2, nx2 = _oracle(pts, m1, m2, 0.2, 4)
    assert np.array_equal(got["idx1"].cpu().numpy(), i1) and np.array_equal(got["idx2"].cpu().numpy(), i2)
    assert np.array_equal(got["xyz2"].cpu().numpy(), nx2)


@pytest.mark.parametrize("n,ps,m1,m2,masked", [(1024, 6, 32, 16, False), (300, 6, 32, 16, True), (512, 3, 64, 16, False),
                                               (644, 4, 20, 7, True)])
def test_sampling_launch_with_the_plans_inside_is_the_sampling_launch(n, ps, m1, m2, masked):
    """msr3d_sa_fps2_query_plan writes what msr3d_sa_fps2_query_flags writes (the plans themselves are pinned through the
    levels that consume them: tests/test_sa_rows_gpu.py)."""
    b = 9
    pts = _clouds(b, n, ps, seed=3 * n + m1).cuda()
    pts[2] = pts[2, :1]
    valid = None
    if masked:
        valid = torch.tensor([1, 0, 1, 1, 0, 1, 1, 1, 0], dtype=torch.bool, device="cuda")
    rc0, ref = _run("flags", pts, m1, m2, 0.2, 32, valid)
    rc1, got = _run("plan", pts, m1, m2, 0.2, 32, valid, plan=True)
    assert rc0 == 0 and rc1 == 0
    for k in ref:
        assert torch.equal(ref[k], got[k], ) or (ref[k].dtype.is_floating_point and
                                                 torch.equal(torch.nan_to_num(ref[k], nan=-7.0), torch.nan_to_num(got[k], nan=-7.0))), k


def test_sampling_launch_with_plans_refuses_what_the_planners_do_not_take():
    pts = _clouds(2, 512, 6, seed=1).cuda()
    assert _run("plan", pts, 32, 16, 0.2, 16, plan=True)[0] == -22        # rows of 16 slots
    assert _run("plan", pts, 32, 17, 0.2, 32, plan=True)[0] == -22        # 17 level-2 centres
    assert _run("plan", _clouds(2, 301, 3, seed=2).cuda(), 32, 16, 0.2, 32, plan=True)[0] == -22   # rows not 16-byte aligned
    assert _run("plan", _clouds(2, 2048, 3, seed=2).cuda(), 32, 16, 0.2, 32, plan=True)[0] == -22  # not the fused kernel's shape
