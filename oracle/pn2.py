"""ctypes binding of oracle/pn2_oracle.c (the CPU restatement of pointnet2._ext).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under msr3d_amd/ may import this.

Functions take / return numpy arrays (C-contiguous f32 / i32) with the shapes of
the reference's Python module `pointnet2._ext`
(/root/reference/modules/third_party/pointnet2/_ext_src/src/bindings.cpp:6-19).
`ext_module()` wraps them for torch CPU tensors so the oracle can be injected as
`pointnet2_utils._ext` when the reference's Python is imported (golden
generation, tests/golden/make_golden.py).
"""
import ctypes
import os
import subprocess
import types

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpn2_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "pn2_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libpn2_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.pn2o_opt_n_threads.restype = ctypes.c_int
        _lib.pn2o_max_threads.restype = ctypes.c_int
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


CONTRACTS = {0: "fma(c,c,fma(a,a,b*b))  [LLVM, default]", 1: "(a*a+b*b)+c*c  [no fma]",
             2: "fma(c,c,fma(b,b,a*a))", 3: "fma(a,a,fma(b,b,c*c))",
             4: "contract 0 + 1 ulp  [envelope, not a contract]", 5: "contract 0 - 1 ulp  [envelope, not a contract]"}


def set_contract(c):
    """Floating-point contract of the squared distance a*a + b*b + c*c in FPS / ball query / three_nn
    (see pn2_oracle.c: sq3).  0 = default = what the HIP kernels are built with."""
    lib().pn2o_set_contract(int(c))


def get_contract():
    lib().pn2o_get_contract.restype = ctypes.c_int
    return lib().pn2o_get_contract()


def set_threads(t):
    lib().pn2o_set_threads(int(t))


def max_threads():
    return lib().pn2o_max_threads()


def opt_n_threads(work):
    return lib().pn2o_opt_n_threads(int(work))


def opt_block_config(x, y):
    bx, by = ctypes.c_int(), ctypes.c_int()
    lib().pn2o_opt_block_config(int(x), int(y), ctypes.byref(bx), ctypes.byref(by))
    return bx.value, by.value


def furthest_point_sampling(xyz, m):
    xyz, p = _f(xyz)
    b, n, _ = xyz.shape
    out = np.zeros((b, m), np.int32)
    lib().pn2o_furthest_point_sampling(b, n, int(m), p, out.ctypes.data_as(_i32p))
    return out


def gather_points(points, idx):
    points, pp = _f(points)
    idx, ip = _i(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.zeros((b, c, m), np.float32)
    lib().pn2o_gather_points(b, c, n, m, pp, ip, out.ctypes.data_as(_f32p))
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, gp = _f(grad_out)
    idx, ip = _i(idx)
    b, c, m = grad_out.shape
    out = np.zeros((b, c, n), np.float32)
    lib().pn2o_gather_points_grad(b, c, int(n), m, gp, ip, out.ctypes.data_as(_f32p))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    new_xyz, qp = _f(new_xyz)
    xyz, pp = _f(xyz)
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    out = np.zeros((b, m, nsample), np.int32)
    lib().pn2o_ball_query(b, n, m, ctypes.c_float(radius), int(nsample), qp, pp,
                          out.ctypes.data_as(_i32p))
    return out


def group_points(points, idx):
    points, pp = _f(points)
    idx, ip = _i(idx)
    b, c, n = points.shape
    _, npoints, nsample = idx.shape
    out = np.zeros((b, c, npoints, nsample), np.float32)
    lib().pn2o_group_points(b, c, n, npoints, nsample, pp, ip, out.ctypes.data_as(_f32p))
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, gp = _f(grad_out)
    idx, ip = _i(idx)
    b, c, npoints, nsample = grad_out.shape
    out = np.zeros((b, c, n), np.float32)
    lib().pn2o_group_points_grad(b, c, int(n), npoints, nsample, gp, ip,
                                 out.ctypes.data_as(_f32p))
    return out


def three_nn(unknown, known):
    unknown, up = _f(unknown)
    known, kp = _f(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.zeros((b, n, 3), np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    lib().pn2o_three_nn(b, n, m, up, kp, dist2.ctypes.data_as(_f32p), idx.ctypes.data_as(_i32p))
    return dist2, idx


def three_interpolate(points, idx, weight):
    points, pp = _f(points)
    idx, ip = _i(idx)
    weight, wp = _f(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.zeros((b, c, n), np.float32)
    lib().pn2o_three_interpolate(b, c, m, n, pp, ip, wp, out.ctypes.data_as(_f32p))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, gp = _f(grad_out)
    idx, ip = _i(idx)
    weight, wp = _f(weight)
    b, c, n = grad_out.shape
    out = np.zeros((b, c, m), np.float32)
    lib().pn2o_three_interpolate_grad(b, c, n, int(m), gp, ip, wp, out.ctypes.data_as(_f32p))
    return out


def ext_module():
    """A stand-in for `pointnet2._ext` over torch CPU tensors (golden generation only)."""
    import torch

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def npf(x):
        return x.detach().cpu().contiguous().numpy()

    m = types.SimpleNamespace()
    m.furthest_point_sampling = lambda p, k: t(furthest_point_sampling(npf(p), k))
    m.gather_points = lambda p, i: t(gather_points(npf(p), npf(i)))
    m.gather_points_grad = lambda g, i, n: t(gather_points_grad(npf(g), npf(i), n))
    m.ball_query = lambda q, p, r, ns: t(ball_query(npf(q), npf(p), r, ns))
    m.group_points = lambda p, i: t(group_points(npf(p), npf(i)))
    m.group_points_grad = lambda g, i, n: t(group_points_grad(npf(g), npf(i), n))
    m.three_nn = lambda u, k: [t(a) for a in three_nn(npf(u), npf(k))]
    m.three_interpolate = lambda p, i, w: t(three_interpolate(npf(p), npf(i), npf(w)))
    m.three_interpolate_grad = lambda g, i, w, mm: t(
        three_interpolate_grad(npf(g), npf(i), npf(w), mm))
    return m
