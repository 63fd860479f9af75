"""`pointnet2._ext` for MI355X: the nine functions of the reference's pybind module
(/root/reference/modules/third_party/pointnet2/_ext_src/src/bindings.cpp:6-19), same
names, argument order, shapes, dtypes and error behaviour, over libmsr3d_hip.so.

Boundary contract:
  * inputs are borrowed and never mutated; every output is allocated here on the
    input's device and fully defined by the kernel;
  * argument checks mirror include/utils.h:5-25 -- contiguous, f32 / i32, on the GPU;
    a violation raises RuntimeError (AT_ASSERT -> RuntimeError in the reference);
    CPU tensors raise "CPU not supported" (sampling.cpp:33-35 et al.);
  * kernels are enqueued on torch's CURRENT stream, asynchronously, no host sync;
  * a failed launch raises RuntimeError (the reference prints and exit(-1)s).
"""
import ctypes

import torch

from .. import _lib


def _chk(t, name, dtype):
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != dtype:
        kind = "a float" if dtype == torch.float32 else "an int"
        raise RuntimeError(f"{name} must be {kind} tensor")
    if not t.is_cuda:
        raise RuntimeError("CPU not supported")


def _same_device(a, *rest):
    for t in rest:
        if t.device != a.device:
            raise RuntimeError("all tensors must be on the same device")


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _run(fn_name, device, *args):
    lib = _lib.load()
    with torch.cuda.device(device):
        with _lib.kernel_timer(fn_name):
            st = getattr(lib, fn_name)(*args, _lib.current_stream_ptr(device))
    _lib.check(st, fn_name)


def gather_points(points, idx):
    _chk(points, "points", torch.float32)
    _chk(idx, "idx", torch.int32)
    _same_device(points, idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.empty((b, c, m), dtype=torch.float32, device=points.device)
    _run("msr3d_gather_points", points.device, b, c, n, m, _p(points), _p(idx), _p(out))
    return out


def gather_points_grad(grad_out, idx, n):
    _chk(grad_out, "grad_out", torch.float32)
    _chk(idx, "idx", torch.int32)
    _same_device(grad_out, idx)
    b, c, m = grad_out.shape
    out = torch.empty((b, c, int(n)), dtype=torch.float32, device=grad_out.device)
    _run("msr3d_gather_points_grad", grad_out.device, b, c, int(n), m, _p(grad_out), _p(idx),
         _p(out))
    return out


def furthest_point_sampling(points, nsamples):
    _chk(points, "points", torch.float32)
    b, n, three = points.shape
    if three != 3:
        raise RuntimeError("points must be (B, N, 3)")
    out = torch.empty((b, int(nsamples)), dtype=torch.int32, device=points.device)
    _run("msr3d_furthest_point_sampling", points.device, b, n, int(nsamples), _p(points), _p(out),
         ctypes.c_void_p(0))
    return out


def three_nn(unknowns, knows):
    _chk(unknowns, "unknowns", torch.float32)
    _chk(knows, "knows", torch.float32)
    _same_device(unknowns, knows)
    b, n, _ = unknowns.shape
    m = knows.shape[1]
    dist2 = torch.empty((b, n, 3), dtype=torch.float32, device=unknowns.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=unknowns.device)
    _run("msr3d_three_nn", unknowns.device, b, n, m, _p(unknowns), _p(knows), _p(dist2), _p(idx))
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    _chk(points, "points", torch.float32)
    _chk(idx, "idx", torch.int32)
    _chk(weight, "weight", torch.float32)
    _same_device(points, idx, weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = torch.empty((b, c, n), dtype=torch.float32, device=points.device)
    _run("msr3d_three_interpolate", points.device, b, c, m, n, _p(points), _p(idx), _p(weight),
         _p(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    _chk(grad_out, "grad_out", torch.float32)
    _chk(idx, "idx", torch.int32)
    _chk(weight, "weight", torch.float32)
    _same_device(grad_out, idx, weight)
    b, c, n = grad_out.shape
    out = torch.empty((b, c, int(m)), dtype=torch.float32, device=grad_out.device)
    _run("msr3d_three_interpolate_grad", grad_out.device, b, c, n, int(m), _p(grad_out), _p(idx),
         _p(weight), _p(out))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    # NB: centres first, like the reference's binding (ball_query.cpp:8-9)
    _chk(new_xyz, "new_xyz", torch.float32)
    _chk(xyz, "xyz", torch.float32)
    _same_device(new_xyz, xyz)
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = torch.empty((b, m, int(nsample)), dtype=torch.int32, device=new_xyz.device)
    _run("msr3d_ball_query", new_xyz.device, b, n, m, ctypes.c_float(radius), int(nsample),
         _p(new_xyz), _p(xyz), _p(idx))
    return idx


def group_points(points, idx):
    _chk(points, "points", torch.float32)
    _chk(idx, "idx", torch.int32)
    _same_device(points, idx)
    b, c, n = points.shape
    _, npoints, nsample = idx.shape
    out = torch.empty((b, c, npoints, nsample), dtype=torch.float32, device=points.device)
    _run("msr3d_group_points", points.device, b, c, n, npoints, nsample, _p(points), _p(idx),
         _p(out))
    return out


def group_points_grad(grad_out, idx, n):
    _chk(grad_out, "grad_out", torch.float32)
    _chk(idx, "idx", torch.int32)
    _same_device(grad_out, idx)
    b, c, npoints, nsample = grad_out.shape
    out = torch.empty((b, c, int(n)), dtype=torch.float32, device=grad_out.device)
    _run("msr3d_group_points_grad", grad_out.device, b, c, int(n), npoints, nsample, _p(grad_out),
         _p(idx), _p(out))
    return out
