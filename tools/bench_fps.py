"""Times the sampling launch of the frozen encoder by itself on the bench's clouds: msr3d_sa_fps2_flags (the two FPS
levels alone) and msr3d_sa_fps2_query_flags (with level 1's ball query beside the chain), HIP events, median of --iters.
    python tools/bench_fps.py [--batch 16] [--iters 50]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--dense", action="store_true")
ap.add_argument("--lib", default=None, help="a shared object built from csrc/sa_fused.hip alone (A/B builds)")
args = ap.parse_args()

from msr3d_amd import _lib  # noqa: E402
from msr3d_amd.synth import synth_batch  # noqa: E402

lib = _lib.load()
if args.lib:
    import ctypes
    raw = ctypes.CDLL(os.path.abspath(args.lib))

    class _Raw:
        def __getattr__(self, name):
            fn = getattr(raw, name)
            fn.argtypes = _lib._SIGNATURES[name]
            fn.restype = ctypes.c_int
            return fn
    lib = _Raw()
pts = synth_batch(10000, args.batch, device="cuda", dense=args.dense)["obj_fts"].reshape(-1, 1024, 6).contiguous()
b = pts.shape[0]
i32 = dict(dtype=torch.int32, device="cuda")
idx1, idx2 = torch.empty(b, 32, **i32), torch.empty(b, 16, **i32)
xyz1, xyz2 = torch.empty(b, 32, 3, device="cuda"), torch.empty(b, 16, 3, device="cuda")
ball = torch.empty(b, 32, 32, **i32)
const = torch.empty(b, dtype=torch.uint8, device="cuda")
st = _lib.current_stream_ptr()


def fps():
    _lib.check(lib.msr3d_sa_fps2_flags(b, 1024, 6, 32, 16, pts.data_ptr(), idx1.data_ptr(), xyz1.data_ptr(),
                                       idx2.data_ptr(), xyz2.data_ptr(), None, const.data_ptr(), st), "fps2")


def fps_query():
    _lib.check(lib.msr3d_sa_fps2_query_flags(b, 1024, 6, 32, 16, pts.data_ptr(), idx1.data_ptr(), xyz1.data_ptr(),
                                             idx2.data_ptr(), xyz2.data_ptr(), None, 0.2, 32, ball.data_ptr(),
                                             const.data_ptr(), st), "fps2_query")


for name, fn in (("msr3d_sa_fps2_flags", fps), ("msr3d_sa_fps2_query_flags", fps_query)):
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
    for a, c in ev:
        a.record()
        fn()
        c.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(c) * 1e3 for a, c in ev)
    print(f"{name:28s} median {ts[len(ts) // 2]:7.1f} us   min {ts[0]:7.1f} us   ({b} clouds)")
print("checksums", int(idx1.sum()), int(idx2.sum()), int(ball.sum()), int(const.sum()))
