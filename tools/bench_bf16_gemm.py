"""bf16 GEMM of the language-model layer (msr3d_bf16_gemm_lowrank) on its shapes: time, TFLOP/s, check against torch.
    MSR3D_BF16_GEMM=wide|glds|reg python tools/bench_bf16_gemm.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from msr3d_amd import _lib  # noqa: E402

lib = _lib.load()
if os.environ.get("MSR3D_GEMM_LIB"):        # an ablation build of lora_linear.hip alone (tools/_prof/)
    lib = ctypes.CDLL(os.environ["MSR3D_GEMM_LIB"])
    lib.msr3d_bf16_gemm_lowrank.argtypes = _lib.load().msr3d_bf16_gemm_lowrank.argtypes
    lib.msr3d_bf16_gemm_lowrank.restype = ctypes.c_int
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)   # noqa: E731
SHAPES = [(2304, 4096, 4096, 0), (2304, 4096, 4096, 64), (2304, 11008, 4096, 0), (2304, 4096, 11008, 0),
          (2000, 4096, 4096, 0), (576, 4096, 4096, 0), (8192, 8192, 8192, 0)]
if os.environ.get("MSR3D_GEMM_ONE"):
    SHAPES = SHAPES[:1]
CHECK = not os.environ.get("MSR3D_GEMM_LIB")
print("path", os.environ.get("MSR3D_BF16_GEMM", "(default)"))
for M, N, K, R in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    P = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).bfloat16()
    Q = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1).bfloat16()
    P2 = (torch.rand(M, max(R, 8), device="cuda", generator=g) * 2 - 1).bfloat16()
    Q2 = (torch.rand(N, max(R, 8), device="cuda", generator=g) * 2 - 1).bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    st = _lib.current_stream_ptr(torch.device("cuda"))

    def run():
        rc = lib.msr3d_bf16_gemm_lowrank(M, N, K, R, vp(P), K, vp(Q), K, vp(P2) if R else None, P2.shape[1],
                                         vp(Q2) if R else None, Q2.shape[1], vp(C), N, 0, ctypes.c_float(1.0), st)
        _lib.check(rc, "msr3d_bf16_gemm_lowrank")
    for _ in range(3):
        run()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):          # (back to back: the launch gap of a single timed kernel is ~10 % at these sizes)
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e2)
    ts.sort()
    t = ts[len(ts) // 2]
    want = P.float() @ Q.float().T + (P2[:, :R].float() @ Q2[:, :R].float().T if R else 0)
    err = float((C.float() - want).abs().max() / want.abs().max())
    print(f"M={M:5d} N={N:5d} K={K:5d} R={R:2d}  {t:8.1f} us  {2 * M * N * (K + R) / t / 1e6:7.1f} TFLOP/s  rel err {err:.2e}")
