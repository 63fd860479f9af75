"""Micro-benchmark of msr3d_gemm_f32 on the shapes of the trainable part.
    python tools/bench_gemm.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from msr3d_amd import hipops  # noqa: E402

SHAPES = [  # name, a_kc, b_kc, M, N, K
    ("qkvc fwd  NT", 1, 1, 960, 816, 256), ("fc fwd    NT", 1, 1, 960, 256, 256),
    ("ffn1 fwd  NT", 1, 1, 960, 2048, 256), ("ffn2 fwd  NT", 1, 1, 960, 256, 2048),
    ("proj fwd  NT", 1, 1, 960, 4096, 256), ("ffn1 dx   NN", 1, 0, 960, 256, 2048),
    ("ffn2 dx   NN", 1, 0, 960, 2048, 256), ("proj dx   NN", 1, 0, 960, 256, 4096),
    ("ffn1 dW   TN", 0, 0, 2048, 256, 960), ("ffn2 dW   TN", 0, 0, 256, 2048, 960),
    ("proj dW   TN", 0, 0, 4096, 256, 960), ("qkvc dW   TN", 0, 0, 816, 256, 960),
    ("steady    NT", 1, 1, 960, 2048, 4096),
]
for name, akc, bkc, M, N, K in SHAPES:
    A = torch.randn((M, K) if akc else (K, M), device="cuda")
    B = torch.randn((N, K) if bkc else (K, N), device="cuda")
    C = torch.empty((M, N), device="cuda")
    lda = K if akc else M
    ldb = K if bkc else N
    for _ in range(3):
        hipops._gemm(akc, bkc, M, N, K, A, lda, B, ldb, C, N)
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hipops._gemm(akc, bkc, M, N, K, A, lda, B, ldb, C, N)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    t = ts[len(ts) // 2]
    print(f"{name}  M={M:5d} N={N:5d} K={K:5d}  {t:7.1f} us  {2 * M * N * K / t / 1e6:7.1f} TFLOP/s")
