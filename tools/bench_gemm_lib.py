"""The same shapes as tools/bench_gemm.py through torch (rocBLAS / hipBLASLt), for comparison.
    python tools/bench_gemm_lib.py"""
import torch

SHAPES = [  # name, expression
    ("qkvc fwd  NT", (960, 816, 256), "nt"), ("fc fwd    NT", (960, 256, 256), "nt"),
    ("ffn1 fwd  NT", (960, 2048, 256), "nt"), ("ffn2 fwd  NT", (960, 256, 2048), "nt"),
    ("proj fwd  NT", (960, 4096, 256), "nt"), ("ffn1 dx   NN", (960, 256, 2048), "nn"),
    ("ffn2 dx   NN", (960, 2048, 256), "nn"), ("proj dx   NN", (960, 256, 4096), "nn"),
    ("ffn1 dW   TN", (2048, 256, 960), "tn"), ("ffn2 dW   TN", (256, 2048, 960), "tn"),
    ("proj dW   TN", (4096, 256, 960), "tn"), ("qkvc dW   TN", (816, 256, 960), "tn"),
]
for name, (M, N, K), kind in SHAPES:
    if kind == "nt":
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        f = lambda: torch.mm(A, B.t())
    elif kind == "nn":
        A, B = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
        f = lambda: torch.mm(A, B)
    else:
        A, B = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
        C = torch.zeros(M, N, device="cuda")
        f = lambda: C.addmm_(A.t(), B)
    for _ in range(5):
        f()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    t = ts[len(ts) // 2]
    print(f"{name}  M={M:5d} N={N:5d} K={K:5d}  {t:7.1f} us  {2 * M * N * K / t / 1e6:7.1f} TFLOP/s")
