"""Time msr3d_fp8_gemm_lowrank over K at the language model's shapes: the intercept is a launch's fixed cost.
python tools/prof_fp8_gemm.py"""
import sys

import torch

sys.path.insert(0, ".")
from msr3d_amd.llm.lora import PAD_R, _gemm_fp8, quant_rows_fp8  # noqa: E402

dev = torch.device("cuda:0")
for M, N in ((2304, 4096), (2304, 11008), (11520, 4096)):
    for K in (128, 1024, 2048, 4096, 8192, 11008):
        for lora in (False, True):
            x = torch.randn(M, K, device=dev).bfloat16()
            w = torch.randn(N, K, device=dev).bfloat16()
            xq, sx = quant_rows_fp8(x)
            wq, sw = quant_rows_fp8(w)
            u = torch.randn(M, PAD_R, device=dev).bfloat16() if lora else None
            b2 = torch.randn(N, PAD_R, device=dev).bfloat16() if lora else None
            y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            for _ in range(3):
                _gemm_fp8(M, N, K, xq, sx, wq, sw, u, b2, y, dev)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                _gemm_fp8(M, N, K, xq, sx, wq, sw, u, b2, y, dev)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            print(f"M={M} N={N} K={K:5d} lora={int(lora)}  {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TFLOP/s", flush=True)
