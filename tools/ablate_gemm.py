"""Main-loop ablation of gemm_f32_kernel.  Build, from msr3d_amd/csrc:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I ../../include \\
        -DMSR3D_GEMM_ABLATE -shared -o ../../tools/_prof/libgemm_ablate.so gemm_f32.hip
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_prof", "libgemm_ablate.so"))
M, N, K = 960, 2048, 4096
A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
names = {0: "full", 1: "no global loads", 2: "no MFMA/LDS reads", 4: "no LDS stores", 5: "no loads+stores", 8: "no barrier (wrong)", 13: "MFMA+LDS reads only", 3: "skeleton: stores+barrier only"}
for abl, nm in names.items():
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.msr3d_gemm_f32(1, 1, M, N, K, p(A), K, p(B), K, p(C), N, None, None, abl << 8, ctypes.c_float(0.0), ctypes.c_float(0.0), None, 0, None, ctypes.c_size_t(0), st)
        e1.record(); torch.cuda.synchronize(); assert rc == 0
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = sorted(ts)[len(ts) // 2]
    print(f"{nm:32s} {t:8.1f} us   ({2*M*N*K/t/1e6:6.1f} TF-equivalent)")
