"""Phase stamps of panel_gemm.hip's workgroups (the library rebuilt into tools/_prof/ with
tools/prof/panel_gemm_stamped.hip in place of csrc/panel_gemm.hip):
where a workgroup's life goes -- B-panel issue, A staging + barrier, MFMA phase, epilogue."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from msr3d_amd import build as hb

out = os.path.join(ROOT, "tools", "_prof")
os.makedirs(out, exist_ok=True)
lib_path = os.path.join(out, "libmsr3d_prof.so")
if not os.path.exists(lib_path) or "--rebuild" in sys.argv:
    objs = []
    for src, extra in hb.SOURCES:
        o = os.path.join(out, src.replace(".hip", ".o"))
        path = os.path.join(ROOT, "tools", "prof", "panel_gemm_stamped.hip") if src == "panel_gemm.hip" else os.path.join(hb.CSRC, src)
        subprocess.check_call([hb.hipcc()] + hb.COMMON + extra + ["-c", path, "-o", o])
        objs.append(o)
    subprocess.check_call([hb.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs)
if "--build-only" in sys.argv:
    sys.exit(0)
from msr3d_amd import _lib
_lib.LIB_PATH = lib_path
lib = _lib.load()
dev = torch.device("cuda", 0)
M, D, FF = 960, 256, 2048
t = lambda *s: torch.randn(*s, device=dev)
cases = {
    "dx1 (a_kc, b_kr) K=2048": dict(a_kc=1, b_kc=0, M=M, N=D, K=FF, A=t(M, FF), lda=FF, B=t(FF, D), ldb=D, C=torch.zeros(M, D, device=dev), ldc=D, beta=1.0),
    "ffn2 (a_kc, b_kc) K=2048": dict(a_kc=1, b_kc=1, M=M, N=D, K=FF, A=t(M, FF), lda=FF, B=t(D, FF), ldb=FF, C=torch.zeros(M, D, device=dev), ldc=D, beta=1.0),
    "dW1 (a_kr, b_kr) K=960": dict(a_kc=0, b_kc=0, M=FF, N=D, K=M, A=t(M, FF), lda=FF, B=t(M, D), ldb=D, C=torch.zeros(FF, D, device=dev), ldc=D, beta=1.0),
}
for name, kw in cases.items():
    arr = (_lib.GemmProblem * 1)()
    for k, v in kw.items():
        setattr(arr[0], k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    st = _lib.current_stream_ptr(dev)
    for _ in range(3):
        lib.msr3d_gemm_multi_f32(1, arr, st)
    torch.cuda.synchronize()
    tiles, stages = ((kw["N"] + 63) // 64) * ((kw["M"] + 63) // 64), (kw["K"] + 127) // 128
    gz = max(1, min((240 + tiles - 1) // tiles, stages // 2))
    spw = (stages + gz - 1) // gz
    nb = tiles * ((stages + spw - 1) // spw)
    buf = (ctypes.c_ulonglong * (8 * nb))()
    lib.msr3d_prof_panel_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rc = lib.msr3d_prof_panel_stamps(buf, nb); assert rc == 0
    s = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
    t0 = s[:, 0].min()
    ph = np.diff(s[:, :5], axis=1)
    print(f"{name}: {nb} workgroups; start spread {np.percentile(s[:,0]-t0,[50,95,100])} cyc; "
          f"kernel span {s[:,4].max()-t0} cyc")
    for i, lab in enumerate(["operand issue", "first A staged", "all stages", "epilogue"]):
        print(f"   {lab:16s} median {np.median(ph[:, i]):8.0f}  p95 {np.percentile(ph[:, i], 95):8.0f} cycles")
