"""Build libmsr3d_hip.so (gfx950) in-tree with hipcc.  `python -m msr3d_amd.build`.

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels
to the GPU box with the tree.  Objects are cached under msr3d_amd/csrc/build/ and
rebuilt when the source (or a header) is newer.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libmsr3d_hip.so")
# the labelled reduced-split variant of the set-abstraction SharedMLPs (MSR3D_SA_MMA=split2): csrc/sa_split.hip alone,
# compiled with MSR3D_SPLIT_TERMS=3, under the same entry names
LIB_SPLIT2 = os.path.join(HERE, "libmsr3d_hip_split2.so")
# the labelled bf16 variant of the trainable part (MSR3D_TRAIN_MMA=bf16): csrc/scene_block.hip + csrc/wgrad_split.hip
# compiled with MSR3D_TRAIN_PLANES=1 (one bf16 MFMA product per product instead of six), under the same entry names
LIB_BF16 = os.path.join(HERE, "libmsr3d_hip_bf16.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
          "-Wall", "-Wno-unused-function", "-I", INCLUDE]

# (source, extra flags).  The index ops pin their fma chains explicitly, so the
# compiler must not contract anything further there.
SOURCES = [
    ("pn2_ops.hip", ["-ffp-contract=off"]),
    ("sa_fused.hip", ["-ffp-contract=off"]),
    ("sa_split.hip", ["-ffp-contract=off"]),
    ("gemm_f32.hip", []),
    ("attn_spatial.hip", []),
    ("optim_flat.hip", ["-ffp-contract=off"]),
    ("rowops.hip", []),
    ("prologue.hip", ["-ffp-contract=off"]),
    ("scene_scatter.hip", []),
    ("preprocess.hip", ["-ffp-contract=off"]),
    ("bn_train.hip", []),
    ("strip_gemm.hip", []),
    ("panel_gemm.hip", []),
    ("seq_ce.hip", []),
    ("lora_linear.hip", []),
    ("lora_fp8.hip", []),
    ("llm_layer.hip", []),
    ("llm_attn.hip", []),
    ("prompter_rows.hip", ["-ffp-contract=off"]),
    ("anchor_front.hip", ["-ffp-contract=off"]),
    ("scene_block.hip", []),
    ("scene_rows.hip", []),
    ("wgrad_split.hip", []),
    ("rows_gemm_split.hip", []),
    ("rows_linear.hip", []),
]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: cannot build libmsr3d_hip.so")


def _newest_header():
    t = 0.0
    for d in (CSRC, INCLUDE):
        for f in os.listdir(d):
            if f.endswith((".h", ".hpp")):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return max(t, os.path.getmtime(os.path.abspath(__file__)))


def build(force=False, verbose=False, sqdist_contract=None):
    """sqdist_contract: rebuild the index kernels under another floating-point contract of the squared
    distance (csrc/pn2_device.h; 0 = default).  The choice is recorded in csrc/build/.sqdist_contract so a
    later plain build() keeps it until it is changed back."""
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    mark = os.path.join(OBJ, ".sqdist_contract")
    current = int(open(mark).read().strip()) if os.path.exists(mark) else 0
    if sqdist_contract is None:
        sqdist_contract = current
    if sqdist_contract != current:
        force = True
    with open(mark, "w") as f:
        f.write(str(int(sqdist_contract)))
    contract_flag = ["-DMSR3D_SQDIST_CONTRACT=%d" % int(sqdist_contract)]
    hdr_t = _newest_header()
    objs, rebuilt = [], False
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_t):
            cmd = [cc] + COMMON + extra + contract_flag + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            rebuilt = True
    if rebuilt or not os.path.exists(LIB):
        cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    s2 = os.path.join(CSRC, "sa_split.hip")
    if force or not os.path.exists(LIB_SPLIT2) or os.path.getmtime(LIB_SPLIT2) < max(os.path.getmtime(s2), hdr_t):
        cmd = [cc] + COMMON + ["-ffp-contract=off", "-DMSR3D_SPLIT_TERMS=3"] + contract_flag + ["-shared", s2, "-o", LIB_SPLIT2]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    srcs = [os.path.join(CSRC, "scene_block.hip"), os.path.join(CSRC, "wgrad_split.hip")]
    if force or not os.path.exists(LIB_BF16) or os.path.getmtime(LIB_BF16) < max([os.path.getmtime(x) for x in srcs] + [hdr_t]):
        cmd = [cc] + COMMON + ["-DMSR3D_TRAIN_PLANES=1", "-shared"] + srcs + ["-o", LIB_BF16]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    contract = None
    if "--sqdist-contract" in sys.argv:
        contract = int(sys.argv[sys.argv.index("--sqdist-contract") + 1])
    print(build(force="--force" in sys.argv, verbose=True, sqdist_contract=contract))
