"""Phase stamps of the distinct-row level-2 kernel (csrc/sa_split.hip::sa2_rows_kernel) on the bench's 960 clouds:
builds tools/prof/sa_rows_stamped.hip into tools/_prof/sa_rows_stamp.so (on the box: hipcc is in the image), runs the
encoder once through the shipping library to get level 2's real inputs, then the stamped level-2 entry on them, and prints
per-phase totals per block (s_memtime ticks of 10 ns -> us).
    python tools/prof_sa_rows.py [--batch 16]"""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
args = ap.parse_args()
so = os.path.join(ROOT, "tools/_prof/sa_rows_stamp.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC",
                       "-munsafe-fp-atomics", "-I", os.path.join(ROOT, "include"),
                       os.path.join(ROOT, "tools/prof/sa_rows_stamped.hip"), "-o", so])

from msr3d_amd.modules.layers.pointnet import PointNetPP  # noqa: E402
from msr3d_amd.pointnet2 import fused  # noqa: E402
from msr3d_amd.synth import synth_batch  # noqa: E402

torch.manual_seed(0)
net = PointNetPP(sa_n_points=[32, 16, None], sa_n_samples=[32, 32, None], sa_radii=[0.2, 0.4, None],
                 sa_mlps=[[3, 64, 64, 128], [128, 128, 128, 256], [256, 256, 512, 768]]).cuda().eval()
pts = synth_batch(10000, args.batch, device="cuda")["obj_fts"].reshape(-1, 1024, 6).contiguous()
with torch.no_grad():
    _, dbg = fused.forward(net, pts, return_internals=True)
b = pts.shape[0]
S = fused.get_plan(net)["split2"]
new1, feat1, new2, const = dbg["new_xyz1"], dbg["feat1"], dbg["new_xyz2"], dbg["constant"]
out = torch.empty(b, 16, 256, device="cuda")
ws = torch.empty(b * 1056 + b * 8 + 64, dtype=torch.uint8, device="cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lib = ctypes.CDLL(so)
fn = lib.msr3d_sa_level2_rows
fn.argtypes = [ctypes.c_int] * 3 + [ctypes.c_float] + [ctypes.c_void_p] * 14 + [ctypes.c_int, ctypes.c_void_p]
fn.restype = ctypes.c_int


def call():
    rc = fn(b, 32, 16, ctypes.c_float(0.4), p(new1), p(feat1), p(new2), p(S[0][0]), p(S[0][1]), p(S[1][0]), p(S[1][1]),
            p(S[2][0]), p(S[2][1]), p(out), p(None), p(None), p(const), p(ws), 0, st)
    assert rc == 0, rc


for _ in range(3):
    call()
torch.cuda.synchronize()
assert torch.equal(out, dbg["feat2"])
lib.msr3d_prof_rows_clear()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); call(); e1.record()
torch.cuda.synchronize()
print(f"stamped launch: {e0.elapsed_time(e1) * 1e3:.1f} ticks")
KL = 240
buf = np.zeros(512 * 8 * KL, np.uint64)
lib.msr3d_prof_rows_log.argtypes = [ctypes.c_void_p]
assert lib.msr3d_prof_rows_log(buf.ctypes.data) == 0
log = buf.reshape(512, 8, KL)
names = {4: "(chunk top)", 5: "rows -> operand + barrier F", 6: "layer 1 product + xyz", 7: "layer 1 epilogue (A, B)",
         8: "layer 2 product", 9: "layer 2 epilogue (C, D)", 10: "layer 3 product", 11: "segmented max",
         12: "next rows issued + barrier E", 13: "flush"}
tot = {}
spans, objs, chunks = [], 0, 0
us_per_tick = 1.0     # (s_memtime bases differ between XCDs: spans are per block; printed in ticks = shader cycles)
for blk in range(512):
    e = log[blk, 0]
    e = e[e != 0]
    if len(e) == 0:
        continue
    ids = (e >> np.uint64(56)).astype(int)
    t = (e & np.uint64((1 << 56) - 1)).astype(np.int64)
    spans.append((t[-1] - t[0]) * us_per_tick)
    objs += int((ids == 13).sum())
    chunks += int((ids == 4).sum())
    for i in range(1, len(e)):
        tot.setdefault(ids[i], []).append((t[i] - t[i - 1]) * us_per_tick)
print(f"blocks with work: {len(spans)}, objects logged {objs}, chunks {chunks}; per-block span median {np.median(spans):.1f} us max {np.max(spans):.1f} ticks")
for i in sorted(tot):
    v = np.array(tot[i])
    print(f"  {names.get(i, i):28s} n {len(v):6d}  median {np.median(v):7.2f} tk  mean {v.mean():7.2f}  sum/block {v.sum() / len(spans):7.1f} ticks")


# ---- level 1 (sa1_rows_kernel): marks 0 round top, 1 after layer 1, 2 point rows issued, 3 after layer 3, 4 after the maxima, 5 chunk barrier
fn1 = lib.msr3d_sa_level1_rows
fn1.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 13 + [ctypes.c_int, ctypes.c_void_p]
fn1.restype = ctypes.c_int
S1 = fused.get_plan(net)["split1"]
ball1, feat1_ref = dbg["ball1"], dbg["feat1"]
out1 = torch.empty(b, 32, 128, device="cuda")
ws1 = torch.empty(b * 32 * 132 + 64, dtype=torch.uint8, device="cuda")


def call1():
    rc = fn1(b, 1024, 32, p(pts), p(new1), p(ball1), p(S1[0][0]), p(S1[0][1]), p(S1[1][0]), p(S1[1][1]), p(S1[2][0]), p(S1[2][1]),
             p(out1), p(None), p(const), p(ws1), 0, st)
    assert rc == 0, rc


for _ in range(3):
    call1()
torch.cuda.synchronize()
assert torch.equal(out1, feat1_ref)
lib.msr3d_prof_rows_clear()
e0.record(); call1(); e1.record()
torch.cuda.synchronize()
print(f"level 1 stamped launch (plan + products): {e0.elapsed_time(e1) * 1e3:.1f} us")
assert lib.msr3d_prof_rows_log(buf.ctypes.data) == 0
log = buf.reshape(512, 8, KL)
n1 = {0: "chunk hand-over", 1: "operand + layer 1", 2: "issue point rows", 3: "layers 2, 3", 4: "maxima + stores", 5: "chunk barrier"}
tot, spans, rounds = {}, [], 0
for blk in range(256):
    for w in range(8):
        e = log[blk, w]
        e = e[e != 0]
        if len(e) < 2:
            continue
        ids = (e >> np.uint64(56)).astype(int)
        t = (e & np.uint64((1 << 56) - 1)).astype(np.int64)
        spans.append(t[-1] - t[0])
        rounds += int((ids == 0).sum())
        for i in range(1, len(e)):
            tot.setdefault((ids[i - 1], ids[i]), []).append(t[i] - t[i - 1])
print(f"waves logged {len(spans)}, rounds {rounds}; per-wave span median {np.median(spans):.0f} max {np.max(spans):.0f} ticks")
for kk in sorted(tot):
    v = np.array(tot[kk])
    print(f"  {kk[0]}->{kk[1]} {n1.get(kk[1], '')[:24]:24s} n {len(v):6d}  median {np.median(v):8.0f}  mean {v.mean():8.0f}  sum/wave {v.sum() / len(spans):9.0f}")
