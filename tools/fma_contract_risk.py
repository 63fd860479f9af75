"""How much does the unpinnable floating-point contract of the index ops matter?

The reference's FPS / ball query compute the squared distance as `x*x + y*y + z*z` in CUDA source
(_ext_src/src/sampling_gpu.cu:99-104, ball_query_gpu.cu:32-35); which of the products nvcc fuses into fma
is not recoverable without an NVIDIA toolchain, and this build's oracle + kernels pin ONE choice (contract 0,
what an LLVM device compiler emits).  This script re-runs the oracle's FPS and ball query under every
plausible contraction (oracle/pn2.py: CONTRACTS) on

  * the bench batch (16 scenes x 60 objects x 1024 points = 960 clouds, bench.py's rank-0 batch 0), both
    set-abstraction levels, and the frozen encoder's output for each choice;
  * the tie-heavy test clouds ("dup", "grid", "mixed" of tests/test_pn2_ops_gpu.py),

plus, as an envelope, contract 0 with EVERY squared distance pushed one ulp up / down (4, 5: every decision that
any rounding difference could flip, flipped together), and reports, against contract 0: the fraction of clouds whose index vectors differ at all, the fraction of
differing index entries, and the relative L2 distance of the encoder output (PcdObjEncoder, eval, torch-CPU
mirror driven by the oracle).  Runs on the CPU.   python tools/fma_contract_risk.py [--scenes 16] [--out F]
"""
import argparse
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pn2  # noqa: E402


def unit_ball_cloud(rng, b, n, dup_frac=0.0):
    x = rng.standard_normal((b, n, 3)).astype(np.float32)
    x /= np.maximum(np.linalg.norm(x, axis=-1, keepdims=True), 1e-6)
    x *= rng.uniform(0.05, 1.0, (b, n, 1)).astype(np.float32) ** (1 / 3)
    if dup_frac:
        k = int(n * dup_frac)
        src = rng.integers(0, n, (b, k))
        dst = rng.integers(0, n, (b, k))
        for i in range(b):
            x[i, dst[i]] = x[i, src[i]]
    return x


def test_clouds(kind, rng, b=64, n=1024):
    if kind == "dup":
        return unit_ball_cloud(rng, b, n, dup_frac=0.7)
    if kind == "grid":
        return rng.integers(-3, 4, (b, n, 3)).astype(np.float32) * 0.25
    x = unit_ball_cloud(rng, b, n, dup_frac=0.5)          # "mixed"
    x[:, ::3] *= 0.02
    return x


def index_ops(xyz, levels=((32, 0.2, 32), (16, 0.4, 32))):
    """Both set-abstraction levels' FPS + ball query under the oracle's current contract."""
    out = []
    cur = xyz
    for m, r, ns in levels:
        if cur.shape[1] < m:
            break
        idx = pn2.furthest_point_sampling(cur, m)
        new = np.take_along_axis(cur, idx[..., None].astype(np.int64).repeat(3, -1), 1)
        ball = pn2.ball_query(new, cur, r, ns)
        out += [idx, ball]
        cur = new
    return out


def diff_stats(a, b):
    clouds = np.any(a.reshape(a.shape[0], -1) != b.reshape(b.shape[0], -1), axis=1)
    return float(clouds.mean()), float((a != b).mean())


def encoder_out(enc, obj_fts):
    from msr3d_amd.pointnet2 import pointnet2_utils
    saved = pointnet2_utils._ext
    pointnet2_utils._ext = pn2.ext_module()
    try:
        with torch.no_grad():
            y, _ = enc(obj_fts)
    finally:
        pointnet2_utils._ext = saved
    return y.double().numpy()


def run(scenes=16, test_b=64, encoder=True, random_clouds=0):
    import msr3d_amd.model  # noqa: F401
    import msr3d_amd.modules  # noqa: F401
    from msr3d_amd.config import AttrDict, default_prompter_cfg
    from msr3d_amd.model import build_model
    from msr3d_amd.synth import synth_batch
    batch = synth_batch(0, scenes, O=60, P=1024)            # bench.py: synth_batch(1000 * rank + i, 16, ...), rank 0, i 0
    fts = batch["obj_fts"]
    xyz = fts.reshape(-1, 1024, 6)[..., :3].contiguous().numpy()
    enc = None
    if encoder:
        torch.manual_seed(0)
        cfg = AttrDict({"prompter": default_prompter_cfg(), "llm_hidden_size": 256, "model": {"name": "MSR3DHotPath"}})
        enc = copy.deepcopy(build_model(cfg).visual_prompter.obj_encoder).eval()
    rng = np.random.default_rng(2024)
    tests = {k: test_clouds(k, rng, b=test_b) for k in ("dup", "grid", "mixed")}
    if random_clouds:       # a larger sample of decisions: Gaussian clouds at the bench shape
        tests["random"] = (rng.standard_normal((random_clouds, 1024, 3)) * 0.4).astype(np.float32)
    res = {"clouds": int(xyz.shape[0]), "random_clouds": int(random_clouds),
           "decisions_per_cloud": {"fps_argmax": 31 + 15, "ball_tests": 32 * 1024 + 16 * 32}, "contracts": {}}
    base = {}
    names = ["fps1", "ball1", "fps2", "ball2"]
    try:
        for c in (0, 1, 2, 3, 4, 5):
            pn2.set_contract(c)
            cur = {"bench": index_ops(xyz)}
            for k, v in tests.items():
                cur[k] = index_ops(v)
            y = encoder_out(enc, fts) if enc is not None else None
            if c == 0:
                base, y0 = cur, y
                continue
            entry = {"contract": pn2.CONTRACTS[c]}
            for k in cur:
                for nme, a, b in zip(names, cur[k], base[k]):
                    fc, fe = diff_stats(a, b)
                    entry[f"{k}/{nme}"] = {"clouds_differing": fc, "entries_differing": fe}
            if y is not None:
                entry["enc_out_rel_l2"] = float(np.linalg.norm(y - y0) / np.linalg.norm(y0))
                per = np.linalg.norm((y - y0).reshape(-1, y.shape[-1]), axis=1) / np.linalg.norm(y0.reshape(-1, y.shape[-1]), axis=1)
                entry["enc_out_objects_changed"] = float((per > 0).mean())
                entry["enc_out_max_object_rel_l2"] = float(per.max())
            res["contracts"][str(c)] = entry
    finally:
        pn2.set_contract(0)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=16)
    ap.add_argument("--random", type=int, default=4000, help="additional Gaussian clouds (index ops only)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_fma_contract_risk.json"))
    a = ap.parse_args()
    res = run(a.scenes, random_clouds=a.random)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print(f"{res['clouds']} bench clouds; differences against contract 0 = {pn2.CONTRACTS[0]}")
    for c, e in res["contracts"].items():
        print(f"contract {c}: {e['contract']}")
        for k, v in e.items():
            if isinstance(v, dict):
                print(f"    {k:14s} clouds {100 * v['clouds_differing']:7.3f} %   entries {100 * v['entries_differing']:8.4f} %")
        if "enc_out_rel_l2" in e:
            print(f"    encoder output rel-L2 {e['enc_out_rel_l2']:.3e}; objects changed {100 * e['enc_out_objects_changed']:.2f} %, "
                  f"worst object {e['enc_out_max_object_rel_l2']:.3e}")


if __name__ == "__main__":
    main()
