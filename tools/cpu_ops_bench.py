"""Per-op micro-benchmarks at the GPU kernels' shapes: the CPU rows beside the GPU kernel
rows of DESIGN.md §6.  `bench.py --cpu-ops` runs this and writes profiles/r03_cpu_ops.json.

One scene = 60 objects x 1024 points (b = 60), configs/msr3d.yaml shapes:
    fps          furthest_point_sampling   (60, 1024, 3) -> (60, 32)
    ball_query   r = 0.2, nsample 32       32 centres x 1024 points per object
    group        group_points              (60, 6, 1024) by (60, 32, 32) -> (60, 6, 32, 32)
    sa1 / sa2 / sa3 + fc                   grouping + SharedMLP (eval BatchNorm) + max, whole level
    spatial layer                          one TransformerSpatialEncoderLayer forward + backward, 60 tokens
    projector                              llm_proj 256 -> 4096 forward + backward, 60 tokens

CPU: the C oracle (OpenMP over objects) for the index ops, torch-CPU for the rest, `threads` threads and one
thread.  GPU (when present): the shipped kernels on the same tensors, HIP events, per scene of 60 objects
AND per 16 scenes (the bench's launch shape; the kernels are sized for that).  The oracle is used here as
the timed CPU baseline only (cpu_baseline leg), never by the product."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _median(f, reps, warm=2):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3


def _gpu_ms(f, reps=20, warm=3):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run(threads=32, reps=7, gpu=None):
    import msr3d_amd.model  # noqa: F401
    import msr3d_amd.modules  # noqa: F401
    from msr3d_amd.config import AttrDict, default_prompter_cfg
    from msr3d_amd.model import build_model
    from msr3d_amd.pointnet2 import pointnet2_utils
    from msr3d_amd.synth import synth_batch
    from oracle import pn2
    gpu = torch.cuda.is_available() if gpu is None else gpu
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(threads, avail))
    torch.manual_seed(0)
    cfg = AttrDict({"prompter": default_prompter_cfg(dropout=0.0), "llm_hidden_size": 4096,
                    "model": {"name": "MSR3DHotPath"}})
    model = build_model(cfg)
    enc = model.visual_prompter.obj_encoder.eval()
    net = enc.pcd_net
    batch = synth_batch(0, 1, O=60, P=1024)
    pts = batch["obj_fts"].reshape(-1, 1024, 6).contiguous()
    xyz = pts[..., :3].contiguous()
    xyz_np = xyz.numpy()
    feats = pts.transpose(1, 2).contiguous()                      # (60, 6, 1024)
    idx1 = pn2.furthest_point_sampling(xyz_np, 32)
    new1 = np.take_along_axis(xyz_np, idx1[..., None].astype(np.int64).repeat(3, -1), 1)
    ball1 = pn2.ball_query(new1, xyz_np, 0.2, 32)
    layer = model.visual_prompter.spatial_encoder[0]
    tok = torch.randn(1, 60, 256)
    pw = torch.randn(1, 60, 60, 5)
    mask = torch.zeros(1, 60, dtype=torch.bool)
    rows = {}
    saved = pointnet2_utils._ext
    pointnet2_utils._ext = pn2.ext_module()
    try:
        for label, nt in (("cpu_ms", threads), ("cpu_1thread_ms", 1)):
            torch.set_num_threads(nt)
            pn2.set_threads(nt)
            r = rows.setdefault
            r("fps (60 x 1024 -> 32)", {})[label] = _median(lambda: pn2.furthest_point_sampling(xyz_np, 32), reps)
            r("ball_query (32 x 1024, r .2, ns 32)", {})[label] = _median(lambda: pn2.ball_query(new1, xyz_np, 0.2, 32), reps)
            r("group_points (6 ch)", {})[label] = _median(lambda: pn2.group_points(feats.numpy(), ball1), reps)
            with torch.no_grad():
                sa = net.encoder
                x1, f1 = sa[0](xyz, feats[:, 3:].contiguous())
                x2, f2 = sa[1](x1, f1)
                r("sa1: FPS + ball + group + MLP 6-64-64-128 + max", {})[label] = _median(lambda: sa[0](xyz, feats[:, 3:].contiguous()), reps)
                r("sa2: FPS + ball + group + MLP 131-128-128-256 + max", {})[label] = _median(lambda: sa[1](x1, f1), reps)
                r("sa3 (group-all) + fc: MLP 259-256-512-768, 768-768", {})[label] = _median(lambda: net.fc(sa[2](x2, f2)[1].squeeze(-1)), reps)

            def layer_step():
                t = tok.clone().requires_grad_(True)
                y = layer(t, tgt_pairwise_locs=pw, tgt_key_padding_mask=mask)
                (y[0] if isinstance(y, tuple) else y).sum().backward()
            r("spatial layer fwd+bwd (60 tokens)", {})[label] = _median(layer_step, reps)

            def proj_step():
                t = tok.clone().requires_grad_(True)
                model.llm_proj(t).sum().backward()
            r("llm_proj 256 -> 4096 fwd+bwd (60 tokens)", {})[label] = _median(proj_step, reps)
    finally:
        pointnet2_utils._ext = saved
        torch.set_num_threads(threads)
    if gpu:
        from msr3d_amd.pointnet2 import _ext
        dev = torch.device("cuda", 0)
        for scenes in (1, 16):
            b = synth_batch(0, scenes, O=60, P=1024, device=dev)
            p = b["obj_fts"].reshape(-1, 1024, 6).contiguous()
            xg = p[..., :3].contiguous()
            fg = p.transpose(1, 2).contiguous()
            ig = _ext.furthest_point_sampling(xg, 32)
            ng = torch.gather(xg, 1, ig.long()[..., None].expand(-1, -1, 3)).contiguous()
            bg = _ext.ball_query(ng, xg, 0.2, 32)
            key = "gpu_ms_1scene" if scenes == 1 else "gpu_ms_16scenes"
            rows["fps (60 x 1024 -> 32)"][key] = _gpu_ms(lambda: _ext.furthest_point_sampling(xg, 32))
            rows["ball_query (32 x 1024, r .2, ns 32)"][key] = _gpu_ms(lambda: _ext.ball_query(ng, xg, 0.2, 32))
            rows["group_points (6 ch)"][key] = _gpu_ms(lambda: _ext.group_points(fg, bg))
            g_enc = model.visual_prompter.obj_encoder.to(dev).eval()
            with torch.no_grad():
                rows.setdefault("whole frozen encoder (fused: fps, ball, sa1-3, fc)", {})[key] = _gpu_ms(
                    lambda: g_enc(b["obj_fts"]))
                # ... and per level: HIP events around each launch of the fused path (msr3d_amd/_lib.py: kernel_timer).
                # The two FPS levels are ONE launch (msr3d_sa_fps2) and go with level 1; level 1's figure includes its
                # ball-query launch, level 2's query runs inside its kernel; level 3 = the group-all kernel (the fc
                # GEMM is the difference to the whole-encoder row)
                from msr3d_amd import _lib
                sink = {k: [] for k in ("msr3d_sa_fps2", "msr3d_sa_level1", "msr3d_sa_level2", "msr3d_sa_level3")}
                for _ in range(3):
                    g_enc(b["obj_fts"])
                torch.cuda.synchronize()
                _lib.set_timing_sink(sink)
                for _ in range(20):
                    g_enc(b["obj_fts"])
                torch.cuda.synchronize()
                _lib.set_timing_sink(None)
                ms = {k: sum(a.elapsed_time(e) for a, e in v) / max(len(v), 1) for k, v in sink.items()}
                rows["sa1: FPS + ball + group + MLP 6-64-64-128 + max"][key] = ms["msr3d_sa_fps2"] + ms["msr3d_sa_level1"]
                rows["sa2: FPS + ball + group + MLP 131-128-128-256 + max"][key] = ms["msr3d_sa_level2"]
                rows["sa3 (group-all) + fc: MLP 259-256-512-768, 768-768"][key] = ms["msr3d_sa_level3"]
            model.visual_prompter.obj_encoder.to("cpu")
    host = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                host = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"threads": threads, "host": {"cpu": host, "hw_threads": os.cpu_count(), "threads_available": avail},
            "shape": "one scene: 60 objects x 1024 points (CPU rows); GPU rows per 1 and per 16 scenes",
            "rows": rows}
