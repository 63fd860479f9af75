"""Throughput of the device-side sample construction (msr3d_preprocess_pcd) at BASELINE sizes
(16 scenes x 60 objects x 1024 points), against its HBM roofline, with the oracle's numpy
restatement of the reference's host path timed beside it.

    python tools/bench_preprocess.py [--batch 16] [--iters 50] [--scan-points 150000] [--no-cpu]

Algorithmic bytes per launch: pass 1 reads every point of every selected object
once (12 B xyz; colours are not touched), pass 2 gathers P points (12 B xyz + 3 B rgb) and
writes P x 24 B, per object; padding slots write P x 24 B.
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--scan-points", type=int, default=150000)
    ap.add_argument("--scans", type=int, default=8)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--host-idx", action="store_true", help="feed caller-drawn indices (skips the device draw)")
    args = ap.parse_args()
    from msr3d_amd.data import SceneInputBuilder, SceneStore
    from msr3d_amd.synth import synth_scan

    rng = np.random.default_rng(0)
    st = SceneStore("cuda")
    host_scans = []
    t0 = time.perf_counter()
    for s in range(args.scans):
        pts, col, lab = synth_scan(rng, 60 + 5 * s, args.scan_points)
        host_scans.append((pts, col, lab))
    for s, (pts, col, lab) in enumerate(host_scans):
        st.add_scan(f"scan{s}", pts, col, lab)
    torch.cuda.synchronize()
    samples = [{"scan_id": f"scan{b % args.scans}", "insts": [1, 2, 3],
                "situation": (np.zeros(3), np.array([0, 0, 0, 1.0]))} for b in range(args.batch)]
    bld = SceneInputBuilder(st)
    random.seed(0)
    out = bld.build(samples)
    torch.cuda.synchronize()

    # algorithmic bytes of one launch
    O, P = bld.max_obj_len, bld.num_points
    nbytes = 0
    for smp in samples:
        sel = bld.select_objects(smp["scan_id"], smp["insts"])
        nbytes += sum(st.scans[smp["scan_id"]]["count"][i] for i in sel) * 12
        nbytes += len(sel) * P * (15 + 24) + (O - len(sel)) * P * 24

    # whole build() (host selection + descriptor upload + launch) and the kernel alone
    t0 = time.perf_counter()
    for _ in range(args.iters):
        out = bld.build(samples, out=out)
    torch.cuda.synchronize()
    t_build = (time.perf_counter() - t0) / args.iters
    sel = [bld.select_objects(s["scan_id"], s["insts"]) for s in samples]
    rots = [None] * args.batch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    # kernel time: HIP events on torch's current stream (the stream the launch goes to)
    import ctypes
    from msr3d_amd import _lib
    from msr3d_amd.data.scene_store import _p
    lib = _lib.load()
    begin = np.zeros((args.batch, O), np.int64)
    count = np.zeros((args.batch, O), np.int32)
    for b, smp in enumerate(samples):
        for o, i in enumerate(sel[b]):
            begin[b, o] = st.scans[smp["scan_id"]]["begin"][i]
            count[b, o] = st.scans[smp["scan_id"]]["count"][i]
    d_begin, d_count = torch.from_numpy(begin).cuda(), torch.from_numpy(count).cuda()
    stream = _lib.current_stream_ptr(torch.device("cuda"))

    d_idx = None
    if args.host_idx:
        d_idx = torch.from_numpy((rng.integers(0, 1 << 30, (args.batch, O, P)) % np.maximum(count, 1)[..., None])
                                 .astype(np.int32)).cuda()

    def launch(seed):
        rc = lib.msr3d_preprocess_pcd(args.batch, O, P, _p(st.points), _p(st.colors), _p(d_begin), _p(d_count),
                                      None, _p(d_idx), ctypes.c_ulonglong(seed), _p(out["obj_fts"]),
                                      _p(out["obj_locs"]), _p(out["obj_masks"]), None, stream)
        assert rc == 0
    for i in range(5):
        launch(i)
    ev[0].record()
    for i in range(args.iters):
        launch(100 + i)
    ev[1].record()
    torch.cuda.synchronize()
    t_kernel = ev[0].elapsed_time(ev[1]) / args.iters * 1e-3

    res = {
        "metric": "scene samples constructed / s (60 obj x 1024 pts)", "unit": "samples/s",
        "value": args.batch / t_build, "kernel_only_value": args.batch / t_kernel,
        "ms_per_batch": t_build * 1e3, "kernel_us": t_kernel * 1e6, "batch": args.batch,
        "store_bytes": st.nbytes(),
        "roofline": {"bound": "hbm", "achieved": nbytes / t_kernel / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": nbytes / t_kernel / 8e12, "traffic": None,
                     "algorithmic_bytes_per_launch": nbytes},
    }
    if not args.no_cpu:
        from oracle import sample_input as si
        # the reference's host path for ONE sample: objects are cached per scan (as the reference
        # caches them); timed: rotate/box/subsample/normalise + padding (msr3d.py:181-216)
        pts, col, lab = host_scans[0]
        pcds = si.scan_to_pcds(pts, col)
        kept = st.inst_ids("scan0")
        t0 = time.perf_counter()
        objs_all = {i: pcds[lab == i] for i in kept}
        t_seg = time.perf_counter() - t0
        sel0 = sel[0]
        reps, t0 = 5, time.perf_counter()
        for _ in range(reps):
            objs = [objs_all[i] for i in sel0]
            idxs = [np.random.choice(len(o), size=P, replace=len(o) < P) for o in objs]
            fts, locs = si.preprocess_pcd(objs, idxs, si.rotate_mat(np.pi / 2))
            si.pad_sample(fts, locs, O)
        t_cpu = (time.perf_counter() - t0) / reps
        res["cpu_baseline"] = {"value": 1.0 / t_cpu, "unit": "samples/s", "cores": 1, "kind": "port",
                               "sample": f"{reps} samples of one 60-object scene; per-scan instance masks "
                                         f"({t_seg * 1e3:.0f} ms, cached by the reference) not included"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
