#!/usr/bin/env python
"""bench.py -- MSQA hot-path training throughput on N MI355X of one node.

A "step" is one pass of the hot path over one batch of synthetic scenes, resident in
HBM before the timed region:
    obj_fts -> PointNet++ set abstraction (frozen) -> situated spatial-attention encoder
    -> llm_proj -> synthetic scalar loss on the projector output -> backward
    -> (N>1) RCCL gradient all-reduce overlapped with backward -> global-norm clip -> AdamW
(SURVEY.md §8(d) "primary metric").  The frozen LLM is NOT part of this path.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
prints ONE JSON line on rank 0.  For N > 1 either launch it under torchrun (one rank per GPU; RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment) or run it plainly: without WORLD_SIZE in the environment it re-executes
itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this host needs dmabuf IPC (exported by the image; kept if it is not)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# HBM traffic of one sa2_kernel launch at the bench's shape (960 objects): read at run time from the
# NEWEST committed PMC summary profiles/*pmc_sa*.txt (tools/pmc_sa.sh: FETCH_SIZE and WRITE_SIZE in
# separate --pmc passes, KiB units, FETCH_SIZE doubled per the gfx950 calibration in
# MI355X_MICROARCH.md §HBM).  None if no summary travels with the tree.


def sa2_traffic_from_profiles(kernel="sa2_kernel"):
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_sa*.txt")))     # rNN_vM_...: by name = by age
    for f in reversed(files):
        block, vals = None, {}
        for line in open(f):
            if not line.startswith((" ", "#")) and line.strip():
                block = line.strip()
            m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.eE+]+)", line)
            if m and block and block.startswith(kernel):
                vals[m.group(1)] = float(m.group(2))
        if len(vals) == 2:
            return (vals["FETCH_SIZE"] * 2 + vals["WRITE_SIZE"]) * 1024 / 960.0, os.path.relpath(f, ROOT)
    return None, None


MFMA_F32_PEAK_TF = 157.3       # f32-input MFMA dense peak
MFMA_BF16_PEAK_TF = 2500.0     # bf16 MFMA dense peak (MI355X_MICROARCH.md)
SPLIT_PRODUCTS = 6             # bf16 MFMA products per fp32-accurate product (csrc/sa_split.hip)
O, P = 60, 1024                # objects per scene, points per object (configs/msr3d.yaml:60,153)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="scenes per GPU per step (default 16; --full-step: 4) "
                    "(global 128 on 8 GPUs = configs/msr3d_3_dataset.yaml DDP shape)")
    ap.add_argument("--llm-hidden", type=int, default=4096, help="Vicuna-7B hidden size")
    ap.add_argument("--objects", type=int, default=60, help="objects per scene (120: BASELINE stress config)")
    ap.add_argument("--points", type=int, default=1024, help="points per object (2048: stress config)")
    ap.add_argument("--situation-type", default="as_transform_for_objects")
    ap.add_argument("--accum", type=int, default=1, help="gradient accumulation: micro-batches of --batch "
                    "scenes per optimiser step (the reference launches 4 scenes/GPU x 5: configs/msr3d.yaml:33,164); "
                    "a bench step is then one OPTIMISER step = accum micro-batches")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", action="store_true", help="(default since round 6) overlap the frozen encoder of batch "
                    "k+1 (side stream) with the trainable part of batch k; every step still encodes one batch and "
                    "trains one batch.  The roofline leg no longer times kernels inside the timed region: they are "
                    "event-timed in an eager, un-pipelined census pass after it (roofline.timed)")
    ap.add_argument("--no-pipeline", action="store_true", help="encoder and trainable part of a batch back to back on "
                    "one stream (rounds 1-5's default)")
    ap.add_argument("--census-steps", type=int, default=5, help="steps of the kernel census (0: no roofline leg)")
    ap.add_argument("--dense", action="store_true", help="variant line: the distinct-row kernels' WORST case -- every "
                    "ball query of both levels finds >= 32 different points, no padding slots (synth_batch(dense=True))")
    ap.add_argument("--from-store", action="store_true", help="also build every step's batch on the "
                    "device from HBM-resident scans (msr3d_amd.data: object selection, rotation, "
                    "subsample, normalise, padding) inside the timed region; default: batches "
                    "already resident, as the metric is defined")
    ap.add_argument("--host-inputs", action="store_true", help="secondary, labelled: the batches live in "
                    "pinned HOST memory and every step copies its batch over PCIe first (what the "
                    "reference's loader does, 1.47 MB/sample); the default keeps them resident in HBM")
    ap.add_argument("--skip-padded", action="store_true", help="padding-aware encoder: masked object "
                    "slots (the dataset pads scenes to 60 objects with one constant cloud) take the cached "
                    "feature of that cloud instead of being encoded again; identical outputs, less work. "
                    "Off by default: the headline number encodes all 60 slots like the reference")
    ap.add_argument("--time-all-kernels", action="store_true", help="HIP-event brackets around all four "
                    "encoder launches instead of the dominant one only (each pair of event records costs "
                    "the step ~10 us: 2.15 ms without any, 2.16 with one pair, 2.19 with four)")
    ap.add_argument("--unfrozen", action="store_true", help="variant line: `freeze: False` -- the PointNet++ "
                    "backbone trains too (BatchNorm in training mode, encoder forward + backward inside the "
                    "captured step); use a smaller --batch (4 scenes: 3.9 GiB of saved activations)")
    ap.add_argument("--window-step", action="store_true", help="(default with --accum A) the accumulation window as ONE "
                    "pass (HotPathTrainStep micro_batches=A): encoder and trainable part run once over the A x batch "
                    "scenes, the loss is taken per micro-batch slice, one optimiser step per pass")
    ap.add_argument("--micro-steps", action="store_true", help="with --accum A: A accumulated calls of --batch scenes "
                    "each (graph replay per micro-batch) instead of the window step")
    ap.add_argument("--no-window", action="store_true", help="with --accum: encode every micro-batch on its own "
                    "instead of the whole accumulation window in one encoder pass")
    ap.add_argument("--llm-stack", type=int, default=0, metavar="L", help="SECONDARY, labelled line: a training step of "
                    "the language-model side -- L LoRA-Llama decoder layers (Vicuna-7B shape) + final norm + 32000-way head + "
                    "per-sequence cross-entropy, forward + backward + LoRA gradient exchange (bucketed, from the backward "
                    "hooks) + clip + AdamW")
    ap.add_argument("--full-step", action="store_true", help="SECONDARY, labelled line: the FULL MSR3D training step -- hot path "
                    "(frozen encoder, prompter, llm_proj) -> scatter into inputs_embeds -> --llm-layers LoRA-Llama layers (Vicuna-7B "
                    "shapes, random bf16 weights) -> head -> per-sequence CE -> backward through the language model into the "
                    "prompter -> ONE flat gradient buffer (bucketed exchange from the backward hooks) -> clip + AdamW; "
                    "--batch sequences (default 4 here) x --seq-len tokens per GPU")
    ap.add_argument("--full-step-graph", action="store_true", help="--full-step: the whole step captured into ONE HIP graph "
                    "after an eager step and replayed (one rank only)")
    ap.add_argument("--llm-fp8", action="store_true", help="with --full-step / --llm-layer / --llm-stack: the decoder layers' frozen "
                    "projections on OCP e4m3 operands (per-output-channel weight scales, per-token activation scales, MX matrix "
                    "instruction; LoRA pair and accumulators bf16 / fp32) -- labelled in the line's dtype")
    ap.add_argument("--llm-layers", type=int, default=32, help="decoder layers of --full-step (32 = Vicuna-7B)")
    ap.add_argument("--seq-len", type=int, default=576, help="tokens per sequence of --full-step (prompt incl. 60 scene "
                    "tokens + answer; multiple of 64)")
    ap.add_argument("--round-tag", default=os.environ.get("MSR3D_ROUND_TAG", "r05"), help="prefix of files this run writes "
                    "under profiles/ (--cpu-ops)")
    ap.add_argument("--llm-layer", action="store_true", help="SECONDARY, labelled line: one LoRA-Llama decoder layer "
                    "(Vicuna-7B shape: hidden 4096, 32 heads, MLP 11008, LoRA r 16 on the seven projections), forward + "
                    "backward at 4 sequences x 576 tokens, bf16 -- SURVEY.md §8(f) rank 4; not the headline metric")
    ap.add_argument("--cpu-ops", action="store_true", help="per-op CPU micro-benchmarks at the GPU kernels' shapes "
                    "(BASELINE.md §3.4) beside the GPU kernels' times -> profiles/<round-tag>_cpu_ops.json; no training step")
    ap.add_argument("--no-extra", action="store_true", help="default run only: do not append the full-step figures "
                    "(extra.full_step: the FULL MSR3D training step at 4 x 576 tokens, e4m3 and bf16 projections, measured in "
                    "the same process after the headline when >= 60 GB of HBM are free)")
    ap.add_argument("--no-graph", action="store_true", help="issue the trainable part eagerly "
                    "instead of replaying the captured HIP graph")
    ap.add_argument("--cpu-threads", type=int, default=32, help="threads of the CPU baseline leg "
                    "(small GEMMs stop scaling well before a 2-socket host's 256 HW threads)")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 4 if args.full_step else 16
    args.pipeline = not args.no_pipeline
    return args


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: become `torchrun --nproc-per-node N bench.py ...` (same argv;
    rank 0's JSON line is the child's stdout = ours).  Returns the exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def build(args, device):
    import msr3d_amd.model  # noqa: F401
    import msr3d_amd.modules  # noqa: F401
    from msr3d_amd.config import AttrDict, default_prompter_cfg
    from msr3d_amd.model import build_model
    torch.manual_seed(1234)     # identical initial weights on every rank
    cfg = AttrDict({"prompter": default_prompter_cfg(situation_type=args.situation_type, freeze=not args.unfrozen),
                    "llm_hidden_size": args.llm_hidden, "model": {"name": "MSR3DHotPath"}})
    model = build_model(cfg).to(device)
    model.train()               # dropout active, as in training; frozen backbone stays eval
    if args.unfrozen and not any(p.requires_grad for p in model.visual_prompter.obj_encoder.parameters()):
        raise RuntimeError("--unfrozen: the encoder came out frozen")
    return model


class Trainer:
    """The hot-path step.  Optimiser settings: optim/build.py + configs/msr3d.yaml:43-47
    (AdamW lr 3e-5, betas (0.9, 0.999), wd 0.05), grad clip 5.0 (leo_trainer.py:192-193)."""

    def __init__(self, model, device, example_batch, E, use_graph, accum=1, micro=1):
        from msr3d_amd.dp import FlatGradAllReduce
        from msr3d_amd.train_step import HotPathTrainStep
        self.model = model
        from msr3d_amd import hipops
        params = [p for p in model.parameters() if p.requires_grad]
        on_gpu = device.type == "cuda"
        self.dp = FlatGradAllReduce(params, pack_groups=hipops.collect_pack_groups(model) if on_gpu else None)
        if on_gpu:
            from msr3d_amd.optim import FlatAdamW
            self.opt = FlatAdamW(self.dp, lr=3e-5, betas=(0.9, 0.999), weight_decay=0.05,
                                 max_grad_norm=5.0)
            hipops.attach_packed_views(model, self.dp, self.opt)
        else:
            self.opt = torch.optim.AdamW(params, lr=3e-5, betas=(0.9, 0.999), weight_decay=0.05)
        state = {}

        def loss_fn(out):
            # synthetic scalar loss on the projector output, L = mean(scene_embeds * w); its gradient
            # dL/dscene = w / n is handed to backward directly, the way the language model's
            # backward would deliver it (value and gradient identical to autograd's on `L`)
            y = out["scene_embeds"]
            if "w" not in state:                  # first (eager warm-up) call: the output shape is known
                g = torch.Generator(device="cpu").manual_seed(99)
                state["w"] = torch.randn(tuple(y.shape), generator=g).to(y.device)
                state["g"] = state["w"] / y.numel()
            if not y.is_cuda:
                return (y * state["w"]).sum() / y.numel()
            with torch.no_grad():
                loss = hipops.dot(y, state["g"])
            return loss, y, state["g"]

        self.stepper = HotPathTrainStep(model, self.opt, self.dp, loss_fn, example_batch,
                                        use_graph=use_graph, zero_in_optimizer=True, accum_steps=accum,
                                        micro_batches=micro)
        if on_gpu and use_graph:
            self.stepper.capture(example_batch)

    def step(self, batch, next_batch=None):
        return self.stepper(batch, next_batch)


def cpu_baseline(args, seconds):
    """The oracle (C, OpenMP over objects) + the torch-CPU mirror, same step, batch 1,
    bounded to ~`seconds` of work.  kind = "port": the reference's own ops have no CPU path."""
    from msr3d_amd.pointnet2 import pointnet2_utils
    from msr3d_amd.synth import synth_batch
    from oracle import pn2
    saved = pointnet2_utils._ext
    pointnet2_utils._ext = pn2.ext_module()
    try:
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = max(1, min(avail, args.cpu_threads))
        torch.set_num_threads(cores)
        pn2.set_threads(cores)
        model = build(args, torch.device("cpu"))
        batches = [synth_batch(10_000 + i, 1, O=O, P=P) for i in range(2)]
        tr = Trainer(model, torch.device("cpu"), batches[0], args.llm_hidden, use_graph=False)
        for _ in range(3):                        # warm-ups
            tr.step(batches[0])
        times = []
        t0 = time.perf_counter()
        while True:
            ts = time.perf_counter()
            tr.step(batches[len(times) % 2])
            times.append(time.perf_counter() - ts)
            el = time.perf_counter() - t0
            if (el >= seconds and len(times) >= 5) or len(times) >= 400:
                break
        n = len(times)
        st = sorted(times)
        q = lambda f: st[min(n - 1, int(f * n))]   # noqa: E731
        # one-thread figure on a short sample (a scalar port's number; bounded to a few seconds)
        torch.set_num_threads(1)
        pn2.set_threads(1)
        one = []
        t1 = time.perf_counter()
        while len(one) < 2 or (time.perf_counter() - t1 < min(4.0, seconds / 3) and len(one) < 20):
            ts = time.perf_counter()
            tr.step(batches[len(one) % 2])
            one.append(time.perf_counter() - ts)
        one.sort()
        host = "unknown"
        try:
            for ln in open("/proc/cpuinfo"):
                if ln.startswith("model name"):
                    host = ln.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        return {"value": 1.0 / q(0.5), "unit": "samples/s", "cores": cores, "kind": "port",
                "median_s_per_sample": q(0.5), "p10_s": q(0.10), "p90_s": q(0.90), "mean_value": n / el,
                "one_thread_value": 1.0 / one[len(one) // 2], "one_thread_steps": len(one),
                "thread_scaling_note": (f"{cores} of {avail} available hardware threads: {q(0.5) and (one[len(one) // 2] / q(0.5)):.2f}x the "
                                        "one-thread rate -- a batch-1 step is small GEMMs (60 x 256 tokens) and the serial FPS chain, "
                                        "which stop scaling after a few cores; a stated port, a baseline, never the target"),
                "host": {"cpu": host, "hw_threads": os.cpu_count(), "threads_available": avail},
                "sample": f"{n} steps of batch 1 ({O} obj x {P} pts) in {el:.1f}s after 3 warm-ups: C oracle "
                          "(OpenMP) for the 9 ops + torch-CPU mirror, fwd+bwd+AdamW; value = 1 / median step"}
    finally:
        pointnet2_utils._ext = saved


def llm_layer_line(args):
    """One decoder layer of the language model the scene tokens are fed to (msr3d_amd/llm/decoder.py)."""
    from msr3d_amd.llm import LoRALlamaDecoderLayer
    assert torch.cuda.is_available(), "bench.py --llm-layer needs a GPU"
    dev = torch.device("cuda", 0)
    Bq, T, Hd, NH, FF = 4, 576, 4096, 32, 11008        # configs/msr3d.yaml:164 batch 4; 576 = scene tokens + prompt + answer
    torch.manual_seed(0)
    layer = LoRALlamaDecoderLayer(Hd, NH, FF, r=16, lora_alpha=16, device=dev, base="fp8" if args.llm_fp8 else "bf16")
    with torch.no_grad():
        for grp in (layer.self_attn, layer.mlp):
            for m in grp.values():
                m.load_base_weight(torch.randn(m.out_features, m.in_features, device=dev) / m.in_features ** 0.5)
                m.lora_B.weight.normal_(std=0.02)
    x = torch.randn(Bq, T, Hd, device=dev).bfloat16().requires_grad_(True)
    keep = torch.ones(Bq, T, dtype=torch.uint8, device=dev)
    keep[1, :40] = 0
    gy = (torch.randn(Bq, T, Hd, device=dev) * 0.01).bfloat16()

    def step():
        for p in layer.parameters():
            p.grad = None
        x.grad = None
        layer(x, attention_mask=keep).backward(gy)
    for _ in range(max(args.warmup, 2)):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = e0.elapsed_time(e1) / args.steps
    M = Bq * T
    lin = 2.0 * M * (4 * Hd * Hd + 3 * Hd * FF)                       # the seven projections, forward
    att = 2.0 * 2.0 * Bq * NH * T * T * (Hd // NH)                    # Q K^T and P V, forward
    flop = 2.0 * lin + 3.0 * att                                      # backward: dx only (frozen weights) + 4 attention products
    print(json.dumps({
        "metric": "SECONDARY: LoRA-Llama decoder layer fwd+bwd (Vicuna-7B shape), tokens/s per layer",
        "value": M / (ms * 1e-3), "unit": "tokens/s", "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 2),
        "ms_per_step": ms, "wall_ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True,
        "dtype": "bf16; frozen projections on OCP e4m3 operands" if args.llm_fp8 else "bf16",
        "data": "synthetic (random bf16 weights; no checkpoint on the box)",
        "config": {"workload": "one decoder layer: hidden 4096, 32 heads, MLP 11008, LoRA r=16 alpha=16 on q/k/v/o/gate/up/down, "
                               "4 sequences x 576 tokens, left-padded mask, eager (host-issued) launches"},
        "tflops": flop / (ms * 1e-3) / 1e12, "frac_of_bf16_peak": flop / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF,
        "note": "not the headline metric; the frozen LLM is out of §8(a)-(e) scope (SURVEY §0.5) -- this line prices "
                "the §8(f) rank-4 building block"}))


def full_step_line(args, emit=True):
    """SECONDARY line: the FULL MSR3D training step (msr3d_amd/model/msr3d_full.py + msr3d_amd/full_step.py) at the
    Vicuna-7B shapes of BASELINE configs[1] -- random bf16 weights (no checkpoint on the box), synthetic scenes and
    token ids, everything resident in HBM."""
    import msr3d_amd.model  # noqa: F401
    import msr3d_amd.modules  # noqa: F401
    from msr3d_amd.config import AttrDict, default_prompter_cfg
    from msr3d_amd.full_step import FullTrainStep
    from msr3d_amd.model import build_model
    from msr3d_amd.synth import synth_batch, synth_text
    assert torch.cuda.is_available(), "bench.py --full-step needs a GPU"
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("MSR3D_BENCH_SINGLE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist_on = world > 1 or os.environ.get("MSR3D_BENCH_FORCE_DIST") == "1"
    if dist_on and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ["MSR3D_DP_FORCE_EXCHANGE"] = "1"
    if dist_on and not dist.is_initialized():
        backend = os.environ.get("MSR3D_BENCH_BACKEND", "nccl")
        dist.init_process_group("nccl", device_id=dev) if backend == "nccl" else dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    L, Bq, T, Hd, NH, FF, V = args.llm_layers, args.batch, args.seq_len, args.llm_hidden, 32, 11008, 32000
    if Hd != 4096:
        NH, FF = 40, 13824                                   # Vicuna-13B
    T_out = 32
    torch.manual_seed(1234)
    cfg = AttrDict({"prompter": default_prompter_cfg(situation_type=args.situation_type), "llm_hidden_size": Hd,
                    "llm": {"num_layers": L, "hidden_size": Hd, "num_heads": NH, "intermediate_size": FF, "vocab_size": V,
                            "lora": {"rank": 16, "alpha": 16}, "base": "fp8" if args.llm_fp8 else "bf16"},
                    "device": str(dev), "model": {"name": "MSR3DFullStep"}})
    model = build_model(cfg).to(dev).train()
    net = model.llm_model
    with torch.no_grad():
        for layer in net.layers:
            for grp in (layer.self_attn, layer.mlp):
                for m in grp.values():
                    m.load_base_weight(torch.randn(m.out_features, m.in_features, device=dev) / m.in_features ** 0.5)
                    m.lora_B.weight.normal_(std=0.02)
        net.lm_head.load_weight(torch.randn(V, Hd, device=dev) / Hd ** 0.5)
        model.embed_tokens.copy_(torch.randn(V, Hd, device=dev) * 0.02)
    ts = FullTrainStep(model, lr=3e-5, betas=(0.9, 0.999), weight_decay=0.05, max_grad_norm=5.0,
                       use_graph=args.full_step_graph and not dist_on)
    Ltok = O + (1 if args.situation_type == "as_object" else 0)
    batches = []
    for i in range(3):
        b = synth_batch(1000 * rank + i, Bq, O=O, P=P, device=dev)
        b.update(synth_text(5000 + 1000 * rank + i, Bq, L=Ltok, T_in=T - T_out, T_out=T_out, vocab=V, device=dev))
        batches.append(b)
    for i in range(max(args.warmup, 2)):
        ts(batches[i % 3])
    ranks_seen = 1
    if dist_on:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        ts.dp.timing = True
        ts.dp.comm_events.clear()
        ts.dp.wait_events.clear()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = ts(batches[i % 3])
        marks[i + 1].record()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    comm = None
    if dist_on:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ts.dp.timing = False
        ar = [a.elapsed_time(b) for a, b in ts.dp.comm_events]
        wt = [a.elapsed_time(b) for a, b in ts.dp.wait_events]
        _, spread = ts.dp.replica_checksum(ts.opt.flat_p)
        stats = torch.tensor([sum(ar) / args.steps, sum(wt) / args.steps], device=dev, dtype=torch.float64)
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        comm = {"ranks_seen": ranks_seen, "exchange": ts.dp.exchange_mode, "buckets": len(ts.dp.buckets),
                "collectives_per_step": len(ar) / args.steps, "exchange_ms_per_step_on_comm_stream": float(stats[0]),
                "exchange_exposed_ms_per_step": float(stats[1]), "replica_checksum_spread": spread}
        assert ranks_seen == world and spread == 0.0
    elapsed = float(tt.item())
    if rank == 0:
        per = sorted(a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:]))
        M = Bq * T
        lin = 2.0 * M * (4 * Hd * Hd + 3 * Hd * FF)
        att = 2.0 * 2.0 * Bq * NH * T * T * (Hd // NH)
        flop = L * (2.0 * lin + 3.0 * att) + 2.0 * 2.0 * M * Hd * V + Bq * 8.8e9
        ms = 1e3 * elapsed / args.steps
        line = {
            "metric": "SECONDARY: full MSR3D training step (hot path + LoRA-Llama), samples/s",
            "value": Bq * world * args.steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": max(args.warmup, 2), "ms_per_step": ms,
            "ms_per_step_percentiles": {"p10": per[int(0.1 * len(per))], "p50": per[len(per) // 2], "p90": per[min(len(per) - 1, int(0.9 * len(per)))]},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "hot path f32 (bf16x3 split MFMA); language model bf16 storage, fp32 accumulate; LoRA / optimiser fp32" +
                     ("; frozen projections of the decoder layers: OCP e4m3 operands (forward and dx), per-channel / per-token scales"
                      if args.llm_fp8 else ""),
            "data": "synthetic (random bf16 LLM weights: no checkpoint on the box; synthetic scenes and token ids)",
            "config": {"workload": "configs/msr3d.yaml full step: frozen PointNet++ -> OSE3DSituation -> llm_proj -> scatter into "
                                   f"inputs_embeds -> {L} LoRA-Llama layers (hidden {Hd}, {NH} heads, MLP {FF}, LoRA r=16 on "
                                   "q/k/v/o/gate/up/down) -> RMSNorm -> 32000-way frozen head -> per-sequence CE -> backward through "
                                   "the LLM and the scatter into the prompter -> ONE flat gradient buffer, buckets exchanged from the "
                                   "backward hooks -> clip + AdamW; " + ("ONE captured HIP graph per step" if ts.graph is not None
                                                                         else "eager launches"),
                       "layers": L, "sequences_per_gpu": Bq, "tokens_per_sequence": T, "scene_tokens": Ltok,
                       "objects": O, "points": P, "trainable_parameters": ts.dp.numel, "grad_bytes": ts.dp.numel * 4,
                       "lora_parameters": sum(p.numel() for p in net.lora_parameters()),
                       "unused_parameters_skipped": len(ts.unused_parameters),
                       "prompter_schedule": "blocks" if getattr(model._schedule, "_ran_blocks", False) else "strips/modular",
                       "attention": "fused (msr3d_attn_fwd / _bwd): all tokens of every layer",
                       "head_and_loss_rows": "answer span only: final norm, head, cross-entropy and the LAST layer's MLP over the "
                                             f"{T_out + 1} positions whose logits the loss reads (targets are -100 over the "
                                             "prompt by construction, msr3d.py:384-392); same loss, same gradients",
                       "parallelism": f"dp{world}"},
            "loss": float(loss), "tokens_per_s": world * M * args.steps / elapsed,
            "tflops_per_gpu": flop / (ms * 1e-3) / 1e12, "frac_of_bf16_peak": flop / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF,
            "hbm_allocated_gb": torch.cuda.max_memory_allocated(dev) / 1e9,
            "note": "not the headline metric (SURVEY 0.5 / 8(d): the hot-path line is); this is the step BASELINE configs[1] names"}
        if comm is not None:
            line["comm"] = comm
    if not emit:                 # (bench.py's default run attaches this line's numbers to the headline as extra.full_step)
        del ts, model, net, batches
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        return line
    if dist_on:
        dist.destroy_process_group()
        import ctypes
        ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(line), flush=True)


def full_step_extra(args, tr, model, batches, device):
    """-> {"fp8": {...}, "bf16": {...}} | {"skipped": reason}: bench.py --full-step [--llm-fp8] at 4 sequences x 576 tokens,
    8 timed steps each, run inside the default invocation once the headline has been measured and its state released."""
    import copy
    import gc
    del tr, model, batches
    gc.collect()
    torch.cuda.empty_cache()
    free = torch.cuda.mem_get_info(device)[0]
    if free < 60e9:
        return {"skipped": f"{free / 1e9:.0f} GB of HBM free, 60 needed"}
    out = {}
    for name, fp8 in (("fp8", True), ("bf16", False)):
        a = copy.copy(args)
        a.full_step, a.llm_fp8, a.batch, a.steps, a.warmup, a.full_step_graph = True, fp8, 4, 8, 2, False
        a.llm_layers, a.seq_len = 32, 576
        try:
            ln = full_step_line(a, emit=False)
        except Exception as e:                 # noqa: BLE001 -- the headline line must still be printed
            out[name] = {"error": repr(e)[:300]}
            continue
        out[name] = {k: ln[k] for k in ("value", "unit", "ms_per_step", "ms_per_step_percentiles", "tokens_per_s", "tflops_per_gpu",
                                        "frac_of_bf16_peak", "hbm_allocated_gb", "dtype", "loss")}
        out[name]["config"] = {k: ln["config"][k] for k in ("layers", "sequences_per_gpu", "tokens_per_sequence", "scene_tokens",
                                                            "trainable_parameters", "prompter_schedule")}
        out[name]["steps"], out[name]["warmup"] = a.steps, a.warmup
    out["note"] = ("SECONDARY figures (python bench.py --full-step [--llm-fp8]): random bf16 LLM weights, synthetic scenes and "
                   "token ids; not the headline metric")
    return out


def llm_stack_line(args):
    """The language-model side of a step: msr3d_amd/llm/stack.py on the flat-gradient engine (dp.py, optim.py)."""
    import torch.distributed as dist
    from msr3d_amd.dp import FlatGradAllReduce
    from msr3d_amd.llm import LoRALlamaStack
    from msr3d_amd.optim import FlatAdamW
    assert torch.cuda.is_available(), "bench.py --llm-stack needs a GPU"
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=dev)
    L, Bq, T, Hd, NH, FF, V = args.llm_stack, 4, 576, 4096, 32, 11008, 32000
    torch.manual_seed(0)
    net = LoRALlamaStack(L, Hd, NH, FF, V, r=16, lora_alpha=16, device=dev, base="fp8" if args.llm_fp8 else "bf16")
    with torch.no_grad():
        for layer in net.layers:
            for grp in (layer.self_attn, layer.mlp):
                for m in grp.values():
                    m.load_base_weight(torch.randn(m.out_features, m.in_features, device=dev) / m.in_features ** 0.5)
                    m.lora_B.weight.normal_(std=0.02)
        net.lm_head.load_weight(torch.randn(V, Hd, device=dev) / Hd ** 0.5)
    dp = FlatGradAllReduce(net.lora_parameters(), bucket_bytes=1 << 20, overlap=True)   # a bucket ~ one layer's LoRA pair set
    opt = FlatAdamW(dp, lr=3e-5, betas=(0.9, 0.999), weight_decay=0.05, max_grad_norm=5.0)
    g = torch.Generator(device="cpu").manual_seed(1 + rank)
    x = (torch.randn(Bq, T, Hd, generator=g) * 0.5).to(dev).bfloat16()
    keep = torch.ones(Bq, T, dtype=torch.uint8, device=dev)
    keep[1, :40] = 0
    targets = torch.randint(0, V, (Bq, T), generator=g).to(dev)
    targets[:, :200] = -100                                     # scene tokens + prompt are not supervised

    def step():
        dp.zero_grad()
        net(x, attention_mask=keep, targets=targets).mean().backward()     # buckets leave from the hooks (world > 1)
        dp.finish()
        opt.step()
    for _ in range(max(args.warmup, 2)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    ms = e0.elapsed_time(e1) / args.steps
    M = Bq * T
    lin = 2.0 * M * (4 * Hd * Hd + 3 * Hd * FF)
    att = 2.0 * 2.0 * Bq * NH * T * T * (Hd // NH)
    flop = L * (2.0 * lin + 3.0 * att) + 2.0 * 2.0 * M * Hd * V          # + the head, forward and dx
    if rank == 0:
        print(json.dumps({
            "metric": "SECONDARY: LoRA-Llama training step, language-model side (Vicuna-7B shapes), tokens/s",
            "value": world * M * args.steps / wall, "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": max(args.warmup, 2), "ms_per_step": 1e3 * wall / args.steps, "gpu_ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "dtype": "bf16",
            "data": "synthetic (random bf16 weights; no checkpoint on the box)",
            "config": {"workload": f"{L} decoder layers (hidden 4096, 32 heads, MLP 11008, LoRA r=16 on q/k/v/o/gate/up/down) + "
                                   "RMSNorm + 32000-way frozen head + per-sequence mean cross-entropy; 4 sequences x 576 "
                                   "tokens per GPU, left-padded mask; forward + backward + bucketed LoRA-gradient exchange "
                                   "from the backward hooks + clip + AdamW; eager launches",
                       "layers": L, "lora_parameters": dp.numel, "grad_bytes": dp.numel * 4},
            "tflops_per_gpu": flop / (ms * 1e-3) / 1e12, "frac_of_bf16_peak": flop / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF,
            "note": "not the headline metric; the frozen LLM is out of §8(a)-(e) scope (SURVEY §0.5) -- this line prices "
                    "the §8(f) rank-4 step assembled from the C-ABI pieces"}))
    if world > 1:
        dist.destroy_process_group()


def kernel_census(tr, batches, steps, run_window, first):
    """-> {entry key: {"us": mean launch duration, "per_step": launches per step}}"""
    from msr3d_amd import _lib
    st = tr.stepper
    graph, st.graph = st.graph, None                 # eager issue: __call__ takes _train_part() / _micro_step()
    saved_step = tr.step
    plain = lambda b, nb=None: saved_step(b, None)   # noqa: E731 -- un-pipelined: no next batch on the side stream
    tr.step = plain
    sink = {}
    try:
        run_window(first)                            # one untimed eager step (lazy allocations of the eager path)
        torch.cuda.synchronize()
        _lib.set_timing_sink(sink, census=True)
        for i in range(steps):
            run_window(first + 1 + i)
        torch.cuda.synchronize()
    finally:
        _lib.set_timing_sink(None)
        st.graph, tr.step = graph, saved_step
    return {k: {"us": 1e3 * sum(a.elapsed_time(b) for a, b in v) / len(v), "per_step": len(v) / steps}
            for k, v in sink.items() if v}


# entry key -> the HIP kernel it launches at the bench shape (names as rocprofv3 prints them, for the PMC summaries)
CENSUS_KERNELS = {
    "msr3d_sa_fps2_query_flags": "fps_query_kernel", "msr3d_sa_fps2_query_plan": "fps_query_plan_kernel",
    "msr3d_sa_plan12": "sa12_plan_kernel",
    "msr3d_sa_level1_rows": "sa1_rows_kernel", "msr3d_sa_level2_rows": "sa2_rows_kernel",
    "msr3d_sa_level_split[1]": "sa1_split_kernel", "msr3d_sa_level_split[2]": "sa2_split_kernel",
    "msr3d_sa_level_split[3]": "sa3_split4_kernel", "msr3d_sa_level3_tiles": "sa3_tiles_kernel", "msr3d_rows_linear_split": "rows_linear_kernel",
    "msr3d_split_pack_begin": "split_pack_kernel", "msr3d_gemm_multi_f32": "panel_multi_kernel",
    "msr3d_pos_embed_fwd": "pos_embed_fwd_kernel", "msr3d_pos_embed_bwd": "pos_embed_bwd_kernel",
    "msr3d_scene_block[attn_fwd]": "scene_attn_fwd2_kernel", "msr3d_scene_block[attn_bwd]": "scene_attn_bwd3_kernel",
    "msr3d_scene_block[ffn_fwd]": "scene_block_kernel<1, 8>", "msr3d_scene_block[ffn_bwd]": "scene_block_kernel<2, 8>",
    "msr3d_scene_block[linear]": "scene_block_kernel<4, 4>", "msr3d_scene_block[linear_ksplit]": "scene_block_kernel<5, 4>",
    "msr3d_wgrad_split_mixed": "wgrad_mixed_kernel", "msr3d_wgrad_split_colsum": "wgrad_split_kernel",
    "msr3d_wgrad_split": "wgrad_split_kernel", "msr3d_wgrad_stream": "wgrad_stream_kernel (+ wgrad_fixup_kernel)",
    "msr3d_adamw_flat_scaled": "adamw_kernel (+ sumsq_kernel)", "msr3d_dot_f32": "dot_kernel",
    "msr3d_scene_prologue": "scene_prologue_kernel",
}


def pmc_counters_from_profiles():
    """{kernel name: {counter: value}} from the NEWEST committed PMC summaries (profiles/*pmc_blocks*.txt, *pmc_sa*.txt:
    tools/pmc_blocks.sh / pmc_sa.sh, counters in their own rocprofv3 --pmc passes) + the files they came from."""
    import glob
    import re
    out, used = {}, []
    for pat in ("*pmc_sa*.txt", "*pmc_blocks*.txt"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
        if not files:
            continue
        used.append(os.path.relpath(files[-1], ROOT))
        block = None
        for line in open(files[-1]):
            if not line.startswith((" ", "#")) and line.strip():
                block = line.strip()
                out.setdefault(block, {})
            m = re.match(r"\s+([A-Za-z0-9_]+)\s+([0-9.eE+]+)", line)
            if m and block:
                out[block][m.group(1)] = float(m.group(2))
    return out, used


def census_table(census, model, world_rows, E, level_flops):
    """One row per entry point: duration, launches per step, the FLOPs the RESULT needs (fp32 products counted once,
    whatever the number of bf16 MFMA products each takes), their rate against the fp32-accurate roof (2500 / 6 TFLOP/s),
    the matrix-pipe busy fraction of the newest committed PMC pass."""
    sched = getattr(model, "_schedule", None)
    dm = getattr(sched, "dims", None) or {}
    Bs, L, M, D, W, H, FF = (dm.get(k, 0) for k in ("B", "L", "M", "D", "W", "H", "FF"))
    KE = dm.get("KE", 768)
    flop = dict(level_flops)
    if M:
        core = 2.0 * Bs * H * L * L * (D // max(H, 1))                       # one (L x L x 32) product per head and scene
        flop.update({
            "msr3d_scene_block[attn_fwd]": 2.0 * M * D * W + 2 * core + 2.0 * M * D * D,        # qkvc | QK^T, PV | out-proj
            "msr3d_scene_block[attn_bwd]": 2.0 * M * D * D + 4 * core + 2.0 * M * W * D,        # d ctx | dP dV dQ dK | d x
            "msr3d_scene_block[ffn_fwd]": 4.0 * M * D * FF, "msr3d_scene_block[ffn_bwd]": 4.0 * M * D * FF,
            "msr3d_scene_block[linear]": 2.0 * M * D * E, "msr3d_scene_block[linear_ksplit]": 2.0 * M * D * E,
            "msr3d_gemm_multi_f32": 2.0 * M * KE * D,
        })
        wg = getattr(sched, "wgrad", None)
        if wg is not None:
            f = sum(2.0 * p.M * p.n_out * p.k_in for p in wg.probs)
            for k in ("msr3d_wgrad_split_mixed", "msr3d_wgrad_split_colsum", "msr3d_wgrad_split", "msr3d_wgrad_stream"):
                flop[k] = f
    flop["msr3d_rows_linear_split"] = 2.0 * world_rows * 768 * 768
    pmc, pmc_files = pmc_counters_from_profiles()
    peak = MFMA_BF16_PEAK_TF / SPLIT_PRODUCTS
    rows = {}
    for k, v in census.items():
        kern = CENSUS_KERNELS.get(k, k)
        r = {"kernel": kern, "us": round(v["us"], 2), "launches_per_step": v["per_step"]}
        if k in flop and v["us"] > 0:
            tf = flop[k] / (v["us"] * 1e-6) / 1e12
            r.update({"useful_gflop": round(flop[k] / 1e9, 3), "achieved_tflops": round(tf, 2), "frac": round(tf / peak, 4)})
        for name, c in pmc.items():
            if kern.split(" ")[0] in name and "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE", 0) > 0:
                # busy cycles summed over the chip's 1024 SIMDs / (1024 x the launch's cycles; GRBM_GUI_ACTIVE sums 8 XCDs)
                r["mfma_busy_pmc"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0), 4)
                if "FETCH_SIZE" in c and "WRITE_SIZE" in c:      # KiB; FETCH_SIZE x 2: the gfx950 calibration of the guide
                    r["hbm_bytes_pmc"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
                break
        rows[k] = r
    return rows, pmc_files


def measure_variant(args, device, steps=30, warmup=5, **over):
    """The default step on other inputs (bench.py's extra.* lines): fresh model, trainer and batches -> ms per step."""
    import copy
    from msr3d_amd.synth import synth_batch
    a = copy.copy(args)
    for k, v in over.items():
        setattr(a, k, v)
    from msr3d_amd import scene_blocks
    prev_mma = scene_blocks.set_train_mma(getattr(a, "train_mma", None)) if getattr(a, "train_mma", None) else None
    if getattr(a, "train_mma", None):
        from msr3d_amd import _lib as _lib_mod
        _lib_mod.load_bf16()
    model = build(a, device)
    # (the headline's resident batches: the encoder's time depends on the scenes -- distinct rows, padding slots)
    batches = [synth_batch(i, a.batch, O=O, P=P, device=device, dense=a.dense) for i in range(4)]
    tr = Trainer(model, device, batches[0], a.llm_hidden, use_graph=True)
    nxt = (lambda i: batches[(i + 1) % 4]) if a.pipeline else (lambda i: None)
    for i in range(warmup):
        tr.step(batches[i % 4], nxt(i))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(batches[(warmup + i) % 4], nxt(warmup + i))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    from msr3d_amd.pointnet2 import fused as _fused
    net = model.visual_prompter.obj_encoder.pcd_net
    st = _fused.row_statistics(net, batches[0]["obj_fts"].reshape(-1, P, 6))
    res = {"value": a.batch * steps / el, "unit": "samples/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "warmup": warmup,
           "distinct_rows_per_launch": {f"level{l}": st[l]["distinct_rows"] for l in (1, 2)},
           "nominal_rows_per_launch": {f"level{l}": st[l]["nominal_rows"] for l in (1, 2)},
           "constant_objects": st["constant_objects"],
           "schedule": "blocks" if getattr(getattr(model, "_schedule", None), "_ran_blocks", False) else "strips/modular"}
    if getattr(a, "train_mma", None):
        # how far the reduced variant's outputs are from the fp32-accurate path: same weights, same batch, dropout off
        res["rel_l2_vs_f32"] = train_mma_distance(model, batches[0])
        scene_blocks.set_train_mma(prev_mma)
    del tr, model, batches
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return res


def train_mma_distance(model, batch):
    """rel-L2 of obj_tokens / scene_embeds of the CURRENT trainable-part arithmetic against the fp32-accurate blocks."""
    from msr3d_amd import scene_blocks
    outs = {}
    cur = scene_blocks.train_mma()
    drops = [(m, m.p) for m in model.modules() if isinstance(m, torch.nn.Dropout)]
    try:
        for m, _ in drops:          # (the schedule reads the modules' p: masks are keyed by a per-forward salt, so the two
            m.p = 0.0               #  passes could not draw the same ones)
        for mode in (cur, "f32"):
            scene_blocks.set_train_mma(mode)
            out = model(dict(batch))
            outs[mode] = (out["obj_tokens"].detach().double().clone(), out["scene_embeds"].detach().double().clone())
    finally:
        scene_blocks.set_train_mma(cur)
        for m, p0 in drops:
            m.p = p0
    a, b = outs[cur], outs["f32"]
    return {"obj_tokens": float((a[0] - b[0]).norm() / b[0].norm()), "scene_embeds": float((a[1] - b[1]).norm() / b[1].norm())}


def main():
    global O, P
    args = parse()
    O, P = args.objects, args.points
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not (args.llm_layer or args.cpu_ops):
        sys.exit(relaunch_under_torchrun(args))
    if args.full_step:
        return full_step_line(args)
    if args.llm_stack:
        return llm_stack_line(args)
    if args.llm_layer:
        return llm_layer_line(args)
    if args.cpu_ops:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        import cpu_ops_bench          # (times the oracle as the CPU baseline: lives outside the product package)
        res = cpu_ops_bench.run(threads=args.cpu_threads)
        out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"{args.round_tag}_cpu_ops.json")
        with open(out, "w") as f:
            json.dump(res, f, indent=1)
        for k, v in res["rows"].items():
            print(f"{k:56s} " + "  ".join(f"{n} {x:9.3f}" for n, x in v.items()))
        print(json.dumps({"cpu_ops": out, "threads": res["threads"], "host": res["host"]}))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    if world > 1:
        # The 21 MB gradient all-reduce runs beside the next batch's encoder and has ~0.6 ms to finish:
        # eight channels (~20 GB/s each over xGMI) are plenty, and the encoder's persistent kernels then
        # leave exactly that many CUs free (msr3d_amd/dp.py -> msr3d_set_reserved_cus).  Overridable.
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "8")
    # test hooks (tests/test_bench_ranks_gpu.py): run several ranks on ONE GPU over gloo to
    # exercise the multi-rank code path where only a single device exists
    if os.environ.get("MSR3D_BENCH_SINGLE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # second test hook: MSR3D_BENCH_FORCE_DIST=1 with ONE rank initialises RCCL anyway and takes the
    # multi-rank schedule (split graph, all-reduce on the communication stream hidden behind the
    # next batch's encoder) over a one-rank communicator -- the real backend on a one-GPU box
    dist_on = world > 1 or os.environ.get("MSR3D_BENCH_FORCE_DIST") == "1"
    if dist_on and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ["MSR3D_DP_FORCE_EXCHANGE"] = "1"
    if dist_on:
        backend = os.environ.get("MSR3D_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from msr3d_amd import _lib
    from msr3d_amd.synth import synth_batch
    if os.environ.get("MSR3D_BENCH_RESERVE_CUS"):       # experiment knob: CUs the persistent encoder kernels leave free
        _lib.set_reserved_cus(int(os.environ["MSR3D_BENCH_RESERVE_CUS"]))
    # (the attention forward block's form is the library's default in every schedule -- two workgroups per (scene, head):
    #  with a scene's workgroups on one XCD it is the faster one pipelined as well, 0.846 against 0.855 ms;
    #  MSR3D_ATTN_FWD_SPLIT=0 selects one workgroup)
    model = build(args, device)
    if args.skip_padded:
        model.visual_prompter.obj_encoder.skip_padded = True
    B = args.batch
    # distinct resident batches per rank, cycled (weak scaling: per-GPU work fixed)
    # --window-step: a call is a whole accumulation window (accum x B scenes back to back), one call per optimiser step
    wstep = not args.micro_steps and args.accum > 1 and not args.unfrozen and not args.host_inputs and not args.from_store
    calls = 1 if wstep else args.accum                    # step calls per optimiser step
    Bcall = B * args.accum if wstep else B                # scenes per call
    n_resident = 4 if calls == 1 else max(4, 2 * calls)     # (a window's micro-batches are distinct)
    batches = [synth_batch(1000 * rank + i, Bcall, O=O, P=P, device=device, dense=args.dense) for i in range(n_resident)]
    tr = Trainer(model, device, batches[0], args.llm_hidden, use_graph=not args.no_graph, accum=calls,
                 micro=args.accum if wstep else 1)

    # Software pipelining (msr3d_amd/train_step.py): the frozen encoder of batch k+1 runs on a
    # side stream while batch k trains.  Every timed step still encodes exactly one batch and
    # trains exactly one batch; the first timed batch's features come from the last warm-up step,
    # the last timed step encodes the batch that would follow.
    # With several ranks the next batch is ALWAYS handed to the step: its frozen encoder is then
    # issued on the compute stream between the start of the gradient all-reduce and the optimiser,
    # which hides the exchange (msr3d_amd/train_step.py) -- no side stream, kernels do not share CUs.
    pipe = args.pipeline or dist_on

    def nxt(i):
        return batches[(i + 1) % n_resident] if pipe else None

    if args.from_store:
        # the reference builds samples on the host (num_workers: 0) and
        # ships 1.47 MB/sample over PCIe; here 8 synthetic scans (150 k points, 60-95 instances)
        # live in HBM and each step's batch is one msr3d_preprocess_pcd launch.
        import random
        from msr3d_amd.data import SceneInputBuilder, SceneStore
        from msr3d_amd.synth import synth_scan
        rng = np.random.default_rng(77 + rank)
        store = SceneStore(device)
        for sidx in range(8):
            store.add_scan(f"scan{sidx}", *synth_scan(rng, 60 + 5 * sidx, 150_000))
        builder = SceneInputBuilder(store, max_obj_len=O, num_points=P, split="train", seed=rank)
        random.seed(rank)

        def yaw():                                   # facing direction in the xy plane, as MSQA's
            a = rng.uniform(0, np.pi)
            return np.array([0.0, 0.0, np.sin(a), np.cos(a)])

        descs = [[{"scan_id": f"scan{int(rng.integers(8))}", "insts": [int(x) for x in rng.integers(0, 60, 4)],
                   "situation": (rng.uniform(-3, 3, 3), yaw())} for _ in range(B)] for _ in range(n_resident)]
        built = [None]
        plain_step = tr.step

        def step_from_store(_batch, _next=None, _i=[0]):
            built[0] = builder.build(descs[_i[0] % n_resident], out=built[0])
            _i[0] += 1
            return plain_step(built[0], None)
        tr.step = step_from_store

    if args.host_inputs:
        # PCIe-inclusive variant: pinned host batches; batch i+1 crosses PCIe on a copy
        # stream into the second of two device buffers while step i computes
        host = [{k: v.cpu().pin_memory() for k, v in b.items()} for b in batches]
        bufs = [{k: torch.empty_like(v) for k, v in batches[0].items()} for _ in range(2)]
        copy_stream = torch.cuda.Stream()
        ready = [None, None]
        staged_step = tr.step

        def upload(i):       # batch i -> bufs[i % 2] on the copy stream, once step i-2 no longer reads it
            copy_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(copy_stream):
                for k, v in host[i % n_resident].items():
                    bufs[i % 2][k].copy_(v, non_blocking=True)
                ready[i % 2] = torch.cuda.Event()
                ready[i % 2].record(copy_stream)

        def step_from_host(_batch, _next=None, _i=[0]):
            i = _i[0]
            _i[0] += 1
            if ready[i % 2] is None:         # very first call
                upload(i)
            upload(i + 1)                    # waits for step i-1 (the last reader of that buffer), then
            torch.cuda.current_stream().wait_event(ready[i % 2])      # runs while step i computes
            return staged_step(bufs[i % 2], None)
        tr.step = step_from_host

    # gradient accumulation over a frozen encoder: the whole window's objects go through ONE encoder pass
    window = calls > 1 and not args.unfrozen and not args.no_window and not args.host_inputs

    def window_of(k):
        return [batches[(k * calls + m) % n_resident] for m in range(calls)]

    def run_window(k):
        """optimiser step k = calls micro-steps"""
        if window and not tr.stepper.window_ready(window_of(k)):
            tr.stepper.encode_window(window_of(k))
        for m in range(calls):
            j = k * calls + m
            nb = nxt(j)
            if window:       # N > 1: the NEXT window's encoder pass is what runs beside the last micro-step's exchange
                nb = window_of(k + 1) if (dist_on and pipe and m == calls - 1) else None
            tr.step(batches[j % n_resident], nb)

    for i in range(args.warmup):
        run_window(i)

    # N > 1 self-checks (nobody can watch an 8-GPU run: the line has to prove it was one).
    #   ranks_seen: an all-reduce of ones over the communicator the gradients travel on
    ranks_seen = 1
    if dist_on:
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        tr.dp.timing = True
        tr.dp.comm_events.clear()
        tr.dp.wait_events.clear()

    # The timed region carries NO per-kernel instrumentation (round 6): an event pair idles the GPU for ~12 us around a
    # launch and, with the next batch's encoder on the side stream, would time kernels that share the chip.  The
    # roofline leg's durations come from the kernel census below, after the timed region.
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    # (no per-step event records in the timed region either: one record a step cost 3-5 us of the step, 0.8421 -> 0.8396 /
    #  0.8416 -> 0.8359 ms in two interleaved pairs; the per-step spread is taken in a short pass of its own below)
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_window(args.warmup + i)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist_on:
        tr.dp.timing = False          # (the exchange statistics below are the timed region's)
    # per-step spread: the same steps again with an event record between them (up to 30; not part of `value`)
    n_spread = min(args.steps, 30)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_spread + 1)]
    marks[0].record()
    for i in range(n_spread):
        run_window(args.warmup + args.steps + i)
        marks[i + 1].record()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()


    # MSR3D_DP_GRAPH_COMM=1: the RCCL call was captured with the rest, the step is one graph at N > 1 as well
    whole_graph = tr.stepper.graph is not None and not tr.stepper.split
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if dist_on:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    comm = None
    if dist_on:
        tr.dp.timing = False
        ar = [a.elapsed_time(b) for a, b in tr.dp.comm_events]          # on the communication stream
        wt = [a.elapsed_time(b) for a, b in tr.dp.wait_events]          # compute stream waiting for it
        # replicas must hold the same weights after the same number of steps
        _, spread = tr.dp.replica_checksum(getattr(tr.opt, "flat_p", None))
        stats = torch.tensor([sum(ar) / max(len(ar), 1), sum(wt) / max(len(wt), 1)], device=device,
                             dtype=torch.float64)
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        comm = {"ranks_seen": ranks_seen, "exchange": tr.dp.exchange_mode,
                "rccl_max_channels": os.environ.get("NCCL_MAX_NCHANNELS"),
                "cus_left_to_rccl": getattr(tr.dp, "reserved_cus", 0),
                "grad_bytes": int(tr.dp.numel * 4), "exchanges": len(ar),
                "allreduce_ms": float(stats[0]), "allreduce_exposed_ms": float(stats[1]),
                "replica_checksum_spread": spread,
                # start-up self-check of the captured exchange (train_step._capture_checked); None: forced by env
                "graph_comm_check": getattr(tr.stepper, "graph_comm_check", None)}
        assert ranks_seen == world, f"RCCL saw {ranks_seen} ranks, expected {world}"
        assert spread == 0.0, f"replicas diverged: checksum spread {spread}"

    # Kernel census: `--census-steps` more steps of the SAME work, eagerly issued (no graph), un-pipelined, EVERY launch
    # of the library bracketed by HIP events on the launching stream (msr3d_amd/_lib.py::_Entry) -> mean duration per
    # entry point and launches per step.  Run on every rank (the steps hold the gradient exchange), read on rank 0;
    # after the exchange statistics above were taken (its steps exchange gradients too).
    census = kernel_census(tr, batches, args.census_steps, run_window, args.warmup + 2 * args.steps) if args.census_steps > 0 else {}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = B * args.accum * world * args.steps / elapsed
        per_step = sorted(a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:]))
        pct = lambda q: per_step[min(len(per_step) - 1, int(q * len(per_step)))]   # noqa: E731
        # Per-launch durations: the census after the timed region (eager, un-pipelined, HIP events around every launch).
        cus = lambda k: census.get(k, {}).get("us", 0.0)               # noqa: E731
        rows_keys = {1: ("msr3d_sa_plan12", "msr3d_sa_level1_rows"), 2: ("msr3d_sa_level2_rows",), 3: ("msr3d_sa_level3_tiles",)}
        kern_ms = {}
        for lvl in (1, 2, 3):
            t = sum(cus(k) for k in rows_keys[lvl])
            if lvl < 3 and not cus(rows_keys[lvl][-1]):                  # the all-rows kernels (MSR3D_SA_ROWS=0 / f32 MFMA path)
                t = cus(f"msr3d_sa_level_split[{lvl}]") or cus(f"msr3d_sa_level[{lvl}]")
            if lvl == 3 and not t:
                t = cus("msr3d_sa_level_split[3]") or cus("msr3d_sa_level[3]")
            kern_ms[f"msr3d_sa_level{lvl}"] = t / 1e3 if t else None
        # Roofline leg: the three SharedMLP levels of the frozen encoder (98 % of the path's FLOPs), each priced as
        #   frac = FLOPs of the DISTINCT rows the result needs / launch time / peak
        # with the nominal figure of SURVEY.md 8(d) (every one of the 32 neighbourhood slots multiplied: 512 x
        # (131*128 + 128*128 + 128*256) MACs x 2 = 67.5 MFLOP per object at level 2) beside it as `nominal_tflops`:
        # ball_query repeats a neighbourhood's first hit in its empty slots, the SharedMLP acts row by row and max is
        # idempotent, so the kernels of round 5 multiply only the different rows -- same bits, and a rate priced on the
        # nominal FLOPs would exceed the roof and read as "work skipped".  Row counts: one untimed pass per resident batch.
        from msr3d_amd.pointnet2 import fused as _fused
        split = _fused._sa_mma[0] in ("split", "split2")
        reduced = _fused._sa_mma[0] == "split2"          # LABELLED variant: 2 bf16 terms per operand, 3 products
        SPLIT_PRODUCTS = 3 if reduced else 6
        rows_on = split and _fused._sa_rows[0]
        objs_per_launch = float(Bcall * O) * (calls if window else 1)    # (a whole window per encoder launch)
        if args.skip_padded:      # only real objects are encoded: count them over the timed steps
            first = args.warmup + (1 if pipe else 0)    # (pipelined: step i encodes batch i + 1)
            counts = [int(batches[(first + i) % n_resident]["obj_masks"].sum()) for i in range(args.steps)]
            objs_per_launch = sum(counts) / max(len(counts), 1)
        roof = None
        if not args.unfrozen:
            net = model.visual_prompter.obj_encoder.pcd_net
            stats = [_fused.row_statistics(net, bt["obj_fts"].reshape(-1, P, 6)) for bt in batches]
            scale = (calls if window else 1)
            peak = MFMA_BF16_PEAK_TF / SPLIT_PRODUCTS if split else MFMA_F32_PEAK_TF
            levels = {}
            for lvl in (1, 2, 3):
                ms = kern_ms.get(f"msr3d_sa_level{lvl}") or 0.0
                mean = lambda k: scale * sum(st[lvl][k] for st in stats) / len(stats)   # noqa: E731
                nom, dis, pip = mean("nominal_flop"), mean("distinct_flop"), mean("pipe_flop")
                if not rows_on:
                    dis = pip = nom
                tf = lambda f: f / (ms * 1e-3) / 1e12 if ms > 0 else 0.0                  # noqa: E731
                levels[f"level{lvl}"] = {
                    "kernel_ms": ms, "distinct_rows": mean("distinct_rows") if rows_on else mean("nominal_rows"),
                    "nominal_rows": mean("nominal_rows"), "achieved_tflops": tf(dis), "frac": tf(dis) / peak,
                    "nominal_tflops": tf(nom), "mfma_executed_tflops": tf(pip) * (SPLIT_PRODUCTS if split else 1),
                    "mfma_pipe_frac": (tf(pip) * SPLIT_PRODUCTS / MFMA_BF16_PEAK_TF) if split else tf(pip) / MFMA_F32_PEAK_TF}
            dom = max(levels, key=lambda k: levels[k]["kernel_ms"])
            level_flops = {}
            for lvl in (1, 2, 3):                                     # the census rows of the level kernels: distinct-row FLOPs
                for k in rows_keys[lvl][-1:] + (f"msr3d_sa_level_split[{lvl}]", f"msr3d_sa_level[{lvl}]"):
                    level_flops[k] = levels[f"level{lvl}"]["achieved_tflops"] * 1e12 * levels[f"level{lvl}"]["kernel_ms"] * 1e-3
            names = {"level1": "sa1_rows_kernel (+ sa1_plan_kernel; msr3d_sa_level1_rows)" if rows_on else "sa1_split_kernel",
                     "level2": "sa2_rows_kernel (+ sa2_plan_kernel; msr3d_sa_level2_rows)" if rows_on else "sa2_split_kernel",
                     "level3": "sa3_tiles_kernel (msr3d_sa_level3_tiles)" if cus("msr3d_sa_level3_tiles") else "sa3_split4_kernel (msr3d_sa_level_split level 3)"}
            if not split:
                names = {k: f"sa{k[-1]}_kernel (msr3d_sa_level, f32-input MFMA)" for k in names}
            d = levels[dom]
            traffic_per_obj, traffic_src = sa2_traffic_from_profiles(names[dom].split(" ")[0])
            if traffic_per_obj and rows_on and dom in ("level1", "level2"):        # the level = its plan launch + its rows launch
                plan_per_obj, _ = sa2_traffic_from_profiles(f"sa{dom[-1]}_plan_kernel")
                traffic_per_obj += plan_per_obj or 0.0
            roof = {"bound": "mfma", "kernel": names[dom], "achieved": d["achieved_tflops"], "peak": peak, "unit": "TFLOP/s",
                    "frac": d["frac"], "nominal_tflops": d["nominal_tflops"],
                    "achieved_note": "FLOPs of the distinct neighbourhood rows (what the result needs; SURVEY.md 8(d)'s nominal "
                                     "figure counts every padded slot: nominal_tflops) / the launch's duration",
                    "peak_note": (f"bf16 dense MFMA peak 2500 TFLOP/s / {SPLIT_PRODUCTS} products per product" if split
                                  else "f32-input MFMA peak"),
                    "mfma_executed_tflops": d["mfma_executed_tflops"], "mfma_pipe_frac": d["mfma_pipe_frac"],
                    "vs_f32_mfma_peak": d["achieved_tflops"] / MFMA_F32_PEAK_TF,
                    "dtype": ("VARIANT split2: f32 operands as 2 bf16 terms, 3 x v_mfma_f32_16x16x32_bf16 per product (~16 significant "
                              "bits per product, fp32 accumulate): NOT fp32 accuracy -- encoder output rel-L2 vs the fp32 path "
                              "in extra.split2_rel_l2") if reduced else
                             ("f32 operands as 3 bf16 terms, 6 x v_mfma_f32_16x16x32_bf16 per product, fp32 accumulate "
                              "(fp32 accuracy; MSR3D_SA_MMA=f32 selects the f32-input MFMA kernels)") if split
                    else "f32-input MFMA (v_mfma_f32_16x16x4_f32)",
                    "levels": levels, "constant_objects_per_launch": scale * sum(st["constant_objects"] for st in stats) / len(stats),
                    "traffic": traffic_per_obj * objs_per_launch if traffic_per_obj else None,
                    "traffic_unit": f"bytes/launch (PMC passes of tools/pmc_sa.sh, {traffic_src})",
                    "kernel_ms": d["kernel_ms"]}
            # The line's kernel is the LONGEST kernel of the step, whichever part it is in (round 5's line looked at the
            # three levels only and missed the weight-gradient launch); `kernels`: every launch >= 15 us.
            table, pmc_files = census_table(census, model, objs_per_launch, args.llm_hidden, level_flops)
            big = {k: r for k, r in table.items() if r["us"] >= 15.0}
            rated = {k: r for k, r in big.items() if "frac" in r}
            if rated and split:
                top = max(rated, key=lambda k: rated[k]["us"])
                r = rated[top]
                roof.update({"kernel": f"{r['kernel']} ({top})", "achieved": r["achieved_tflops"], "frac": r["frac"],
                             "kernel_ms": r["us"] / 1e3, "useful_gflop_per_launch": r["useful_gflop"],
                             "launches": int(round(r["launches_per_step"] * args.census_steps)),
                             "mfma_busy_pmc": r.get("mfma_busy_pmc"),
                             "traffic": r.get("hbm_bytes_pmc", roof["traffic"] if top in level_flops else None)})
                if top not in level_flops:
                    roof["achieved_note"] = ("fp32 FLOPs the result needs (each counted once, whatever the number of bf16 MFMA "
                                             "products it takes) / the launch's duration")
                    roof["traffic_unit"] = "bytes/launch (FETCH_SIZE x 2 + WRITE_SIZE of the newest PMC pass) or null"
                    for k in ("nominal_tflops", "mfma_executed_tflops", "mfma_pipe_frac", "vs_f32_mfma_peak"):
                        roof.pop(k, None)
            roof["kernels"] = big
            roof["kernels_note"] = ("every launch of the step >= 15 us: mean duration, launches per step, fp32-equivalent useful "
                                    "GFLOP per launch, frac of 2500 / 6 TFLOP/s; mfma_busy_pmc = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs "
                                    f"x GRBM_GUI_ACTIVE / 8) of the newest committed counter pass ({', '.join(pmc_files) or 'none'})")
            roof["timed"] = (f"kernel census: {args.census_steps} eager (no graph), un-pipelined steps after the timed region, HIP "
                             "events on the launching stream around EVERY launch; the timed region itself carries no events")
            roof["step_launch_us_sum"] = sum(r["us"] * r["launches_per_step"] for r in table.values())
        else:
            # variant line: the fused frozen-encoder launches are not on this path (the SharedMLPs run as
            # group_rows -> token GEMM -> BatchNorm(train) kernels under autograd); no roofline leg
            roof = {"bound": "mfma", "kernel": None, "achieved": None, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                    "frac": None, "note": "unfrozen-backbone variant: see DESIGN.md 4.1a", "traffic": None}
        line = {
            "metric": f"MSQA train samples/sec (whole node), {O} obj x {P} pts",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step,
            "ms_per_step_percentiles": {"p10": pct(0.10), "p50": pct(0.50), "p90": pct(0.90),
                                        "note": "HIP events on the compute stream between steps, this rank, in a pass of its own after the timed region "
                                                "(an event record a step costs the step 3-5 us)"},
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("VARIANT (MSR3D_SA_MMA=split2): frozen encoder on 2-term bf16 splits, 3 MFMA products per product (~16 bits); "
                      "trainable part f32-accurate (6 products)") if (reduced and not args.unfrozen) else
                     "f32 (every product as 6 bf16 MFMA products of exact 3-way bf16 splits, fp32 accumulate: fp32 accuracy)" if (split and not args.unfrozen) else "f32",
            "data": "synthetic",
            "config": {"workload": "configs/msr3d.yaml hot path (OSE3DSituation + llm_proj, "
                                   "frozen PointNet++), synthetic ScanNet-like scenes",
                       "objects": O, "points": P, "per_gpu_batch": B, "global_batch": B * world * args.accum,
                       "grad_accumulation": args.accum, "window_step": wstep,
                       "llm_hidden": args.llm_hidden, "situation_type": args.situation_type,
                       "step": "fwd+bwd+allreduce+clip+AdamW, LLM excluded",
                       "backbone": "unfrozen (freeze: False, BatchNorm in training mode)" if args.unfrozen
                       else "frozen (every shipped config)",
                       "hip_graph": not args.no_graph, "encoder_prefetch": args.pipeline or (dist_on and whole_graph),
                       "padded_slots_skipped": args.skip_padded,
                       "objects_encoded_per_step": objs_per_launch,
                       "allreduce_hidden_behind_next_encoder": dist_on and not whole_graph and not args.unfrozen,
                       "exchange_inside_graph": dist_on and whole_graph,
                       "inputs": ("built per step on the device from HBM-resident scans "
                                  "(msr3d_preprocess_pcd)" if args.from_store else
                                  "pinned host memory, copied over PCIe every step" if args.host_inputs
                                  else "resident in HBM"),
                       "parallelism": f"dp{world}"},
            "roofline": roof,
            "kernels_ms": kern_ms,
        }
        from msr3d_amd import hipops as _hipops
        if _hipops._attn_mma[0] != "f32":     # a LABELLED reduced-precision object attention (never the headline)
            line["dtype"] = (f"VARIANT (MSR3D_ATTN_MMA={_hipops._attn_mma[0]}): object-attention QK^T / PV "
                             + {"bf16": "and the backward's products on one bf16 MFMA product",
                                "fp8_bf16": "on OCP e4m3 MFMA (v_mfma_f32_16x16x32_fp8_fp8), the backward's products on bf16",
                                "fp8": "on OCP e4m3 MFMA"}.get(_hipops._attn_mma[0], "") + "; otherwise " + line["dtype"])
            line["config"]["attention_mma"] = _hipops._attn_mma[0]
        if census:
            line["census_us"] = {k: [round(v["us"], 2), v["per_step"]] for k, v in sorted(census.items())}
        if comm is not None:
            line["comm"] = comm
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, args.cpu_baseline_seconds)
        plain = not (dist_on or args.unfrozen or args.skip_padded or args.no_pipeline or args.from_store or args.host_inputs
                     or args.accum > 1 or args.no_graph or args.time_all_kernels or (O, P) != (60, 1024) or args.dense
                     or args.llm_hidden != 4096 or args.batch != 16 or args.no_cpu_baseline
                     or args.situation_type != "as_transform_for_objects")
        if reduced and not args.unfrozen:
            # how far the reduced variant's encoder output is from the fp32-accurate path, on the first resident batch
            net = model.visual_prompter.obj_encoder.pcd_net
            pts0 = batches[0]["obj_fts"].reshape(-1, P, 6)
            with torch.no_grad():
                y2 = _fused.forward(net, pts0).double()
                prev = _fused.set_sa_mma("split")
                y6 = _fused.forward(net, pts0).double()
                _fused.set_sa_mma(prev)
            line.setdefault("extra", {})["split2_rel_l2"] = float((y2 - y6).norm() / y6.norm())
        if plain and not args.no_extra and not reduced:
            # SURVEY 8(f) rank 4, where > 99 % of a real step's time lives: the FULL step of BASELINE configs[1] (prompter ->
            # llm_proj -> scatter -> 32 LoRA-Llama layers at the Vicuna-7B shapes -> head -> CE -> backward into the
            # prompter -> clip + AdamW), in this process, after the headline: labelled secondary figures, never `value`
            extra = line.setdefault("extra", {})
            # the object-attention kernels' place against north_star's ">= 50 % MFMA utilisation" clause, in the record
            kt = (roof or {}).get("kernels", {})
            extra["object_attention"] = {
                k.split("[")[1][:-1]: {q: kt[k].get(q) for q in ("kernel", "us", "useful_gflop", "frac", "mfma_busy_pmc")}
                for k in ("msr3d_scene_block[attn_fwd]", "msr3d_scene_block[attn_bwd]") if k in kt}
            extra["object_attention"]["note"] = ("fp32-accurate form (6 bf16 MFMA products per product); frac: useful fp32 FLOPs / "
                                                 "time / (2500 / 6 TFLOP/s); mfma_busy_pmc: matrix-pipe busy cycles of the newest "
                                                 "committed counter pass.  60 tokens x 32 channels a head: DESIGN.md 4.2c")
            variants = [("dense_neighbourhoods", dict(dense=True),
                         "the distinct-row kernels' worst case: every ball query finds >= 32 different points, all 60 slots real"),
                        ("as_object", dict(situation_type="as_object"),
                         "configs/leo_3_dataset_pure_txt.yaml's prompter (anchor token, L = 61) at 60 x 1024"),
                        ("no_pipeline", dict(no_pipeline=True, pipeline=False),
                         "encoder and trainable part back to back on one stream (rounds 1-5's default schedule)")]
            if os.environ.get("MSR3D_TRAIN_MMA", "f32") == "f32":
                variants.append(("bf16_trainable", dict(train_mma="bf16"),
                                 "LABELLED reduced variant (MSR3D_TRAIN_MMA=bf16): the trainable part's products on ONE bf16 MFMA "
                                 "product of bf16-rounded operands (fp32 accumulate) instead of six -- not fp32 accuracy"))
            for name, over, note in variants:
                try:
                    extra[name] = dict(measure_variant(args, device, **over), note=note)
                except Exception as e:                 # noqa: BLE001 -- the headline line must still be printed
                    extra[name] = {"error": repr(e)[:300], "note": note}
            extra["full_step"] = full_step_extra(args, tr, model, batches, device)
    else:
        line = None
    if dist_on:
        dist.destroy_process_group()
        # RCCL's rank 0 writes a version banner through C stdio, which sits in its buffer until exit:
        # push it out now so that the JSON line below is the LAST thing on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
    if line is not None:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
