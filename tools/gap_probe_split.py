"""gap_probe for the multi-rank schedule over a one-rank RCCL communicator: host enqueue time per
step and HIP-event intervals [load | graph fwd/bwd | exchange start .. encoder-ahead | optimiser].
    python tools/gap_probe_split.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ["MSR3D_DP_FORCE_EXCHANGE"] = "1"
import torch
import torch.distributed as dist
import bench
sys.argv = ["bench.py", "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
from msr3d_amd.synth import synth_batch
model = bench.build(args, dev)
B = args.batch
batches = [synth_batch(1000 + i, B, O=60, P=1024, device=dev) for i in range(4)]
tr = bench.Trainer(model, dev, batches[0], args.llm_hidden, use_graph=True)
st = tr.stepper
ev = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((name, e))
host = {}
def wrap(obj, attr, before, after):
    f = getattr(obj, attr)
    def g(*a, **k):
        if before: mark(before)
        t = time.perf_counter()
        r = f(*a, **k)
        host[attr] = host.get(attr, 0.0) + time.perf_counter() - t
        if after: mark(after)
        return r
    setattr(obj, attr, g)
wrap(st, "_load", "load", "loaded")
wrap(st.graph, "replay", None, "graph_done")
wrap(st.dp, "start", None, "ar_started")
wrap(st, "encode_ahead", None, "encoded")
wrap(st.dp, "wait", None, "ar_waited")
wrap(st.opt, "step", None, "opt_done")
for i in range(6): tr.step(batches[i % 4], batches[(i + 1) % 4])
torch.cuda.synchronize(); ev.clear(); host.clear()
t0 = time.perf_counter()
N = 20
for i in range(N):
    tr.step(batches[i % 4], batches[(i + 1) % 4])
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
import statistics as S, collections
d = collections.defaultdict(list)
for (a, ea), (b, eb) in zip(ev, ev[1:]):
    d[f"{a}->{b}"].append(ea.elapsed_time(eb) * 1e3)
print(f"host enqueue per step {t_host/N*1e3:.3f} ms; wall per step {t_all/N*1e3:.3f} ms")
print("host ms per step by call:", {k: round(v / N * 1e3, 3) for k, v in host.items()})
for k, v in d.items():
    print(f"  {k:28s} median {S.median(v):8.1f} us")
dist.destroy_process_group()
