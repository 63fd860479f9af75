"""Where a step's wall time goes without a profiler attached: HIP events around the eager encoder
(+ input copies), the captured graph, and the GPU idle gap between one step's last kernel and the
next step's first (host run-ahead check).  python tools/gap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sys.argv = ["bench.py", "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
from msr3d_amd.synth import synth_batch
model = bench.build(args, dev)
B = args.batch
batches = [synth_batch(1000 + i, B, O=60, P=1024, device=dev) for i in range(4)]
tr = bench.Trainer(model, dev, batches[0], args.llm_hidden, use_graph=True)
st = tr.stepper
ev = []
orig_load = st._load
def load(b):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(("load", e))
    orig_load(b)
    e2 = torch.cuda.Event(enable_timing=True); e2.record(); ev.append(("loaded", e2))
st._load = load
for i in range(5): tr.step(batches[i % 4])
torch.cuda.synchronize(); ev.clear()
t0 = time.perf_counter()
for i in range(20):
    tr.step(batches[i % 4])
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(("end", e))
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
seq = ev
enc, graph, gap = [], [], []
for i in range(len(seq) - 1):
    (a, ea), (b, eb) = seq[i], seq[i + 1]
    d = ea.elapsed_time(eb) * 1e3
    if a == "load" and b == "loaded": enc.append(d)
    if a == "loaded" and b == "end": graph.append(d)
    if a == "end" and b == "load": gap.append(d)
import statistics as S
print(f"host enqueue per step {t_host/20*1e3:.3f} ms; wall per step {t_all/20*1e3:.3f} ms")
print(f"encoder+copies {S.median(enc):.1f} us, graph {S.median(graph):.1f} us, gap end->next load {S.median(gap):.1f} us")
