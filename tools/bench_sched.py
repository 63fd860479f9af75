"""Per-launch timing of the trainable schedule (msr3d_amd/fused_model.py): runs the bench's model
eagerly with HIP events around each launch, prints mean microseconds per call site."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msr3d_amd import fused_model, _lib
from msr3d_amd.synth import synth_batch

args = bench.parse()
dev = torch.device("cuda", 0)
model = bench.build(args, dev)
B = args.batch
batch = synth_batch(1, B, O=60, P=1024, device=dev)
tr = bench.Trainer(model, dev, batch, args.llm_hidden, use_graph=False)
sched = model._schedule
rec = collections.OrderedDict()
order = []

def wrap(name):
    orig = getattr(fused_model.PrompterSchedule, name)
    def f(self, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(self, *a, **kw); e1.record()
        order.append((name, kw.get("pro", None), kw.get("N", None), e0, e1))
        return r
    setattr(fused_model.PrompterSchedule, name, f)
wrap("_strip"); wrap("_multi")
for it in range(6):
    order.clear()
    tr.step(batch)
torch.cuda.synchronize()
tot = 0
for i, (name, pro, N, e0, e1) in enumerate(order):
    t = e0.elapsed_time(e1) * 1e3
    tot += t
    print(f"{i:3d} {name:8s} pro={pro} N={N} {t:8.1f} us")
print("sum", tot)
