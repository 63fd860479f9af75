import sys, torch
sys.path.insert(0, ".")
sys.argv = [sys.argv[0]] + sys.argv[1:]
exec(open("tools/prof_llm_layer.py").read().split("from torch.profiler")[0])
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
ev = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::add", "aten::zero_", "aten::fill_", "aten::to", "aten::_to_copy")]
ev.sort(key=lambda e: -e.device_time_total)
for e in ev[:14]:
    print(f"{e.key:18s} n={e.count:3d} cuda_us={e.device_time_total:9.1f} shapes={e.input_shapes}")
print("---- stacks of the biggest copies")
ev2 = [e for e in prof.key_averages(group_by_stack_n=6) if e.key in ("aten::copy_",)]
ev2.sort(key=lambda e: -e.device_time_total)
for e in ev2[:5]:
    print(e.count, round(e.device_time_total, 1), [s for s in e.stack if "msr3d_amd" in s][:3])
