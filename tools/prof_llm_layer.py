"""torch.profiler view of one LoRA-Llama decoder layer forward + backward (Vicuna-7B shape, 4 x 576 tokens): which aten
ops (torch glue) surround the C-ABI kernels.   python tools/prof_llm_layer.py"""
import sys

import torch

sys.path.insert(0, ".")
from msr3d_amd.llm import LoRALlamaDecoderLayer  # noqa: E402

dev = torch.device("cuda", 0)
Bq, T, Hd, NH, FF = 4, 576, 4096, 32, 11008
torch.manual_seed(0)
layer = LoRALlamaDecoderLayer(Hd, NH, FF, r=16, lora_alpha=16, device=dev, base=("fp8" if "--fp8" in sys.argv else "bf16"))
with torch.no_grad():
    for grp in (layer.self_attn, layer.mlp):
        for m in grp.values():
            m.load_base_weight(torch.randn(m.out_features, m.in_features, device=dev) / m.in_features ** 0.5)
            m.lora_B.weight.normal_(std=0.02)
x = torch.randn(Bq, T, Hd, device=dev).bfloat16().requires_grad_(True)
keep = torch.ones(Bq, T, dtype=torch.uint8, device=dev)
gy = (torch.randn(Bq, T, Hd, device=dev) * 0.01).bfloat16()


def step():
    for p in layer.parameters():
        p.grad = None
    x.grad = None
    layer(x, attention_mask=keep).backward(gy)
    with torch.no_grad():
        for p in layer.parameters():
            if p.requires_grad:
                torch.autograd.graph.increment_version(p)      # as an optimiser step would: shadows rebuild


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
