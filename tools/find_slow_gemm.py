#!/usr/bin/env python3
"""Diagnostic: list GEMM-type ATen ops of a full training step (tuned GEMM table on) slower than a threshold, with shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
from medmamba_amd.tuning import enable_tuned_gemms
if os.environ.get("MM_TUNED_GEMMS", "1") == "1":
    enable_tuned_gemms()
dev = torch.device("cuda:0")
torch.manual_seed(42)
net = VSSM(num_classes=6, **MEDMAMBA_CONFIGS["S"]).to(dev).train()
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
x = torch.randn(64, 3, 224, 224, device=dev); y = torch.randint(0, 6, (64,), device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(net(x), y); loss.backward(); opt.step()
for _ in range(4): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::mm", "aten::bmm", "aten::addmm", "aten::baddbmm", "aten::addmm_", "aten::baddbmm_", "aten::linear", "aten::matmul")]
rows.sort(key=lambda e: -e.self_device_time_total / max(1, e.count))
for e in rows[:14]:
    print(f"{e.key:<14} calls {e.count:>3} avg_us {e.self_device_time_total / max(1, e.count):>9.1f}  {str(e.input_shapes)[:140]}")
