#!/usr/bin/env python3
"""List the small ATen ops (copies, fills, adds, sums) of one SS_Conv_SSM block fwd+bwd with their input shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.modules import SS_Conv_SSM
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = 64
dim, hw = [(96, 56), (192, 28), (384, 14), (768, 7)][stage]
dev = torch.device("cuda:0")
blk = SS_Conv_SSM(hidden_dim=dim, drop_path=0.0, norm_layer=torch.nn.LayerNorm).to(dev).train()
x = torch.randn(B, hw, hw, dim, device=dev, requires_grad=True)
for _ in range(3):
    blk.zero_grad(set_to_none=True); x.grad = None
    blk(x).sum().backward()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    blk.zero_grad(set_to_none=True); x.grad = None
    y = blk(x); y.backward(torch.ones_like(y)); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True)
        if e.key in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::sum", "aten::mul", "aten::cat", "aten::stack", "aten::exp", "aten::neg")]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:40]:
    print(f"{e.key:<14} calls {e.count:>3} self_cuda_us {e.self_device_time_total:>9.1f}  {str(e.input_shapes)[:150]}")
