#!/usr/bin/env python3
"""List the GEMM-type ATen ops (mm / bmm / addmm / baddbmm / matmul) of one SS_Conv_SSM block fwd+bwd with shapes."""
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
    blk.zero_grad(set_to_none=True); blk(x).sum().backward()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    blk.zero_grad(set_to_none=True)
    y = blk(x); y.backward(torch.ones_like(y)); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::mm", "aten::bmm", "aten::addmm", "aten::baddbmm", "aten::addmm_", "aten::baddbmm_")]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = 0
for e in rows:
    tot += e.self_device_time_total
    print(f"{e.key:<14} calls {e.count:>3} self_cuda_us {e.self_device_time_total:>9.1f}  {str(e.input_shapes)[:150]}")
print("total gemm us", tot)
allk = sum(e.self_device_time_total for e in prof.key_averages())
print("total device us", allk)
