#!/usr/bin/env python3
"""torch.profiler op-level breakdown of one SS_Conv_SSM block (fwd+bwd) at a MedMamba-S stage shape.
usage: python tools/profile_block.py [stage 0-3] [batch]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.modules import SS_Conv_SSM

stage = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dim, hw = [(96, 56), (192, 28), (384, 14), (768, 7)][stage]
dev = torch.device("cuda:0")
torch.manual_seed(0)
blk = SS_Conv_SSM(hidden_dim=dim, drop_path=0.0, norm_layer=torch.nn.LayerNorm).to(dev).train()
x = torch.randn(B, hw, hw, dim, device=dev, requires_grad=True)
for _ in range(3):
    blk(x).sum().backward()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        y = blk(x)
        y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=45, max_shapes_column_width=70))
