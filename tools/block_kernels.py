#!/usr/bin/env python3
"""Run one SS_Conv_SSM block (fwd+bwd, single stream) at a MedMamba-S stage shape N times — meant to be run under
`rocprofv3 --kernel-trace --stats` to get the GPU-side duration of every kernel of the block at that stage.
usage: python tools/block_kernels.py [stage 0-3] [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MM_TWO_STREAMS", "0")
from medmamba_amd.modules import SS_Conv_SSM
from medmamba_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dim, hw = [(96, 56), (192, 28), (384, 14), (768, 7)][stage]
dev = torch.device("cuda:0")
torch.manual_seed(0)
blk = SS_Conv_SSM(hidden_dim=dim, drop_path=0.1, norm_layer=torch.nn.LayerNorm).to(dev).train()
x = torch.randn(64, hw, hw, dim, device=dev, requires_grad=True)
g = torch.randn(64, hw, hw, dim, device=dev)
for _ in range(iters + 3):
    blk.zero_grad(set_to_none=True); x.grad = None
    blk(x).backward(g)
torch.cuda.synchronize()
