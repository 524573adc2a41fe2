#!/usr/bin/env python3
"""GEMM-type ATen ops of one SS_Conv_SSM block (fwd + bwd) with input shapes AND the device kernel each one launched, with
the recorded GEMM table active (what bench.py runs).  usage: tools/profile_block_gemm_kernels.py STAGE [BATCH] [SIZE]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.modules import SS_Conv_SSM
from medmamba_amd.tuning import enable_tuned_gemms
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
size = sys.argv[3] if len(sys.argv) > 3 else "S"
dims = {"S": [(96, 56), (192, 28), (384, 14), (768, 7)], "B": [(128, 96), (256, 48), (512, 24), (1024, 12)]}[size]
dim, hw = dims[stage]
print("tuned gemms:", enable_tuned_gemms())
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
names = ("aten::mm", "aten::bmm", "aten::addmm", "aten::baddbmm", "aten::addmm_", "aten::baddbmm_")
rows = []
for e in prof.events():
    if e.name in names and e.kernels:
        rows.append((sum(k.duration for k in e.kernels), e.name, str(e.input_shapes)[:90], " + ".join(k.name[:70] for k in e.kernels)))
rows.sort(key=lambda r: -r[0])
for us, n, shp, ks in rows:
    print(f"{us:8.1f} us  {n:<14} {shp:<92} {ks}")
print("total gemm us", sum(r[0] for r in rows))
