#!/usr/bin/env python3
"""Every device kernel of one SS_Conv_SSM block (fwd + bwd), in launch order, with the ATen / autograd op that launched it.
usage: tools/block_launch_list.py STAGE [BATCH]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.modules import SS_Conv_SSM
from medmamba_amd.tuning import enable_tuned_gemms
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dim, hw = [(96, 56), (192, 28), (384, 14), (768, 7)][stage]
enable_tuned_gemms()
dev = torch.device("cuda:0")
blk = SS_Conv_SSM(hidden_dim=dim, drop_path=0.1, norm_layer=torch.nn.LayerNorm).to(dev).train()
x = torch.randn(B, hw, hw, dim, device=dev, requires_grad=True)
for _ in range(3):
    blk.zero_grad(set_to_none=True); blk(x).sum().backward()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    blk.zero_grad(set_to_none=True)
    y = blk(x); y.backward(torch.ones_like(y)); torch.cuda.synchronize()
rows = []
for e in prof.events():
    for k in e.kernels:
        fr = [f for f in (e.stack or []) if "medmamba_amd" in f]
        rows.append((e.time_range.start, e.name, k.name, k.duration, (fr[0].split("medmamba_amd/")[-1] if fr else "")))
rows.sort()
seen = set(); n = 0; tot = 0.0
for t, op, kn, us, where in rows:
    key = (t, kn)
    if key in seen: continue
    seen.add(key); n += 1; tot += us
    print(f"{n:3d} {us:7.1f} us  {op[:30]:<30} {where[:44]:<44} {kn[:60]}")
print("launches", n, "device us", round(tot, 1))
