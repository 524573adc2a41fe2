#!/usr/bin/env python3
"""ATen ops of one MedMamba-S training step that are neither GEMMs nor convolutions, by device time (which small launches are
left around the HIP kernels).  usage: tools/profile_step_small_ops.py [BATCH]"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
from medmamba_amd.tuning import enable_tuned_gemms
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
enable_tuned_gemms()
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = VSSM(num_classes=6, **MEDMAMBA_CONFIGS["S"]).to(dev).train()
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
x = torch.randn(B, 3, 224, 224, device=dev); y = torch.randint(0, 6, (B,), device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(net(x), y); loss.backward(); opt.step()
for _ in range(4): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, set()])
for e in prof.events():
    if not e.kernels: continue
    if any(k in e.name for k in ("mm", "conv", "Conv")): continue
    a = agg[(e.name, str(e.input_shapes)[:110])]
    a[0] += 1; a[1] += sum(k.duration for k in e.kernels); a[2].update(k.name[:60] for k in e.kernels)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print(f"total device us of these ops: {tot:.0f} in {sum(v[0] for _, v in rows)} calls")
for (n, shp), (c, us, ks) in rows[:70]:
    print(f"{us:8.1f} us {c:4d}x {n:<28} {shp:<112} {sorted(ks)[0][:50]}")
