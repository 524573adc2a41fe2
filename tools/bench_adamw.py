#!/usr/bin/env python3
"""AdamW update of MedMamba-S alone: optim.FusedAdamW with mm_adamw_step vs torch's fused multi-tensor kernel (hipEvents, 30 steps)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd import optim
from medmamba_amd.modules import MEDMAMBA_CONFIGS, VSSM
dev = torch.device("cuda:0")
size = sys.argv[1] if len(sys.argv) > 1 else "S"
net = VSSM(num_classes=6, **MEDMAMBA_CONFIGS[size]).to(dev)
for p in net.parameters():
    p.grad = torch.randn_like(p) * 1e-2
n = sum(p.numel() for p in net.parameters())
for name, hip in (("mm_adamw_step", True), ("torch fused", False), ("mm_adamw_step", True), ("torch fused", False)):
    optim._HIP_ADAMW = hip
    opt = optim.FusedAdamW(net.parameters(), lr=1e-4, weight_decay=1e-4)
    for _ in range(5): opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): opt.step()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print(f"{size}: {name:14s} {us:7.1f} us per step  ({28 * n / us / 1e6:.2f} TB/s of 28 B per element, {len(list(net.parameters()))} tensors, {n / 1e6:.1f} M elements)")
