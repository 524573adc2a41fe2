#!/usr/bin/env python3
"""Diagnostic: host time of optimizer.step() (fused AdamW) and zero_grad() for MedMamba-S (265 parameters)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
dev = torch.device("cuda:0")
net = VSSM(num_classes=6, **MEDMAMBA_CONFIGS["S"]).to(dev).train()
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
for p in net.parameters(): p.grad = torch.randn_like(p)
for _ in range(3): opt.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): opt.step()
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"fused AdamW step: {(t1 - t0) / 50 * 1e3:.3f} ms host time, {len(list(net.parameters()))} parameters")
grads = [torch.randn_like(p) for p in net.parameters()]
t0 = time.perf_counter()
for _ in range(50):
    for p, g in zip(net.parameters(), grads): p.grad = g
    opt.zero_grad(set_to_none=True)
t1 = time.perf_counter()
print(f"assign + zero_grad(set_to_none): {(t1 - t0) / 50 * 1e3:.3f} ms host time")
