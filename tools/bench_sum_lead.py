#!/usr/bin/env python3
"""Diagnostic: ops.sum_lead (mm_sum_lead) against torch.sum(0) on the batch-sum shapes of a MedMamba-S step at 64 images."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd import ops
dev = torch.device("cuda:0")
def t(fn, it=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for shape in [(64, 2, 70, 96), (64, 4, 96, 3), (64, 192, 48), (64, 48, 96), (64, 2, 76, 192), (64, 4, 192, 6), (64, 384, 96), (64, 96, 192),
              (64, 48, 48), (64, 96, 96), (16, 48, 432), (16, 96, 864), (64, 768, 384), (256, 192)]:
    x = torch.randn(*shape, device=dev); out = torch.empty(shape[1:], device=dev)
    a = t(lambda: ops.sum_lead(x, out=out)); b = t(lambda: torch.sum(x, 0, out=out))
    print(f"{str(shape):<22} {x.numel() * 4 / 1e6:7.2f} MB   mm_sum_lead {a:6.1f} us   torch.sum {b:6.1f} us")
