#!/usr/bin/env python3
"""Diagnostic: mm_channel_sum_nchw vs torch.sum(dim=(0,2,3)) on the conv branch's gradient shapes (B = 64)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.ops import channel_sum_nchw
dev = torch.device("cuda:0")
def t(fn, it=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for C, hw in [(48, 56), (96, 28), (192, 14), (384, 7)]:
    x = torch.randn(64, C, hw, hw, device=dev)
    print(f"C={C} {hw}x{hw}: ours {t(lambda: channel_sum_nchw(x)):6.1f} us (incl. zero-fill), torch {t(lambda: x.sum(dim=(0, 2, 3))):6.1f} us")
