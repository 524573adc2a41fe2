#!/usr/bin/env python3
"""Diagnostic: the conv branch (BN-conv3x3-BN-ReLU-conv3x3-BN-ReLU-conv1x1) fwd+bwd per stage, NCHW vs channels_last."""
import torch, torch.nn as nn
dev = torch.device("cuda:0")
B = 64
def branch(h):
    return nn.Sequential(nn.BatchNorm2d(h), nn.Conv2d(h, h, 3, 1, 1), nn.BatchNorm2d(h), nn.ReLU(), nn.Conv2d(h, h, 3, 1, 1),
                         nn.BatchNorm2d(h), nn.ReLU(), nn.Conv2d(h, h, 1, 1)).to(dev).train()
def t(fn, it=10):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for h, hw, nblk in [(48, 56, 2), (96, 28, 2), (192, 14, 8), (384, 7, 2)]:
    res = []
    for fmt in (torch.contiguous_format, torch.channels_last):
        net = branch(h).to(memory_format=fmt)
        x = torch.randn(B, h, hw, hw, device=dev).contiguous(memory_format=fmt).requires_grad_()
        g = torch.randn(B, h, hw, hw, device=dev).contiguous(memory_format=fmt)
        def step():
            net.zero_grad(set_to_none=True); x.grad = None
            net(x).backward(g)
        res.append(t(step))
    print(f"half {h:4d} {hw}x{hw} x{nblk}: NCHW {res[0]:8.1f} us   channels_last {res[1]:8.1f} us")
