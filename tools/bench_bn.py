#!/usr/bin/env python3
"""Diagnostic: training-mode BatchNorm2d + ReLU, forward + backward, own kernels (csrc/bn.hip) vs MIOpen / ATen, per stage shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.ops import bn_relu_train
dev = torch.device("cuda:0")
def t(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for B, C, hw in [(64, 48, 56), (64, 96, 28), (64, 192, 14), (64, 384, 7), (32, 64, 96)]:
    x = torch.randn(B, C, hw, hw, device=dev, requires_grad=True); dy = torch.randn_like(x)
    bn = torch.nn.BatchNorm2d(C).to(dev).train()
    def own_f():
        return bn_relu_train(x, bn, True)
    def ref_f():
        return torch.relu(bn(x))
    for name, f in (("own", own_f), ("torch", ref_f)):
        y = f()
        tf = t(lambda: f())
        tb = t(lambda: torch.autograd.grad(y, (x, bn.weight, bn.bias), dy, retain_graph=True))
        print(f"{B}x{C}x{hw}x{hw} {name:5s}: fwd {tf:7.1f} us  bwd {tb:7.1f} us   ({x.numel()*4/1e6:.1f} MB)")
