#!/usr/bin/env python3
"""Diagnostic: the two dense 3x3 convolutions of the conv branch alone (fwd + bwd through ConvBiasFn / autograd), NCHW vs
channels_last memory format, per MedMamba-S stage at 64 images: what MIOpen's layout choice costs."""
import torch
dev = torch.device("cuda:0")
B = 64
def t(fn, it=10):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
import sys
SHAPES = {"S": (64, [(48, 56), (96, 28), (192, 14), (384, 7)]), "B": (32, [(64, 96), (128, 48), (256, 24), (512, 12)])}
B, shapes = SHAPES[sys.argv[1] if len(sys.argv) > 1 else "S"]
for h, hw in shapes:
    res = []
    for fmt in (torch.contiguous_format, torch.channels_last):
        w = (torch.randn(h, h, 3, 3, device=dev) / (3 * h ** 0.5)).contiguous(memory_format=fmt).requires_grad_()
        b = torch.randn(h, device=dev).requires_grad_()
        x = torch.randn(B, h, hw, hw, device=dev).contiguous(memory_format=fmt).requires_grad_()
        g = torch.randn(B, h, hw, hw, device=dev).contiguous(memory_format=fmt)
        def fwd():
            return torch.nn.functional.conv2d(x, w, b, padding=1)
        def step():
            w.grad = None; x.grad = None; b.grad = None
            fwd().backward(g)
        res.append((t(fwd), t(step)))
    print(f"C {h:4d} {hw}x{hw}: NCHW fwd {res[0][0]:7.1f} fwd+bwd {res[0][1]:7.1f} us   channels_last fwd {res[1][0]:7.1f} fwd+bwd {res[1][1]:7.1f} us")
