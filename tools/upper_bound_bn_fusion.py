#!/usr/bin/env python3
"""Diagnostic (timing only, results are NOT the model's): step time with the conv branch's ReLUs, conv bias adds and
conv bias gradients removed — the most a fused BatchNorm(+bias)+ReLU kernel pair could save."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
from medmamba_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
dev = torch.device("cuda:0")
def run(strip):
    torch.manual_seed(42)
    net = VSSM(num_classes=6, **MEDMAMBA_CONFIGS["S"]).to(dev).train()
    if strip:
        for layer in net.layers:
            for blk in layer.blocks:
                seq = blk.conv33conv33conv11
                for i, m in enumerate(seq):
                    if isinstance(m, torch.nn.ReLU) and i != len(seq) - 1:
                        seq[i] = torch.nn.Identity()
                    if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3):
                        m.bias = None
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
    x = torch.randn(64, 3, 224, 224, device=dev); y = torch.randint(0, 6, (64,), device=dev)
    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(net(x), y); loss.backward(); opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3
for strip in (False, True, False, True):
    print(f"strip={strip}: {run(strip):.2f} ms/step", flush=True)
