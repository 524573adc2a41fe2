#!/usr/bin/env python3
"""Diagnostic: is the training step bound by the host's launch rate?  Times enqueue-only vs synchronised steps."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
from medmamba_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
dev = torch.device("cuda:0")
torch.manual_seed(42)
net = VSSM(num_classes=6, **MEDMAMBA_CONFIGS["S"]).to(dev).train()
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(B, 3, 224, 224, device=dev); y = torch.randint(0, 6, (B,), device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(net(x), y); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e2*(t1-t0):.2f} ms/step, complete {1e2*(t2-t0):.2f} ms/step")
# forward-only / backward-only split
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    with torch.no_grad(): net(x)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"no_grad forward: enqueue {1e2*(t1-t0):.2f} ms, complete {1e2*(t2-t0):.2f} ms")
