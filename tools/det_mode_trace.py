#!/usr/bin/env python3
"""Run under `rocprofv3 --kernel-trace`: MedMamba-S training steps (64 x 224^2) with torch.backends.cudnn.deterministic = True, the
reference's mode (train.py:28-29) — what the deterministic weight-gradient path (im2col + batched GEMM + ordered sum) costs per kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
from medmamba_amd.tuning import enable_tuned_gemms
from medmamba_amd.optim import FusedAdamW
enable_tuned_gemms()
torch.backends.cudnn.deterministic = True
dev = torch.device("cuda:0"); torch.manual_seed(42)
net = VSSM(num_classes=6, **MEDMAMBA_CONFIGS["S"]).to(dev).train()
opt = FusedAdamW(net.parameters(), lr=1e-4, weight_decay=1e-4)
x = torch.randn(64, 3, 224, 224, device=dev); y = torch.randint(0, 6, (64,), device=dev)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(net(x), y); loss.backward(); opt.step()
torch.cuda.synchronize()
