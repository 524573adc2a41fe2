#!/usr/bin/env python3
"""Diagnostic: host-side (Python) profile of the training step — where the launch overhead goes."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
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
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45); st.sort_stats("cumulative").print_stats(70)
