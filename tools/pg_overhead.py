#!/usr/bin/env python3
"""Diagnostic: does an initialised RCCL process group by itself slow the (host-bound-ish) training step?"""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
from medmamba_amd.tuning import enable_tuned_gemms
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
if mode != "none":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
    if mode == "nccl_eager":
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    else:
        dist.init_process_group(backend=mode, rank=0, world_size=1)
if mode != "none" and os.environ.get("PG_WARM", "0") == "1":
    t = torch.ones(4, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
enable_tuned_gemms()
torch.manual_seed(42)
net = VSSM(num_classes=6, **MEDMAMBA_CONFIGS["S"]).to(dev).train()
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
x = torch.randn(64, 3, 224, 224, device=dev); y = torch.randint(0, 6, (64,), device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(net(x), y); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"pg={mode}: {(t2 - t0) / 20 * 1e3:.2f} ms/step (host enqueue {(t1 - t0) / 20 * 1e3:.2f} ms)", flush=True)
if mode != "none": dist.destroy_process_group()
