#!/usr/bin/env python3
"""Host-side cost of one training step per operator (torch.profiler, CPU activity only), forward and backward threads alike.
Run at a small batch (the GPU then waits for the host and the step time IS the host time).  usage: tools/host_op_profile.py [BATCH]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
from medmamba_amd.tuning import enable_tuned_gemms
if os.environ.get("MM_TUNED_GEMMS", "1") == "1":
    enable_tuned_gemms()
dev = torch.device("cuda:0")
torch.manual_seed(42)
net = VSSM(num_classes=6, **MEDMAMBA_CONFIGS["S"]).to(dev).train()
if os.environ.get("MM_FUSED_ADAMW", "1") == "1":
    from medmamba_amd.optim import FusedAdamW
    opt = FusedAdamW(net.parameters(), lr=1e-4, weight_decay=1e-4)
else:
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.randn(B, 3, 224, 224, device=dev); y = torch.randint(0, 6, (B,), device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(net(x), y); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print(f"step {1e2 * (time.perf_counter() - t0):.2f} ms (batch {B})")
# phases
def phase():
    torch.cuda.synchronize(); t = [time.perf_counter()]
    opt.zero_grad(set_to_none=True); t.append(time.perf_counter())
    loss = torch.nn.functional.cross_entropy(net(x), y); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    return [1e3 * (b - a) for a, b in zip(t, t[1:])]
ph = [phase() for _ in range(10)]
print("host ms: zero_grad %.2f  forward %.2f  backward %.2f  optimizer %.2f" % tuple(sorted(c)[len(c) // 2] for c in zip(*ph)))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(5): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=60, max_name_column_width=60))
