#!/usr/bin/env python3
"""Which branch of a block does the other one wait for — measured WITHOUT a profiler (device events around the forward / backward of
the SS2D branch on the main stream and of the conv branch on the side stream, ops.BRANCH_TIMER).  Prints, per block of the last of
a few training steps, when each branch started and ended relative to the step's first event, and the sum of the times one branch
finished after the other.  usage: python tools/branch_balance.py [S|B] [batch] [res]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd import ops
from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
from medmamba_amd.optim import FusedAdamW
from medmamba_amd.tuning import enable_tuned_gemms

size = sys.argv[1] if len(sys.argv) > 1 else "S"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
res = int(sys.argv[3]) if len(sys.argv) > 3 else 224
dev = torch.device("cuda:0")
enable_tuned_gemms()
torch.manual_seed(42)
net = VSSM(num_classes=6, **MEDMAMBA_CONFIGS[size]).to(dev).train()
opt = FusedAdamW(net.parameters(), lr=1e-4, weight_decay=1e-4)
x = torch.randn(batch, 3, res, res, device=dev); y = torch.randint(0, 6, (batch,), device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(net(x), y); loss.backward(); opt.step()
    return loss.detach()


for _ in range(6):
    step()
torch.cuda.synchronize()
ops.BRANCH_TIMER.enabled = True
origin = torch.cuda.Event(enable_timing=True); origin.record()
step()
end = torch.cuda.Event(enable_timing=True); end.record()
torch.cuda.synchronize()
ops.BRANCH_TIMER.enabled = False
recs = [(tag, origin.elapsed_time(s) * 1e3, origin.elapsed_time(e) * 1e3) for tag, s, e in ops.BRANCH_TIMER.records]
print(f"# MedMamba-{size} {batch} x {res}^2, one training step with branch events: {origin.elapsed_time(end):.2f} ms (times below in us from the step's start)")
for phase in ("fwd", "bwd"):
    ss = sorted([r for r in recs if r[0] == "ss2d_" + phase], key=lambda r: r[1])
    cv = sorted([r for r in recs if r[0] == "conv_" + phase], key=lambda r: r[1])
    print(f"\n## {phase}: block, SS2D branch [start, end] (main stream), conv branch [start, end] (side stream), who ends later by how much")
    wait_main = wait_side = 0.0
    for i, (a, b) in enumerate(zip(ss, cv)):
        d = b[2] - a[2]
        if d > 0: wait_main += d
        else: wait_side += -d
        print(f"{i:>3}  ss2d [{a[1]:>8.0f} {a[2]:>8.0f}] {a[2]-a[1]:>6.0f}   conv [{b[1]:>8.0f} {b[2]:>8.0f}] {b[2]-b[1]:>6.0f}   {'conv' if d > 0 else 'ss2d'} later by {abs(d):>5.0f}")
    print(f"   the main stream waits for the conv branch {wait_main:.0f} us in all; the conv branch is done earlier by {wait_side:.0f} us in all")
