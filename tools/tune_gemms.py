#!/usr/bin/env python3
"""Regenerate medmamba_amd/tuning/gemm_gfx950.csv: run MedMamba-S training steps (64 x 224^2, the bench workload) with
PyTorch TunableOp timing every rocBLAS / hipBLASLt solution for each GEMM shape it meets.  Run on ONE MI355X.
usage: python tools/tune_gemms.py [out.csv] [size=S] [batch=64] [res=224] [fresh] [det]   (without `fresh`, new shapes are added
to an existing table — e.g. `tools/tune_gemms.py medmamba_amd/tuning/gemm_gfx950.csv B 32 384`)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
from medmamba_amd.tuning import DEFAULT_FILE, enable_tuned_gemms

out = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_FILE
size = sys.argv[2] if len(sys.argv) > 2 else "S"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
res = int(sys.argv[4]) if len(sys.argv) > 4 else 224
if "fresh" in sys.argv[5:] and os.path.exists(out):
    os.remove(out)
if "det" in sys.argv[5:]:        # the shapes of the deterministic weight-gradient path (cudnn.deterministic, the reference's mode)
    torch.backends.cudnn.deterministic = True
torch.cuda.tunable.set_max_tuning_duration(60)      # ms per candidate
torch.cuda.tunable.set_max_tuning_iterations(50)
enable_tuned_gemms(out, tune=True)
dev = torch.device("cuda:0")
torch.manual_seed(42)
net = VSSM(num_classes=6, **MEDMAMBA_CONFIGS[size]).to(dev).train()
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
x = torch.randn(batch, 3, res, res, device=dev); y = torch.randint(0, 6, (batch,), device=dev)
for i in range(3):
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(net(x), y); loss.backward(); opt.step()
    torch.cuda.synchronize()
    print(f"step {i}: loss {float(loss.detach()):.4f}, {len(torch.cuda.tunable.get_results())} tuned GEMM shapes", flush=True)
with torch.no_grad():
    net.eval()(x)                                      # inference-mode shapes are the same GEMMs
torch.cuda.synchronize()
torch.cuda.tunable.write_file(out) if hasattr(torch.cuda.tunable, "write_file") else None
print("written", out)
