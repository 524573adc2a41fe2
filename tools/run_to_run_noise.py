"""Diagnostic: gradients of N identical MedMamba-S training passes (64 x 224^2, single stream) against the second pass — how reproducible\nthe step is on this box.  DET=1: under torch.backends.cudnn.deterministic (bitwise); MM_HOST_CPP=0: the Python launch route."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd import modules, ops
from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
DEV = torch.device("cuda:0")
if os.environ.get("DET") == "1": torch.backends.cudnn.deterministic = True
junk = [torch.empty(1 << 26, device=DEV).normal_() for _ in range(8)]; del junk
torch.manual_seed(42)
net = VSSM(num_classes=6, drop_path_rate=0.0, **MEDMAMBA_CONFIGS["S"]).to(DEV).train()
g = torch.Generator().manual_seed(0)
x = torch.randn(64, 3, 224, 224, generator=g).to(DEV); y = torch.randint(0, 6, (64,), generator=g).to(DEV)
modules._TWO_STREAMS = False
def run():
    net.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(net(x), y); loss.backward(); torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in net.named_parameters()}
run(); ref = run()
rel = lambda a, b: float((a.double() - b.double()).norm() / max(1e-30, float(a.double().norm())))
for i in range(5):
    gr = run()
    w = sorted(((rel(ref[k], gr[k]), k) for k in gr if not k.endswith(("11.1.bias", "11.4.bias"))), reverse=True)[:3]
    print(f"pass {i}:", [(f"{v:.1e}", k.replace("self_attention", "sa").replace("conv33conv33conv11", "cb")) for v, k in w])
