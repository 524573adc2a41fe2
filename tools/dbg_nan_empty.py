"""Every torch.empty / empty_like / new_empty the PYTHON route of this package allocates is filled with NaN: a kernel that reads memory
nobody wrote shows up as NaN in a gradient.  MM_HOST_CPP=0 python tools/dbg_nan_empty.py"""
import os, sys, torch
os.environ["MM_HOST_CPP"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_empty, _empty_like = torch.empty, torch.empty_like
def nan_empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_floating_point() and t.is_cuda: t.fill_(float("nan"))
    return t
def nan_empty_like(*a, **k):
    t = _empty_like(*a, **k)
    if t.is_floating_point() and t.is_cuda: t.fill_(float("nan"))
    return t
torch.empty, torch.empty_like = nan_empty, nan_empty_like
from medmamba_amd import modules, ops
from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
DEV = torch.device("cuda:0")
modules._TWO_STREAMS = False
torch.manual_seed(42)
net = VSSM(num_classes=6, drop_path_rate=0.0, **MEDMAMBA_CONFIGS["S"]).to(DEV).train()
g = torch.Generator().manual_seed(0)
x = _empty(64, 3, 224, 224).normal_(generator=g).to(DEV); y = torch.randint(0, 6, (64,), generator=g).to(DEV)
for it in range(2):
    net.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(net(x), y); loss.backward(); torch.cuda.synchronize()
    bad = [k for k, p in net.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    print(f"pass {it}: loss {float(loss.detach()):.6f}; parameters with non-finite gradients: {len(bad)} {bad[:8]}")
