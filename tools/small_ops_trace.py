#!/usr/bin/env python3
"""Which Python line issues every small ATen op of one SS_Conv_SSM block (fwd + bwd): a TorchDispatchMode that logs the op and
the innermost medmamba_amd frame.  usage: tools/small_ops_trace.py STAGE [BATCH]"""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.utils._python_dispatch import TorchDispatchMode
from medmamba_amd.modules import SS_Conv_SSM
from medmamba_amd.tuning import enable_tuned_gemms
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dim, hw = [(96, 56), (192, 28), (384, 14), (768, 7)][stage]
enable_tuned_gemms()
dev = torch.device("cuda:0")
blk = SS_Conv_SSM(hidden_dim=dim, drop_path=0.0, norm_layer=torch.nn.LayerNorm).to(dev).train()
x = torch.randn(B, hw, hw, dim, device=dev, requires_grad=True)
for _ in range(2):
    blk.zero_grad(set_to_none=True); blk(x).sum().backward()
SKIP = ("aten.view", "aten.permute", "aten.transpose", "aten.expand", "aten.slice", "aten.select", "aten.as_strided", "aten.empty",
        "aten.unsqueeze", "aten.squeeze", "aten.reshape", "aten._unsafe_view", "aten.t.", "aten.detach", "aten.alias", "aten.split",
        "aten.chunk", "aten.unbind", "aten.stride", "aten.sym", "aten.is_", "aten._local_scalar", "aten.lift", "aten.new_empty",
        "aten.empty_like", "aten.set_", "aten.record_stream", "aten.numel", "aten.size", "aten.dim")
log = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            fr = [f for f in traceback.extract_stack() if "medmamba_amd" in f.filename and "small_ops" not in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "(autograd)"
            log[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Log():
    blk.zero_grad(set_to_none=True)
    y = blk(x); y.backward(torch.ones_like(y)); torch.cuda.synchronize()
for (name, where), n in sorted(log.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"{n:3d}  {name:<42} {where}")
