#!/usr/bin/env python3
"""medmamba_amd/tuning/gemm_gfx950_rocblas.csv: rocBLAS-ONLY winners (TunableOp with hipBLASLt switched off) for the weight-gradient
GEMMs of the channel-major SS2D blocks whose overall winner is a hipBLASLt kernel — the kernels the parameter half must not use
when it runs on a third stream.  Operands are built with the strides ss2d_bwd_params hands to gemm_out.
usage: PYTORCH_TUNABLEOP_HIPBLASLT_ENABLED=0 python tools/tune_param_gemms.py [out.csv]"""
import os, sys
os.environ.setdefault("PYTORCH_TUNABLEOP_HIPBLASLT_ENABLED", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "medmamba_amd", "tuning", "gemm_gfx950_rocblas.csv")
if os.path.exists(out):
    os.remove(out)
t = torch.cuda.tunable
t.enable(True); t.tuning_enable(True); t.set_max_tuning_duration(40); t.set_max_tuning_iterations(30)
t.set_filename(out, insert_device_ordinal=False)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
for (Bsz, L, dm, D, R) in ((64, 196, 192, 384, 12), (64, 49, 384, 768, 24)):      # the 14x14 and 7x7 stages of T / S at 64 images
    Q, C = Bsz * L, R + 32
    g_out, y2d = r(dm, Q), r(D, Q)
    torch.mm(g_out, y2d.t(), out=torch.empty(dm, D, device=dev))                               # d(out_proj.weight)
    dd, x_dbl = r(4, D, Q), r(4, C, Q)
    torch.bmm(dd, x_dbl.narrow(1, 0, R).transpose(1, 2), out=torch.empty(4, D, R, device=dev))  # d(dt_projs_weight)
    dx2, u2m = r(2, 2 * C, Q), r(2, D, Q)
    torch.bmm(dx2, u2m.transpose(1, 2), out=torch.empty(2, 2 * C, D, device=dev))              # d(x_proj_weight)
    g_in, x2 = r(2 * D, Q), r(Q, dm)
    torch.mm(g_in, x2, out=torch.empty(2 * D, dm, device=dev))                                 # d(in_proj.weight)
    torch.cuda.synchronize()
    print(f"L={L}: {len(t.get_results())} shapes tuned", flush=True)
t.write_file(out)
print(open(out).read())
