#!/usr/bin/env python3
"""Diagnostic: the weight-gradient GEMMs of a 14x14 block (channel-major planes: K = B*L = 12544) as ONE GEMM (what runs now, with
the library's own split-K) against G strided-batched GEMMs over K-slices + an ordered sum of the G partial products
(deterministic split-K by hand).  Kernel time from device events over 50 launches; torch.bmm / torch.mm with the default heuristics
and (TUNE=1) with TunableOp's search switched on for the new shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd import ops
dev = torch.device("cuda:0")
if os.environ.get("TUNE") == "1":
    torch.cuda.tunable.enable(True); torch.cuda.tunable.tuning_enable(True); torch.cuda.tunable.set_max_tuning_duration(30)
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
Q = 64 * 196
for name, nb, M, N in [("d_out_w", 1, 192, 384), ("dWx", 2, 88, 384), ("d_in_w", 1, 768, 192), ("dWdt", 4, 384, 12)]:
    A = torch.randn(nb, M, Q, device=dev); Bm = torch.randn(nb, N, Q, device=dev)        # both K-contiguous (rows of planes)
    ref = torch.bmm(A, Bm.transpose(1, 2))
    line = f"{name:<8} ({nb} x {M} x {N}, K = {Q}): one GEMM {t(lambda: torch.bmm(A, Bm.transpose(1, 2))):6.1f} us"
    for G in (2, 4, 8, 16, 32):
        kg = Q // G
        Ag = A.view(nb, M, G, kg).permute(0, 2, 1, 3).reshape(nb * G, M, kg) if False else None
        # strided views, no copies: (nb, G, M, kg) with strides (M*Q, kg, Q, 1)
        Av = A.as_strided((nb * G, M, kg), (kg, Q, 1)) if nb == 1 else None
        def run():
            if nb == 1:
                p = torch.bmm(A.as_strided((G, M, kg), (kg, Q, 1)), Bm.as_strided((G, N, kg), (kg, Q, 1)).transpose(1, 2))
                return ops.sum_lead(p)
            outs = []
            p = torch.matmul(A.as_strided((nb, G, M, kg), (M * Q, kg, Q, 1)), Bm.as_strided((nb, G, N, kg), (N * Q, kg, Q, 1)).transpose(2, 3))
            return ops.sum_lead(p.transpose(0, 1).contiguous()) if False else p.sum(1)
        out = run()
        err = float((out.reshape(ref.shape) - ref).abs().max() / ref.abs().max())
        line += f" | G={G}: {t(run):6.1f} us (err {err:.0e})"
    print(line, flush=True)
