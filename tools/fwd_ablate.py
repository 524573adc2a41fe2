#!/usr/bin/env python3
"""Diagnostic (experiments library: MM_HIP_LIB=medmamba_amd/lib/libmedmamba_hip_exp.so): forward scan per S stage, inference and
training form, with the timing ablations of scan_fwd.hip (variant bits 8-15: 1 = no y store, 2 = no recurrence)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.selective_scan_interface import SelectiveScanFn
from tools.bench_scan import STAGES, timeit
dev = torch.device("cuda:0"); K, N = 4, 16; Bz = 64
stages = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2]
for si, (D, L, nblk) in enumerate(STAGES["S"]):
    if si not in stages: continue
    R = max(1, (D // 2 + 15) // 16)
    g = torch.Generator(device=dev).manual_seed(0)
    u = torch.randn(Bz, K * D, L, device=dev, generator=g); delta = torch.randn(Bz, K * D, L, device=dev, generator=g)
    A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(K * D, 1)
    x_dbl = torch.randn(Bz, K, R + 2 * N, L, device=dev, generator=g)
    Bs, Cs = x_dbl[:, :, R:R + N], x_dbl[:, :, R + N:]
    Dp = torch.ones(K * D, device=dev); bias = torch.randn(K * D, device=dev, generator=g) - 4.0
    ins = [t.detach().requires_grad_() for t in (u, delta, A, Bs, Cs, Dp, bias)]
    for dbg in (0, 1, 2, 3):
        v = dbg << 8
        inf, _ = timeit(lambda: SelectiveScanFn.apply(u, delta, A, Bs, Cs, Dp, bias, True, v))
        trn, _ = timeit(lambda: SelectiveScanFn.apply(*ins, True, v))
        print(f"D={D} L={L} dbg={dbg}: inference {inf * 1e3:7.1f} us   training {trn * 1e3:7.1f} us", flush=True)
