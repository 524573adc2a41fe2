#!/usr/bin/env python3
"""Forward scan at the 7x7 stage of S (D = 768, L = 49): launch-plan variants and batch scaling (is the call bound by its tail?)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.selective_scan_interface import SelectiveScanFn
from tools.bench_scan import timeit, bytes_fwd
dev = torch.device("cuda:0"); K, N, D, L = 4, 16, 768, 49
for Bz in (16, 32, 64, 128, 256):
    g = torch.Generator(device=dev).manual_seed(0)
    R = 24
    u = torch.randn(Bz, K * D, L, device=dev, generator=g); delta = torch.randn(Bz, K * D, L, device=dev, generator=g)
    A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(K * D, 1)
    x_dbl = torch.randn(Bz, K, R + 2 * N, L, device=dev, generator=g)
    Bs, Cs = x_dbl[:, :, R:R + N], x_dbl[:, :, R + N:]
    Dp = torch.ones(K * D, device=dev); bias = torch.randn(K * D, device=dev, generator=g) - 4.0
    for name, v in (("plan", 0), ("general(bit25)", 1 << 25), ("per-wave BC(bit26)", 1 << 26), ("ns2", 2), ("ns1", 1)):
        if Bz != 64 and v: continue
        t, _ = timeit(lambda: SelectiveScanFn.apply(u, delta, A, Bs, Cs, Dp, bias, True, v))
        print(f"batch {Bz:4d} {name:20s}: {t * 1e3:7.1f} us  {bytes_fwd(Bz, K, D, N, L) / t / 1e6:7.0f} GB/s", flush=True)
