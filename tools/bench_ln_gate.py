#!/usr/bin/env python3
"""Diagnostic: mm_ln_gate_fwd / _bwd alone per MedMamba-S stage (B = 64, the layout ops.channel_major picks)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd import _lib, ops
dev = torch.device("cuda:0"); lib = _lib.lib(); B = 64
def t(fn, it=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for D, L in [(96, 3136), (192, 784), (384, 196), (768, 49)]:
    cm = ops.channel_major(B, L)
    pl = lambda d: ops._planes(B, d, L, dev, cm).normal_()
    m, z, y, dy, dz = pl(D), pl(D), pl(D), pl(D), pl(D); dm = pl(2 * D)
    g, b_ = torch.randn(D, device=dev), torch.randn(D, device=dev)
    mu, rstd = torch.empty(B, L, device=dev), torch.empty(B, L, device=dev)
    ws = torch.empty(lib.mm_ln_gate_rows(B, D, L), 2 * D, device=dev)
    st = _lib.raw_stream(); P = ops._pl
    f = lambda: lib.mm_ln_gate_fwd(*P(m), *P(z), g.data_ptr(), b_.data_ptr(), 1e-5, *P(y), mu.data_ptr(), rstd.data_ptr(), B, D, L, st)
    bk = lambda: lib.mm_ln_gate_bwd(*P(dy), *P(m), *P(z), g.data_ptr(), b_.data_ptr(), mu.data_ptr(), rstd.data_ptr(), *P(dm), *P(dz), ws.data_ptr(), B, D, L, st)
    unit = B * D * L * 4 / 1e6
    tf, tb = t(f), t(bk)
    print(f"D={D:4d} L={L:5d} fwd {tf:7.1f} us ({3 * unit / tf * 1e3:6.0f} GB/s)   bwd {tb:7.1f} us ({5 * unit / tb * 1e3:6.0f} GB/s)   ws rows {ws.shape[0]}")
