#!/usr/bin/env python3
"""Diagnostic: mm_block_split_fwd / _bwd alone (direct C-ABI calls) per MedMamba-S stage, B = 64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.lib(); B = 64
def t(fn, it=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for C, hw in [(96, 56), (192, 28), (384, 14), (768, 7)]:
    P, C2 = hw * hw, C // 2
    r = lambda *s: torch.randn(*s, device=dev)
    inp, g, b_ = r(B, P, C), r(C2), r(C2)
    left, rn, mu, rstd = r(B, C2, P), r(B, P, C2), r(B * P), r(B * P)
    dleft, drn, dres, dinp = r(B, C2, P), r(B, P, C2), r(B, P, C), torch.empty(B, P, C, device=dev)
    ws = torch.empty(lib.mm_block_split_rows(B, P, C2), 2 * C2, device=dev)
    st = _lib.raw_stream()
    f = lambda: lib.mm_block_split_fwd(inp.data_ptr(), g.data_ptr(), b_.data_ptr(), 1e-5, None, left.data_ptr(), rn.data_ptr(), mu.data_ptr(),
                                       rstd.data_ptr(), B, P, C2, st)
    bk = lambda: lib.mm_block_split_bwd(dleft.data_ptr(), drn.data_ptr(), dres.data_ptr(), inp.data_ptr(), g.data_ptr(), mu.data_ptr(),
                                        rstd.data_ptr(), dinp.data_ptr(), ws.data_ptr(), B, P, C2, st)
    unit = B * P * C * 4 / 1e6
    tf, tb = t(f), t(bk)
    print(f"C={C:4d} P={P:5d} fwd {tf:7.1f} us ({2 * unit / tf * 1e3:6.0f} GB/s)   bwd {tb:7.1f} us ({3.5 * unit / tb * 1e3:6.0f} GB/s)")
