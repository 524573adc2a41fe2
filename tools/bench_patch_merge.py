#!/usr/bin/env python3
"""Diagnostic: mm_patch_merge_ln_fwd / _bwd (the three PatchMerging2D of MedMamba-S) and mm_nchw_ln_rows_fwd / _bwd (PatchEmbed2D's
LayerNorm) alone at 64 images, direct C-ABI calls."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.lib(); B = 64; st = _lib.raw_stream()
def t(fn, it=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
r = lambda *s: torch.randn(*s, device=dev)
for C, hw in [(96, 56), (192, 28), (384, 14)]:
    x = r(B, hw, hw, C); g, b_ = r(4 * C), r(4 * C); n = B * (hw // 2) ** 2
    out, mu, rstd = torch.empty(n, 4 * C, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev)
    dy, dinp = r(n, 4 * C), torch.empty_like(x)
    ws = torch.empty(lib.mm_patch_merge_ln_rows(B, hw, hw), 8 * C, device=dev)
    f = lambda: lib.mm_patch_merge_ln_fwd(x.data_ptr(), g.data_ptr(), b_.data_ptr(), 1e-5, out.data_ptr(), mu.data_ptr(), rstd.data_ptr(), B, hw, hw, C, st)
    bk = lambda: lib.mm_patch_merge_ln_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr(), mu.data_ptr(), rstd.data_ptr(), dinp.data_ptr(), ws.data_ptr(), B, hw, hw, C, st)
    assert f() == 0 and bk() == 0
    mb = x.numel() * 4 / 1e6; tf, tb = t(f), t(bk)
    print(f"patch_merge C={C:4d} {hw}x{hw}: fwd {tf:7.1f} us ({2 * mb / tf * 1e3:5.0f} GB/s)  bwd {tb:7.1f} us ({3 * mb / tb * 1e3:5.0f} GB/s)  ws rows {ws.shape[0]}")
C, HW = 96, 56 * 56
x = r(B, C, HW); g, b_ = r(C), r(C); out = torch.empty(B, HW, C, device=dev); mu, rstd = torch.empty(B * HW, device=dev), torch.empty(B * HW, device=dev)
dy, dx = r(B, HW, C), torch.empty_like(x); ws = torch.empty(lib.mm_nchw_ln_rows_ws_rows(B, HW), 2 * C, device=dev)
f = lambda: lib.mm_nchw_ln_rows_fwd(x.data_ptr(), g.data_ptr(), b_.data_ptr(), 1e-5, out.data_ptr(), mu.data_ptr(), rstd.data_ptr(), B, C, HW, st)
bk = lambda: lib.mm_nchw_ln_rows_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr(), mu.data_ptr(), rstd.data_ptr(), dx.data_ptr(), ws.data_ptr(), B, C, HW, st)
assert f() == 0 and bk() == 0
mb = x.numel() * 4 / 1e6; tf, tb = t(f), t(bk)
print(f"nchw_ln_rows C={C} HW={HW}: fwd {tf:7.1f} us ({2 * mb / tf * 1e3:5.0f} GB/s)  bwd {tb:7.1f} us ({3 * mb / tb * 1e3:5.0f} GB/s)")
