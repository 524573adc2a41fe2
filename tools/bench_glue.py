#!/usr/bin/env python3
"""Diagnostic: every glue op alone per MedMamba-S stage (B = 64): time, bytes moved, achieved GB/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd import ops
dev = torch.device("cuda:0")
B = 64
def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for C, hw in [(96, 56), (192, 28), (384, 14), (768, 7)]:
    L = hw * hw; D = C; dm = C // 2; unit = B * D * L * 4 / 1e6       # MB of one (B, D, L) plane set
    r = lambda *s: torch.randn(*s, device=dev)
    cm = ops.channel_major(B, L)
    pl = lambda d: ops._planes(B, d, L, dev, cm).normal_()
    rows = []
    inp = r(B, hw, hw, C).requires_grad_(); g, b_ = r(dm), r(dm)
    left, rn, res = ops.block_split(inp, g, b_, 1e-5)
    rows.append(("block_split fwd", t(lambda: ops.block_split(inp, g, b_, 1e-5)), 1.0 * unit))          # read C, write C/2 + C/2
    gl, gr, gres = torch.randn_like(left), torch.randn_like(rn), torch.randn_like(res)
    rows.append(("block_split bwd", t(lambda: torch.autograd.grad([left, rn, res], inp, [gl, gr, gres], retain_graph=True)), 2.5 * unit))
    x = pl(D).requires_grad_(); w, bb = r(D, 1, 3, 3).requires_grad_(), r(D).requires_grad_()
    u2 = ops.dwconv_silu_cross(x, w, bb, hw, hw)
    rows.append(("dwconv fwd", t(lambda: ops.dwconv_silu_cross(x, w, bb, hw, hw)), 3 * unit))
    gu = pl(2 * D)
    rows.append(("dwconv bwd", t(lambda: torch.autograd.grad(u2, x, gu, retain_graph=True)), 4 * unit))
    lft = r(B, dm, hw, hw); ssm = pl(dm); xin = r(B, hw, hw, C)
    rows.append(("shuffle_residual fwd", t(lambda: ops.shuffle_residual(lft, ssm, xin, True)), 2 * unit))
    from medmamba_amd import _lib
    lib, st, P = _lib.lib(), _lib.raw_stream(), ops._pl
    out4 = torch.randn(B, 4 * D, L, device=dev); mm_ = pl(D)
    rows.append(("cross_merge", t(lambda: lib.mm_cross_merge_fwd(out4.data_ptr(), *P(mm_), B, D, hw, hw, st)), 5 * unit))
    d2 = pl(2 * D)
    rows.append(("plane_transpose", t(lambda: lib.mm_plane_transpose(*P(d2), *P(d2[:, D:]), B, D, hw, hw, st)), 2 * unit))
    du2, du4, xcf, dxc = pl(2 * D), pl(4 * D), pl(D), pl(D)
    wsc = torch.empty((B, D * lib.mm_dwconv_silu_cross_strips(hw, hw), 10), device=dev)
    wq, bq = r(D, 1, 3, 3), r(D)
    rows.append(("dwconv bwd (fused du4)", t(lambda: lib.mm_dwconv_silu_cross_bwd(*P(du2), *P(du4), *P(xcf), wq.data_ptr(), bq.data_ptr(), *P(dxc),
                                                                                   wsc.data_ptr(), B, D, hw, hw, st)), 8 * unit))
    for name, us, mb in rows:
        print(f"C={C:4d} L={L:5d} {name:<22} {us:8.1f} us  {mb:8.1f} MB  {mb / us * 1e3:8.0f} GB/s")
