#!/usr/bin/env python3
"""Diagnostic: time of mm_ss2d_pack_fwd / _bwd alone, per MedMamba-S stage."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.lib()
HW = {96: 56, 192: 28, 384: 14, 768: 7}
for D, R in [(96, 3), (192, 6), (384, 12), (768, 24)]:
    N, C = 16, R + 32
    n = lib.mm_ss2d_pack_size(D, C, R, N)
    srcs = [torch.randn(4, C, D, device=dev), torch.randn(4, D, R, device=dev), torch.randn(4, D, device=dev),
            torch.randn(4 * D, N, device=dev), torch.randn(4 * D, device=dev)]
    P = torch.empty(n, device=dev); G = torch.empty(n, device=dev); dP = torch.randn(n, device=dev)
    st = _lib.raw_stream()
    def f(): lib.mm_ss2d_pack_fwd(*[t.data_ptr() for t in srcs], P.data_ptr(), D, C, R, N, st)
    Bz = 64; parts = torch.randn(Bz, lib.mm_ss2d_pack_parts_size(D, C, R, N), device=dev)      # per-batch-item dA | dD | dbias partials
    hw = HW[D]; ln_rows = lib.mm_ln_gate_rows(Bz, D, hw * hw); strips = lib.mm_dwconv_silu_cross_strips(hw, hw)
    ln_ws = torch.randn(ln_rows, 2 * D, device=dev); dw_ws = torch.randn(Bz, D * strips, 10, device=dev)
    G = torch.empty(n + 12 * D, device=dev)              # packed gradients | dgamma, dbeta of out_norm | depthwise conv dW, db
    def b(): lib.mm_ss2d_pack_bwd(dP.data_ptr(), P.data_ptr(), parts.data_ptr(), G.data_ptr(), D, C, R, N, Bz, ln_ws.data_ptr(), ln_rows,
                                  G[n:].data_ptr(), dw_ws.data_ptr(), Bz, strips, G[n + 2 * D:].data_ptr(), st)
    for fn, name in ((f, "fwd"), (b, "bwd")):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"D={D} n={n} {name}: {e0.elapsed_time(e1) / 200 * 1e3:.2f} us per launch (back to back)")
    def c(): G[:n].copy_(dP)
    for _ in range(5): c()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): c()
    e1.record(); torch.cuda.synchronize()
    print(f"D={D} n={n} torch copy_: {e0.elapsed_time(e1) / 200 * 1e3:.2f} us per launch")
