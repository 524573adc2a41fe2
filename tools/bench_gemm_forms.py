#!/usr/bin/env python3
"""Which torch formulation of SS2D's skinny fp32 GEMMs does rocBLAS/hipBLASLt run fastest? (stage shapes of MedMamba-S)"""
import sys, torch
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
B = 64
for dim, hw in [(96, 56), (192, 28), (384, 14), (768, 7)]:
    dm, D, L = dim // 2, dim, hw * hw
    R, N = -(-dm // 16), 16
    C = R + 2 * N
    x = torch.randn(B, L, dm, device=dev); W = torch.randn(2 * D, dm, device=dev)
    Xt = x.transpose(1, 2)
    print(f"--- dim {dim} L {L}: in_proj (B,L,{dm}) x ({2*D},{dm})")
    print("  F.linear NHWC          %8.1f us" % t(lambda: torch.nn.functional.linear(x, W)))
    print("  matmul(W[:D], Xt) x2   %8.1f us" % t(lambda: (torch.matmul(W[:D], Xt), torch.matmul(W[D:], Xt))))
    print("  matmul(W, Xt) full     %8.1f us" % t(lambda: torch.matmul(W, Xt)))
    We = W.unsqueeze(0).expand(B, -1, -1)
    print("  bmm(W.expand, Xt)      %8.1f us" % t(lambda: torch.bmm(We, Xt)))
    Xc = Xt.contiguous()
    print("  bmm(W.expand, Xc cont) %8.1f us" % t(lambda: torch.bmm(We, Xc)))
    print("  linear + transpose copy%8.1f us" % t(lambda: torch.nn.functional.linear(x, W).transpose(1, 2).contiguous()))
    y = torch.randn(B, D, L, device=dev); Wo = torch.randn(dm, D, device=dev)
    print(f"  out_proj from y_cf (B,{D},L):")
    print("  matmul(y^T, Wo^T)      %8.1f us" % t(lambda: torch.matmul(y.transpose(1, 2), Wo.t())))
    Woe = Wo.unsqueeze(0).expand(B, -1, -1)
    print("  bmm(Wo.expand, y)^T    %8.1f us" % t(lambda: torch.bmm(Woe, y)))
    print("  bmm(y^T, Wo^T.expand)  %8.1f us" % t(lambda: torch.bmm(y.transpose(1, 2), Wo.t().unsqueeze(0).expand(B, -1, -1))))
    u2 = torch.randn(B, 2, D, L, device=dev); Wx = torch.randn(1, 2, 2 * C, D, device=dev)
    print(f"  x_dbl: (1,2,{2*C},{D}) @ (B,2,{D},L): %8.1f us" % t(lambda: torch.matmul(Wx, u2)))
    print("  x_dbl via bmm:                        %8.1f us" % t(lambda: torch.bmm(Wx.expand(B, -1, -1, -1).reshape(B * 2, 2 * C, D), u2.view(B * 2, D, L))))
    xd = torch.randn(B, 4, C, L, device=dev); Wdt = torch.randn(1, 4, D, R, device=dev)
    print(f"  dts: (1,4,{D},{R}) @ (B,4,{R},L):      %8.1f us" % t(lambda: torch.matmul(Wdt, xd[:, :, :R])))
    xr = xd[:, :, :R].contiguous()
    print("  dts via bmm contiguous rows:          %8.1f us" % t(lambda: torch.bmm(Wdt.expand(B, -1, -1, -1).reshape(B * 4, D, R), xr.view(B * 4, R, L))))
