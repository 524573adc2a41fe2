#!/usr/bin/env python3
"""Diagnostic: the SS2D projection GEMMs per block in (a) batch-major planes (B, D, L) = batched small GEMMs + batch
sums, vs (b) channel-major planes (D, B*L) = few large GEMMs with K or N = B*L.  Prints per-stage totals (us)."""
import sys, torch
dev = torch.device("cuda:0")
B = 64
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for dm, L in [(48, 3136), (96, 784), (192, 196), (384, 49)]:
    D = 2 * dm; R = max(1, (dm + 15) // 16); N = 16; C = R + 2 * N; Q = B * L
    r = lambda *s: torch.randn(*s, device=dev)
    # ---- (a) batch-major
    X = r(B, L, dm); Win = r(2 * D, dm); u2 = r(B, 2, D, L); Wx2 = r(2, 2 * C, D); Wdt = r(4, D, R); Wout = r(dm, D)
    xdbl = r(B, 4, C, L); dd = r(B, 4, D, L); y = r(B, D, L); dout = r(B, dm, L); dxz = r(B, 2 * D, L); dxd2 = r(B, 2, 2 * C, L)
    du2 = r(B * 2, D, L)
    a = {}
    a["in_proj"] = t(lambda: torch.bmm(Win.unsqueeze(0).expand(B, -1, -1), X.transpose(1, 2)))
    a["x_proj"] = t(lambda: torch.matmul(Wx2.unsqueeze(0), u2))
    a["dt"] = t(lambda: torch.matmul(Wdt.unsqueeze(0), xdbl[:, :, :R]))
    a["out_proj"] = t(lambda: torch.bmm(Wout.unsqueeze(0).expand(B, -1, -1), y))
    a["d_out_proj dy"] = t(lambda: torch.bmm(Wout.t().unsqueeze(0).expand(B, -1, -1), dout))
    a["d_out_proj dW"] = t(lambda: torch.bmm(dout, y.transpose(1, 2)).sum(0))
    a["dWdt"] = t(lambda: torch.matmul(dd, xdbl[:, :, :R].transpose(-1, -2)).sum(0))
    a["d dt_low"] = t(lambda: torch.matmul(Wdt.transpose(-1, -2).unsqueeze(0), dd))
    a["du2 += WxT dxdbl"] = t(lambda: du2.baddbmm_(Wx2.transpose(1, 2).unsqueeze(0).expand(B, -1, -1, -1).reshape(B * 2, D, 2 * C), dxd2.reshape(B * 2, 2 * C, L)))
    a["dWx"] = t(lambda: torch.matmul(dxd2, u2.transpose(-1, -2)).sum(0))
    a["d_in_proj dX"] = t(lambda: torch.bmm(dxz.transpose(1, 2), Win.unsqueeze(0).expand(B, -1, -1)))
    a["d_in_proj dW"] = t(lambda: torch.bmm(dxz, X).sum(0))
    # ---- (b) channel-major
    Xr = X.reshape(Q, dm); u2c = r(2, D, Q); xdblc = r(4, C, Q); ddc = r(4, D, Q); yc = r(D, Q); doutc = r(dm, Q); dxzc = r(2 * D, Q)
    dxd2c = xdblc.view(2, 2 * C, Q); du2c = r(2, D, Q)
    b = {}
    b["in_proj"] = t(lambda: torch.mm(Win, Xr.t()))
    b["x_proj"] = t(lambda: torch.bmm(Wx2, u2c))
    b["dt"] = t(lambda: torch.bmm(Wdt, xdblc[:, :R]))
    b["out_proj"] = t(lambda: torch.mm(Wout, yc))
    b["d_out_proj dy"] = t(lambda: torch.mm(Wout.t(), doutc))
    b["d_out_proj dW"] = t(lambda: torch.mm(doutc, yc.t()))
    b["dWdt"] = t(lambda: torch.bmm(ddc, xdblc[:, :R].transpose(1, 2)))
    b["d dt_low"] = t(lambda: torch.bmm(Wdt.transpose(1, 2), ddc))
    b["du2 += WxT dxdbl"] = t(lambda: du2c.baddbmm_(Wx2.transpose(1, 2), dxd2c))
    b["dWx"] = t(lambda: torch.bmm(dxd2c, u2c.transpose(1, 2)))
    b["d_in_proj dX"] = t(lambda: torch.mm(dxzc.t(), Win))
    b["d_in_proj dW"] = t(lambda: torch.mm(dxzc, Xr))
    print(f"== d_model {dm} L {L}: batch-major total {sum(a.values()):.0f} us, channel-major total {sum(b.values()):.0f} us")
    for k in a:
        print(f"   {k:<20} {a[k]:8.1f} {b[k]:8.1f}")
