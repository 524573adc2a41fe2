"""GPU: the fused glue kernels (csrc/glue.hip) against the reference's own op chains (oracle/model_ref.py),
bit-exact where the kernel only moves data / adds once."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(params=["bm", "cm"])
def layout(request, monkeypatch):
    """Run the plane ops in both storage layouts (batch-major / channel-major planes; ops.channel_major)."""
    from medmamba_amd import ops
    monkeypatch.setattr(ops, "_LAYOUT", request.param)
    return request.param


def _cm(t):
    """Same values as (B, D, L) tensor t, stored channel-major (D, B, L)."""
    return t.permute(1, 0, 2).contiguous().permute(1, 0, 2)


@pytest.mark.parametrize("shape", [(2, 8, 6, 5), (1, 48, 56, 56), (3, 33, 7, 9), (2, 64, 4, 4)])
def test_shuffle_residual_forward_backward(shape):
    from medmamba_amd.ops import shuffle_residual
    from oracle.model_ref import shuffle_residual_ref
    B, C2, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    left = torch.randn(B, C2, H, W, generator=g)
    ssm = torch.randn(B, H, W, C2, generator=g)
    inp = torch.randn(B, H, W, 2 * C2, generator=g)
    dout = torch.randn(B, H, W, 2 * C2, generator=g)
    ref_in = [t.clone().requires_grad_() for t in (left, ssm, inp)]
    ref = shuffle_residual_ref(*ref_in)
    ref.backward(dout)
    dev_in = [t.to(DEV).requires_grad_() for t in (left, ssm, inp)]
    out = shuffle_residual(*dev_in)
    out.backward(dout.to(DEV))
    assert torch.equal(out.detach().cpu(), ref.detach())            # one fp32 add per element: bit-exact
    for a, b in zip(dev_in, ref_in):
        assert torch.equal(a.grad.cpu(), b.grad)
    # channel-first SS2D branch (B, C/2, H*W): same values, transposed layout in and out
    ssm_cf = ssm.reshape(B, H * W, C2).transpose(1, 2).contiguous()
    dev_cf = [left.to(DEV).requires_grad_(), ssm_cf.to(DEV).requires_grad_(), inp.to(DEV).requires_grad_()]
    out_cf = shuffle_residual(*dev_cf, channel_first=True)
    out_cf.backward(dout.to(DEV))
    assert torch.equal(out_cf.detach().cpu(), ref.detach())
    assert torch.equal(dev_cf[1].grad.cpu(), ref_in[1].grad.reshape(B, H * W, C2).transpose(1, 2))
    assert torch.equal(dev_cf[0].grad.cpu(), ref_in[0].grad)
    # ... and the same planes stored channel-major (C/2, B, H*W)
    leaf = ssm_cf.to(DEV).requires_grad_()
    dev_cm = [left.to(DEV).requires_grad_(), leaf, inp.to(DEV).requires_grad_()]
    out_cm = shuffle_residual(dev_cm[0], _cm(leaf), dev_cm[2], channel_first=True)
    out_cm.backward(dout.to(DEV))
    assert torch.equal(out_cm.detach().cpu(), ref.detach())
    assert torch.equal(leaf.grad.cpu(), ref_in[1].grad.reshape(B, H * W, C2).transpose(1, 2))
    # folded neighbours: trailing ReLU of the conv branch + per-sample DropPath factor (MedMamba.py:347, 353)
    scale = (torch.rand(B, generator=g) > 0.4).float() / 0.6
    for cf in (False, True):
        ref_in = [t.clone().requires_grad_() for t in (left, ssm, inp)]
        ref = shuffle_residual_ref(*ref_in, ssm_scale=scale, left_relu=True)
        ref.backward(dout)
        dev_in = [left.to(DEV).requires_grad_(), (ssm_cf if cf else ssm).to(DEV).requires_grad_(), inp.to(DEV).requires_grad_()]
        out = shuffle_residual(*dev_in, channel_first=cf, ssm_scale=scale.to(DEV), left_relu=True)
        out.backward(dout.to(DEV))
        assert (out.detach().cpu() - ref.detach()).abs().max().item() <= 1e-6      # fma(s, scale, inp) vs mul + add
        gs = ref_in[1].grad.reshape(B, H * W, C2).transpose(1, 2) if cf else ref_in[1].grad
        assert torch.equal(dev_in[1].grad.cpu(), gs)
        assert torch.equal(dev_in[0].grad.cpu(), ref_in[0].grad)
        assert torch.equal(dev_in[2].grad.cpu(), ref_in[2].grad)
    # ... and the closing 1x1 conv's bias added in front of that ReLU (MedMamba.py:345): value, mask and the bias gradient
    bias = torch.randn(C2, generator=g)
    rb = bias.clone().requires_grad_()
    ref_in = [t.clone().requires_grad_() for t in (left, ssm, inp)]
    ref = shuffle_residual_ref(*ref_in, ssm_scale=scale, left_relu=True, left_bias=rb)
    ref.backward(dout)
    db = bias.to(DEV).requires_grad_()
    dev_in = [left.to(DEV).requires_grad_(), ssm_cf.to(DEV).requires_grad_(), inp.to(DEV).requires_grad_()]
    out = shuffle_residual(*dev_in, channel_first=True, ssm_scale=scale.to(DEV), left_relu=True, left_bias=db)
    out.backward(dout.to(DEV))
    assert (out.detach().cpu() - ref.detach()).abs().max().item() <= 1e-6
    assert torch.equal(dev_in[0].grad.cpu(), ref_in[0].grad)
    assert (db.grad.cpu() - rb.grad).abs().max().item() <= 1e-4 * max(1.0, rb.grad.abs().max().item())


def test_in_proj_cf_forward_backward(layout):
    from medmamba_amd.ops import in_proj_cf
    from oracle.model_ref import in_proj_cf_ref
    g = torch.Generator().manual_seed(3)
    B, L, dm, D = 3, 50, 24, 48
    x, w, bias = torch.randn(B, L, dm, generator=g), torch.randn(2 * D, dm, generator=g), torch.randn(2 * D, generator=g)
    gx, gz = torch.randn(B, D, L, generator=g), torch.randn(B, D, L, generator=g)
    for use_bias in (False, True):
        r = [t.clone().requires_grad_() for t in (x, w, bias)]
        a0, b0 = in_proj_cf_ref(r[0], r[1], r[2] if use_bias else None)
        torch.autograd.backward([a0, b0], [gx, gz])
        d = [t.to(DEV).requires_grad_() for t in (x, w, bias)]
        a1, b1 = in_proj_cf(d[0], d[1], d[2] if use_bias else None)
        assert a1.shape == (B, D, L) and b1.shape == (B, D, L)
        gg = [gx.to(DEV), gz.to(DEV)]
        torch.autograd.backward([a1, b1], [_cm(t) for t in gg] if layout == "cm" else gg)
        close = lambda p, q: (p.detach().cpu() - q.detach()).abs().max().item() <= 2e-5 * max(1.0, q.detach().abs().max().item())
        assert close(a1, a0) and close(b1, b0)
        for p, q in list(zip(d, r))[: 3 if use_bias else 2]:
            assert close(p.grad, q.grad)


@pytest.mark.parametrize("shape", [(2, 8, 5, 7), (1, 96, 56, 56), (2, 16, 14, 14), (3, 5, 7, 7), (1, 4, 33, 40),
                                   (1, 3, 100, 60), (2, 2, 96, 96), (1, 2, 65, 130)])     # the last three: row strips (32 + halo)
def test_dwconv_silu_cross_forward_backward(shape, layout):
    from medmamba_amd.ops import dwconv_silu_cross
    from oracle.model_ref import dwconv_silu_cross_ref
    B, D, H, W = shape
    L = H * W
    g = torch.Generator().manual_seed(sum(shape))
    xz = torch.randn(B, 2 * D, L, generator=g)              # x_cf is a batch-strided view, as inside SS2D
    w, bias = torch.randn(D, 1, 3, 3, generator=g) * 0.5, torch.randn(D, generator=g)
    du2 = torch.randn(B, 2 * D, L, generator=g)
    xr, wr, br = xz.clone().requires_grad_(), w.clone().requires_grad_(), bias.clone().requires_grad_()
    ref = dwconv_silu_cross_ref(xr[:, :D], wr, br, H, W)
    ref.backward(du2)
    xd, wd, bd = xz.to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), bias.to(DEV).requires_grad_()
    out = dwconv_silu_cross(xd[:, :D], wd, bd, H, W)
    out.backward(_cm(du2.to(DEV)) if layout == "cm" else du2.to(DEV))
    close = lambda a, b, tol: (a.detach().cpu() - b.detach()).abs().max().item() <= tol * max(1.0, b.detach().abs().max().item())
    assert close(out, ref, 2e-6)
    assert close(xd.grad, xr.grad, 1e-5)
    assert close(wd.grad, wr.grad, 2e-5)
    assert close(bd.grad, br.grad, 2e-5)


@pytest.mark.parametrize("shape", [(2, 8, 5, 7, 1), (1, 24, 14, 14, 2), (2, 16, 9, 4, 3), (1, 8, 33, 36, 1), (1, 160, 6, 6, 5)])
def test_ss2d_core_forward_backward(shape, layout):
    """projections + scan + cross-merge + out_norm + gate (channel-first HIP path) vs the oracle chain (einsums, explicit
    flips, torch LN).  The last shape has D > 128: a direction is split over several backward workgroups (atomics)."""
    from medmamba_amd.ops import ss2d_core
    from oracle.model_ref import ss2d_core_ref
    B, D, H, W, R = shape
    L, N = H * W, 16
    g = torch.Generator().manual_seed(sum(shape))
    mk = lambda *s: torch.randn(*s, generator=g)
    u2 = mk(B, 2 * D, L)
    Wx, Wdt = mk(4, R + 2 * N, D) / D ** 0.5, mk(4, D, R) / R ** 0.5
    A_logs = mk(4 * D, N) * 0.5
    Dp, dbias = mk(4 * D), mk(4, D) - 3
    z, lw, lb = mk(B, D, L), 1 + 0.1 * mk(D), 0.1 * mk(D)
    dy = mk(B, D, L)
    leaves = (u2, Wx, Wdt, dbias, A_logs, Dp, z, lw, lb)       # parameters in the module's layout / direction order
    ref_in = [t.clone().requires_grad_() for t in leaves]
    ref = ss2d_core_ref(*ref_in, H, W, 1e-5)
    ref.backward(dy)
    dev_in = [t.to(DEV).requires_grad_() for t in leaves]
    if layout == "cm":      # inputs as the neighbouring ops hand them over in this layout; u2 / z stay the autograd leaves
        out = ss2d_core(_cm(dev_in[0]), *dev_in[1:6], _cm(dev_in[6]), *dev_in[7:], H, W, 1e-5)
        out.backward(_cm(dy.to(DEV)))
    else:
        out = ss2d_core(*dev_in, H, W, 1e-5)
        out.backward(dy.to(DEV))
    err = (out.detach().cpu() - ref.detach()).abs().max().item()
    assert err <= 5e-5 * max(1.0, ref.detach().abs().max().item()), err
    names = ["du2", "dWx", "dWdt", "dbias", "dA_logs", "dD", "dz", "dln_w", "dln_b"]
    for n, a, b in zip(names, dev_in, ref_in):
        e = (a.grad.cpu() - b.grad).abs().max().item() / max(1.0, b.grad.abs().max().item())
        assert e <= 5e-4, (n, e)


@pytest.mark.parametrize("shape", [(2, 5, 7, 16), (1, 56, 56, 96), (2, 14, 14, 384), (3, 7, 9, 66), (1, 4, 4, 1024)])
def test_block_split_forward_backward(shape):
    from medmamba_amd.ops import block_split
    from oracle.model_ref import block_split_ref
    B, H, W, C = shape
    g = torch.Generator().manual_seed(sum(shape))
    inp = torch.randn(B, H, W, C, generator=g) * 2 + 0.5
    gamma, beta = 1 + 0.2 * torch.randn(C // 2, generator=g), 0.2 * torch.randn(C // 2, generator=g)
    dl, dr = torch.randn(B, C // 2, H, W, generator=g), torch.randn(B, H, W, C // 2, generator=g)
    ri, rg_, rb = inp.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    dres = torch.randn(B, H, W, C, generator=g)             # gradient arriving through the residual alias of inp
    l0, r0, i0 = block_split_ref(ri, rg_, rb, 1e-5)
    torch.autograd.backward([l0, r0, i0], [dl, dr, dres])
    di, dg, db = inp.to(DEV).requires_grad_(), gamma.to(DEV).requires_grad_(), beta.to(DEV).requires_grad_()
    l1, r1, i1 = block_split(di, dg, db, 1e-5)
    torch.autograd.backward([l1, r1, i1], [dl.to(DEV), dr.to(DEV), dres.to(DEV)])
    assert torch.equal(l1.detach().cpu(), l0.detach()) and torch.equal(i1.detach().cpu(), inp)
    close = lambda a, b, tol: (a.detach().cpu() - b.detach()).abs().max().item() <= tol * max(1.0, b.detach().abs().max().item())
    assert close(r1, r0, 5e-6)
    assert close(di.grad, ri.grad, 2e-5)
    assert close(dg.grad, rg_.grad, 5e-5)
    assert close(db.grad, rb.grad, 5e-5)
    # residual alias unused -> no residual gradient (NULL) path
    di2 = inp.to(DEV).requires_grad_()
    l2, r2, _ = block_split(di2, dg.detach(), db.detach(), 1e-5)
    torch.autograd.backward([l2, r2], [dl.to(DEV), dr.to(DEV)])
    ri2 = inp.clone().requires_grad_()
    l3, r3, _ = block_split_ref(ri2, gamma, beta, 1e-5)
    torch.autograd.backward([l3, r3], [dl, dr])
    assert close(di2.grad, ri2.grad, 2e-5)


@pytest.mark.parametrize("shape", [(2, 8, 5, 7, 1), (1, 24, 14, 14, 2), (2, 16, 9, 4, 3), (1, 160, 6, 6, 5)])
def test_ss2d_conv_core_forward_backward(shape, layout):
    """depthwise conv + SiLU + projections + scan + merge + out_norm + gate as ONE Function (the backward folds the scan's
    per-direction input gradients and the projection's into the conv's backward kernel) vs the oracle chain."""
    from medmamba_amd.ops import ss2d_conv_core
    from oracle.model_ref import ss2d_conv_core_ref
    B, D, H, W, R = shape
    L, N = H * W, 16
    g = torch.Generator().manual_seed(sum(shape) + 1)
    mk = lambda *s: torch.randn(*s, generator=g)
    x = mk(B, D, L)
    cw, cb = mk(D, 1, 3, 3) * 0.5, mk(D) * 0.2
    Wx, Wdt = mk(4, R + 2 * N, D) / D ** 0.5, mk(4, D, R) / R ** 0.5
    A_logs = mk(4 * D, N) * 0.5
    Dp, dbias = mk(4 * D), mk(4, D) - 3
    z, lw, lb = mk(B, D, L), 1 + 0.1 * mk(D), 0.1 * mk(D)
    dy = mk(B, D, L)
    leaves = (x, cw, cb, Wx, Wdt, dbias, A_logs, Dp, z, lw, lb)
    ref_in = [t.clone().requires_grad_() for t in leaves]
    ref = ss2d_conv_core_ref(*ref_in, H, W, 1e-5)
    ref.backward(dy)
    dev_in = [t.to(DEV).requires_grad_() for t in leaves]
    if layout == "cm":
        out = ss2d_conv_core(_cm(dev_in[0]), *dev_in[1:8], _cm(dev_in[8]), *dev_in[9:], H, W, 1e-5)
        out.backward(_cm(dy.to(DEV)))
    else:
        out = ss2d_conv_core(*dev_in, H, W, 1e-5)
        out.backward(dy.to(DEV))
    err = (out.detach().cpu() - ref.detach()).abs().max().item()
    assert err <= 5e-5 * max(1.0, ref.detach().abs().max().item()), err
    names = ["dx", "dconv_w", "dconv_b", "dWx", "dWdt", "dbias", "dA_logs", "dD", "dz", "dln_w", "dln_b"]
    for n, a, b in zip(names, dev_in, ref_in):
        e = (a.grad.cpu() - b.grad).abs().max().item() / max(1.0, b.grad.abs().max().item())
        assert e <= 5e-4, (n, e)


@pytest.mark.parametrize("shape", [(64, 48, 56, 56), (3, 5, 7, 9), (2, 300, 4, 4), (1, 8, 1, 1), (5, 16, 10, 6)])
def test_channel_sum_nchw(shape):
    from medmamba_amd.ops import channel_sum_nchw
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    ref = x.double().sum(dim=(0, 2, 3))
    got = channel_sum_nchw(x.to(DEV)).cpu().double()
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, x.abs().sum(dim=(0, 2, 3)).max().item() / 10)


def test_conv_bias_and_pointwise_functions_match_autograd():
    """ConvBiasFn / PointwiseConvFn: same values and gradients as nn.Conv2d under autograd (bias gradient by our kernel)."""
    from medmamba_amd.ops import PointwiseConvFn, conv2d_bias
    torch.manual_seed(5)
    for cin, k, hw in ((24, 3, 14), (48, 3, 9), (16, 3, 32)):        # the last one has planes >= 512 positions: our bias kernel
        conv = torch.nn.Conv2d(cin, cin, k, padding=k // 2).to(DEV)
        x = torch.randn(4, cin, hw, hw, device=DEV)
        gy = torch.randn(4, cin, hw, hw, device=DEV)
        xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
        ya = conv(xa); ya.backward(gy)
        ref = [xa.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone()]
        conv.zero_grad(set_to_none=True)
        yb = conv2d_bias(xb, conv); yb.backward(gy)
        near = lambda a, b: (a - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item())   # MIOpen may pick another solver
        assert torch.equal(ya, yb) and near(xb.grad, ref[0]) and near(conv.weight.grad, ref[1]) and near(conv.bias.grad, ref[2])
    pw = torch.nn.Conv2d(32, 32, 1).to(DEV)
    x = torch.randn(3, 32, 6, 5, device=DEV); gy = torch.randn(3, 32, 6, 5, device=DEV)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya = pw(xa); ya.backward(gy)
    ref = [xa.grad.clone(), pw.weight.grad.clone(), pw.bias.grad.clone()]
    pw.zero_grad(set_to_none=True)
    yb = PointwiseConvFn.apply(xb, pw.weight, pw.bias); yb.backward(gy)
    close = lambda a, b: (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())
    assert close(yb, ya) and close(xb.grad, ref[0]) and close(pw.weight.grad, ref[1]) and close(pw.bias.grad, ref[2])


@pytest.mark.parametrize("case", ["even", "odd"])
def test_patch_merge_ln_vs_reference_fixture(case):
    """PatchMerging2D (MedMamba.py:79-119) with the fused gather + LayerNorm kernel against the fixture produced
    by the real reference (even input and the odd-size crop path), forward and backward; the gather itself bit-exact."""
    from medmamba_amd import ops
    from medmamba_amd.modules import PatchMerging2D
    fx = load_golden("patchmerge_c8.npz")
    x = torch.from_numpy(fx[f"x_{case}"]).to(DEV).requires_grad_()
    C = x.shape[-1]
    m = PatchMerging2D(dim=C)
    m.load_state_dict({k[3:]: torch.from_numpy(v.copy()) for k, v in fx.items() if k.startswith("sd/")})
    m.to(DEV)
    y = m(x)
    want = fx[f"y_{case}"]
    assert np.abs(y.detach().cpu().numpy() - want).max() <= 5e-6 * max(1.0, np.abs(want).max())
    # bit-exact gather: identity LayerNorm statistics are not available, so check through gamma = 1, beta = 0 on constant rows
    g = torch.ones(4 * C, device=DEV); b = torch.zeros(4 * C, device=DEV)
    xs = x.detach()
    ln = ops.patch_merge_ln(xs, g, b, 1e-5)
    B, H, W, _ = xs.shape
    h2, w2 = H // 2, W // 2
    gathered = xs[:, :2 * h2, :2 * w2].reshape(B, h2, 2, w2, 2, C).permute(0, 1, 3, 4, 2, 5).reshape(B, h2, w2, 4 * C)
    ref = torch.nn.functional.layer_norm(gathered, (4 * C,), g, b, 1e-5)
    assert float((ln - ref).abs().max()) <= 5e-6
    if f"dy_{case}" in fx:
        y.backward(torch.from_numpy(fx[f"dy_{case}"]).to(DEV))
        for name, got, key in (("dx", x.grad, f"dx_{case}"), ("dgamma", m.norm.weight.grad, f"grad_{case}/norm.weight"),
                               ("dbeta", m.norm.bias.grad, f"grad_{case}/norm.bias"), ("dW", m.reduction.weight.grad, f"grad_{case}/reduction.weight")):
            if key in fx:
                w = fx[key]
                assert np.abs(got.cpu().numpy() - w).max() <= 2e-5 * max(1.0, np.abs(w).max()), name
    else:   # no gradient fixture: autograd of the reference op chain on the same device is the check
        x2 = xs.clone().requires_grad_()
        gm = m.norm.weight.detach().clone().requires_grad_(); bt = m.norm.bias.detach().clone().requires_grad_()
        dy = torch.randn_like(ln)
        ops.patch_merge_ln(x2, gm, bt, 1e-5).backward(dy)
        x3 = xs.clone().requires_grad_()
        gm3 = gm.detach().clone().requires_grad_(); bt3 = bt.detach().clone().requires_grad_()
        g3 = x3[:, :2 * h2, :2 * w2].reshape(B, h2, 2, w2, 2, C).permute(0, 1, 3, 4, 2, 5).reshape(B, h2, w2, 4 * C)
        torch.nn.functional.layer_norm(g3, (4 * C,), gm3, bt3, 1e-5).backward(dy)
        for name, a, bb in (("dx", x2.grad, x3.grad), ("dgamma", gm.grad, gm3.grad), ("dbeta", bt.grad, bt3.grad)):
            assert float((a - bb).abs().max()) <= 2e-5 * max(1.0, float(bb.abs().max())), name


def test_patch_merge_ln_model_widths():
    """Row widths of the real models (4C = 384 ... 2048) against the op chain, fwd + bwd."""
    from medmamba_amd import ops
    for B, H, W, C in [(2, 8, 8, 96), (2, 6, 10, 192), (1, 4, 4, 384), (1, 6, 4, 512), (3, 5, 7, 128)]:
        g = torch.Generator(device=DEV).manual_seed(C)
        x = torch.randn(B, H, W, C, device=DEV, generator=g)
        gm = torch.randn(4 * C, device=DEV, generator=g); bt = torch.randn(4 * C, device=DEV, generator=g)
        h2, w2 = H // 2, W // 2
        dy = torch.randn(B, h2, w2, 4 * C, device=DEV, generator=g)
        a = [t.clone().requires_grad_() for t in (x, gm, bt)]
        ops.patch_merge_ln(a[0], a[1], a[2], 1e-5).backward(dy)
        b = [t.clone().requires_grad_() for t in (x, gm, bt)]
        g3 = b[0][:, :2 * h2, :2 * w2].reshape(B, h2, 2, w2, 2, C).permute(0, 1, 3, 4, 2, 5).reshape(B, h2, w2, 4 * C)
        ref = torch.nn.functional.layer_norm(g3, (4 * C,), b[1], b[2], 1e-5)
        ref.backward(dy)
        out = ops.patch_merge_ln(x, gm, bt, 1e-5)
        assert float((out - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
        for name, p, q in zip(("dx", "dgamma", "dbeta"), a, b):
            assert float((p.grad - q.grad).abs().max()) <= 5e-5 * max(1.0, float(q.grad.abs().max())), (C, name)


@pytest.mark.parametrize("cm", [False, True], ids=["bm", "cm"])
@pytest.mark.parametrize("shape", [(2, 192, 784), (3, 384, 196), (2, 768, 49), (2, 1024, 37), (1, 200, 50), (2, 520, 16), (1, 136, 3),
                                   (2, 96, 100), (1, 1500, 20)])
def test_ln_gate_kernels_all_plans(shape, cm):
    """mm_ln_gate_fwd / _bwd (out_norm LayerNorm over channels + SiLU(z) gate on channel-first planes, MedMamba.py:300-301)
    through every plan of glue.hip: one wave per position group (D <= 128), the cooperative kernels (4 / 8 waves share a
    position: the model widths 192 ... 1024 and widths between them), the two-pass fallback (D = 1500; the backward at 1024)
    — against torch's layer_norm + autograd, position counts that do not fill the 16-position tiles included."""
    from medmamba_amd import _lib, ops
    B, D, L = shape
    lib = _lib.lib()
    g = torch.Generator(device=DEV).manual_seed(D + L)
    pl = lambda d=D: ops._planes(B, d, L, DEV, cm).normal_(generator=g)
    m, z, dy = pl(), pl(), pl()
    gamma = 1 + 0.1 * torch.randn(D, device=DEV, generator=g); beta = 0.1 * torch.randn(D, device=DEV, generator=g)
    y, dm, dz = ops._planes(B, D, L, DEV, cm), ops._planes(B, D, L, DEV, cm), ops._planes(B, D, L, DEV, cm)
    mu, rstd = torch.empty(B, L, device=DEV), torch.empty(B, L, device=DEV)
    ws = torch.full((lib.mm_ln_gate_rows(B, D, L), 2 * D), float("nan"), device=DEV)
    P, st = ops._pl, _lib.raw_stream()
    _lib.check(lib.mm_ln_gate_fwd(*P(m), *P(z), gamma.data_ptr(), beta.data_ptr(), 1e-5, *P(y), mu.data_ptr(), rstd.data_ptr(), B, D, L, st), "fwd")
    _lib.check(lib.mm_ln_gate_bwd(*P(dy), *P(m), *P(z), gamma.data_ptr(), beta.data_ptr(), mu.data_ptr(), rstd.data_ptr(), *P(dm), *P(dz),
                                  ws.data_ptr(), B, D, L, st), "bwd")
    mr, zr, gr, br = (t.detach().clone().contiguous().requires_grad_() for t in (m, z, gamma, beta))
    ref = torch.nn.functional.layer_norm(mr.transpose(1, 2), (D,), gr, br, 1e-5).transpose(1, 2) * torch.nn.functional.silu(zr)
    ref.backward(dy.contiguous())
    close = lambda a, b, tol: float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))
    assert close(y, ref.detach(), 2e-5)
    assert close(mu, mr.detach().mean(1), 1e-5)
    assert close(dm, mr.grad, 5e-5) and close(dz, zr.grad, 5e-5)
    s = ws.sum(0)
    assert torch.isfinite(s).all()                      # every workspace row was written
    assert close(s[:D], gr.grad, 1e-4) and close(s[D:], br.grad, 1e-4)


def test_ln_gate_refuses_channel_strides_beyond_its_32_bit_offsets():
    """The register-resident LayerNorm + gate kernels address one step of channel rows with 32-bit byte offsets from a descriptor
    that starts at the step (glue.hip rows_rsrc): a channel stride of 2^26 floats (16 rows x 256 MB) no longer fits and the entry
    points say so (MM_ERR_SHAPE, nothing launched) instead of wrapping around."""
    from medmamba_amd import _lib
    lib, st = _lib.lib(), _lib.raw_stream()
    B, D, L = 2, 192, 64
    t = torch.zeros(4096, device=DEV)                     # never touched: the calls return before launching
    p, big = t.data_ptr(), 1 << 26
    assert lib.mm_ln_gate_fwd(p, D * big, big, p, D * L, L, p, p, 1e-5, p, D * L, L, p, p, B, D, L, st) == -2
    assert lib.mm_ln_gate_bwd(p, D * L, L, p, D * L, L, p, D * L, L, p, p, p, p, p, D * big, big, p, D * L, L, p, B, D, L, st) == -2


@pytest.mark.parametrize("cm", [False, True], ids=["bm", "cm"])
@pytest.mark.parametrize("shape", [(64, 96, 14, 14), (3, 40, 7, 7), (2, 5, 16, 16), (2, 7, 5, 9), (1, 3, 1, 1), (2, 6, 1, 200), (3, 9, 37, 1),
                                   (2, 4, 16, 17), (1, 8, 56, 56), (2, 3, 33, 40)])
def test_cross_merge_and_plane_transpose_kernels(shape, cm):
    """mm_cross_merge_fwd (MedMamba.py:282-286, 298: un-transpose the column-major pair, sum the four) and mm_plane_transpose,
    bit-exact against torch indexing — the one-wavefront-per-plane kernels (H*W <= 256) and the 32x32-tile kernels."""
    from medmamba_amd import _lib, ops
    B, D, H, W = shape
    L = H * W
    lib, st, P = _lib.lib(), _lib.raw_stream(), ops._pl
    g = torch.Generator(device=DEV).manual_seed(H * 100 + W)
    out4 = torch.randn(B, 4, D, L, device=DEV, generator=g)
    m = ops._planes(B, D, L, DEV, cm).fill_(float("nan"))
    _lib.check(lib.mm_cross_merge_fwd(out4.data_ptr(), *P(m), B, D, H, W, st), "merge")
    colmajor = (out4[:, 2] + out4[:, 3]).view(B, D, W, H).transpose(2, 3).reshape(B, D, L)
    want = (out4[:, 0] + out4[:, 1]) + colmajor
    assert torch.equal(m, want)
    src = ops._planes(B, 2 * D, L, DEV, cm).normal_(generator=g)
    before = src[:, :D].clone()
    _lib.check(lib.mm_plane_transpose(*P(src), *P(src[:, D:]), B, D, H, W, st), "transpose")
    assert torch.equal(src[:, D:], before.view(B, D, H, W).transpose(2, 3).reshape(B, D, L))
    assert torch.equal(src[:, :D], before)


def test_patch_embed_weight_gradient_in_deterministic_mode(monkeypatch):
    """cudnn.deterministic (the reference's mode): PatchEmbed2D's 4x4 / stride-4 conv (MedMamba.py:62) gets its weight gradient from
    one permuted copy + one batched GEMM + an ordered sum (ops.ConvBiasFn) instead of MIOpen's per-image solver — the same values
    as the default mode's, the same bits run after run."""
    from medmamba_amd import modules
    torch.manual_seed(4)
    pe = modules.PatchEmbed2D(patch_size=4, in_chans=3, embed_dim=96, norm_layer=torch.nn.LayerNorm).to(DEV)
    x = torch.randn(8, 3, 64, 96, device=DEV)
    g = torch.randn(8, 16, 24, 96, device=DEV)

    def run():
        pe.zero_grad(set_to_none=True)
        pe(x).backward(g)
        return {k: p.grad.clone() for k, p in pe.named_parameters()}

    ref = run()
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)
    a, b = run(), run()
    for k in ref:
        assert torch.equal(a[k], b[k]), k
        assert float((a[k] - ref[k]).abs().max()) <= 2e-4 * max(1e-3, float(ref[k].abs().max())), k


def test_patch_embed_permute_layernorm_kernel():
    """PatchEmbed2D's permute + LayerNorm (MedMamba.py:70-76) as one kernel each way against the op chain: model widths (96, 128),
    widths that are not a multiple of 64, the 512 limit, plane sizes that are not a multiple of the 32-position tile; and the
    module itself (conv + fused tail) against its own op-by-op path."""
    from medmamba_amd import ops
    from medmamba_amd.modules import PatchEmbed2D
    for B, C, H, W in [(2, 96, 56, 56), (1, 128, 12, 9), (3, 16, 5, 7), (1, 200, 3, 11), (2, 512, 4, 10), (1, 8, 1, 1)]:
        g = torch.Generator(device=DEV).manual_seed(C + H)
        x = torch.randn(B, C, H, W, device=DEV, generator=g) * 2 + 0.5
        gm = torch.randn(C, device=DEV, generator=g); bt = torch.randn(C, device=DEV, generator=g)
        dy = torch.randn(B, H, W, C, device=DEV, generator=g)
        a = [t.clone().requires_grad_() for t in (x, gm, bt)]
        out = ops.nchw_ln_rows(a[0], a[1], a[2], 1e-5)
        out.backward(dy)
        b = [t.clone().requires_grad_() for t in (x, gm, bt)]
        ref = torch.nn.functional.layer_norm(b[0].permute(0, 2, 3, 1), (C,), b[1], b[2], 1e-5)
        ref.backward(dy)
        assert out.shape == ref.shape and out.is_contiguous()
        assert float((out - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), (C, H, W)
        for name, p_, q_ in zip(("dx", "dgamma", "dbeta"), a, b):
            assert p_.grad.shape == q_.grad.shape
            assert float((p_.grad - q_.grad).abs().max()) <= 5e-5 * max(1.0, float(q_.grad.abs().max())), (C, H, W, name)
    assert not ops.nchw_ln_rows_supported(513)
    torch.manual_seed(3)
    pe = PatchEmbed2D(patch_size=4, in_chans=3, embed_dim=96, norm_layer=torch.nn.LayerNorm).to(DEV)
    img = torch.randn(2, 3, 40, 24, device=DEV)
    got = pe(img)
    want = pe.norm(pe.proj(img).permute(0, 2, 3, 1))
    assert got.shape == (2, 10, 6, 96) and float((got - want).abs().max()) <= 1e-5
    wide = PatchEmbed2D(patch_size=4, in_chans=3, embed_dim=520, norm_layer=torch.nn.LayerNorm).to(DEV)     # beyond the kernel
    assert wide(img).shape == (2, 10, 6, 520)
    assert PatchEmbed2D(patch_size=4, in_chans=3, embed_dim=32, norm_layer=None).to(DEV)(img).shape == (2, 10, 6, 32)


@pytest.mark.parametrize("shape", [(2, 8, 6, 8, 1), (1, 24, 14, 14, 2), (2, 96, 12, 12, 3), (1, 16, 8, 12, 8), (2, 16, 4, 4, 12)])
def test_ss2d_core_inference_with_fused_dt_projection(shape, layout, monkeypatch):
    """VERDICT r1 item 3 (forward half): under no_grad the dt projection (MedMamba.py:262) runs inside the scan kernel's
    staging phase (mm_scan_args.dt_w) — same values as the oracle chain and as the GEMM path, in both plane layouts; ranks
    above the kernel's limit (last shape: R = 12) and L % 4 != 0 shapes keep the GEMM."""
    from medmamba_amd import ops, selective_scan_interface as ssi
    from oracle.model_ref import ss2d_core_ref
    B, D, H, W, R = shape
    L, N = H * W, 16
    g = torch.Generator().manual_seed(sum(shape))
    mk = lambda *s: torch.randn(*s, generator=g)
    u2 = mk(B, 2 * D, L)
    Wx, Wdt = mk(4, R + 2 * N, D) / D ** 0.5, mk(4, D, R) / R ** 0.5
    A_logs, Dp, dbias = mk(4 * D, N) * 0.5, mk(4 * D), mk(4, D) - 3
    z, lw, lb = mk(B, D, L), 1 + 0.1 * mk(D), 0.1 * mk(D)
    leaves = (u2, Wx, Wdt, dbias, A_logs, Dp, z, lw, lb)
    ref = ss2d_core_ref(*leaves, H, W, 1e-5)
    dev_in = [t.to(DEV) for t in leaves]
    if layout == "cm":
        dev_in[0], dev_in[6] = _cm(dev_in[0]), _cm(dev_in[6])
    fused = []
    orig = ssi._launch_fwd
    def spy(*a, **k):
        fused.append(k.get("dt") is not None)
        return orig(*a, **k)
    monkeypatch.setattr(ssi, "_launch_fwd", spy)
    with torch.no_grad():
        out = ops.ss2d_core(*dev_in, H, W, 1e-5)
        monkeypatch.setattr(ops, "_FUSE_DT", False)
        out_gemm = ops.ss2d_core(*dev_in, H, W, 1e-5)
    assert fused == [R <= 8 and L % 4 == 0, False]
    scale = max(1.0, ref.abs().max().item())
    assert (out.cpu() - ref).abs().max().item() <= 1e-4 * scale
    assert (out.cpu() - out_gemm.cpu()).abs().max().item() <= 2e-5 * scale


@pytest.mark.parametrize("shape", [(64, 48, 56, 56), (8, 96, 28, 28), (4, 192, 14, 14), (3, 5, 7, 7), (2, 16, 1, 3), (64, 384, 7, 7)])
@pytest.mark.parametrize("relu", [False, True])
def test_own_batchnorm_relu_vs_torch(shape, relu):
    """csrc/bn.hip: training-mode BatchNorm2d (+ ReLU) against torch.nn.BatchNorm2d (+ nn.ReLU) on the same device — output,
    input / weight / bias gradients, running statistics and the batch counter (MedMamba.py:338-344)."""
    from medmamba_amd.ops import bn_relu_train
    B, C, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(B, C, H, W, device=DEV, generator=g) * 2 + 0.5
    dy = torch.randn(B, C, H, W, device=DEV, generator=g)
    ref = torch.nn.BatchNorm2d(C).to(DEV).train()
    with torch.no_grad():
        ref.weight.copy_(torch.randn(C, device=DEV, generator=g)); ref.bias.copy_(torch.randn(C, device=DEV, generator=g) * 0.3)
    own = torch.nn.BatchNorm2d(C).to(DEV).train()
    own.load_state_dict(ref.state_dict())
    xr, xo = x.clone().requires_grad_(), x.clone().requires_grad_()
    for step in range(2):                                   # two steps: running statistics accumulate
        yr = ref(xr); yr = torch.relu(yr) if relu else yr
        yo = bn_relu_train(xo, own, relu)
    yr.backward(dy); yo.backward(dy)
    close = lambda a, b, tol: float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))
    assert close(yo, yr, 2e-6)
    assert close(xo.grad, xr.grad, 2e-5)
    assert close(own.weight.grad, ref.weight.grad, 2e-5) and close(own.bias.grad, ref.bias.grad, 2e-5)
    assert close(own.running_mean, ref.running_mean, 1e-6) and close(own.running_var, ref.running_var, 2e-6)
    assert int(own.num_batches_tracked) == int(ref.num_batches_tracked) == 2


def test_own_batchnorm_relu_mask_is_the_forward_mask():
    """the backward's ReLU mask must use the forward's own expression y = fmaf(x, rstd*gamma, beta - mean*rstd*gamma)
    (torch masks on the stored y).  Many pre-activations at or next to zero: wherever the forward output is 0 the input gradient
    must carry no contribution of dy, and vice versa — compared against nn.BatchNorm2d + ReLU, which masks on y."""
    from medmamba_amd.ops import bn_relu_train
    B, C, H, W = 8, 32, 14, 14
    g = torch.Generator(device=DEV).manual_seed(77)
    x = torch.randn(B, C, H, W, device=DEV, generator=g)
    own = torch.nn.BatchNorm2d(C).to(DEV).train()
    with torch.no_grad():
        own.weight.copy_(torch.randn(C, device=DEV, generator=g))
        # beta chosen so that bn(x) of the median element of each channel lands within a few ulps of zero
        mean, var = x.mean((0, 2, 3)), x.var((0, 2, 3), unbiased=False)
        med = x.permute(1, 0, 2, 3).reshape(C, -1).median(dim=1).values
        own.bias.copy_(-(med - mean) / torch.sqrt(var + own.eps) * own.weight)
    ref = torch.nn.BatchNorm2d(C).to(DEV).train()
    ref.load_state_dict(own.state_dict())
    xo, xr = x.clone().requires_grad_(), x.clone().requires_grad_()
    yo = bn_relu_train(xo, own, True)
    yr = torch.relu(ref(xr))
    near = (yr.detach().abs() < 1e-5).float().mean().item()
    assert near > 0.0                                        # the construction really puts elements at the threshold
    # a gradient that isolates the mask: dy = 1 on one element per channel-plane would be drowned by the mean terms, so compare
    # d(sum(y * r)) for random r against the same quantity computed from OUR forward output's mask
    r = torch.randn(B, C, H, W, device=DEV, generator=g)
    (yo * r).sum().backward()
    mask = (yo.detach() > 0).float()
    # reference gradient of BatchNorm for an upstream gradient r*mask (torch's own BN backward, mask from OUR y)
    (ref(xr) * (r * mask)).sum().backward()
    scale = max(1.0, float(xr.grad.abs().max()))
    assert float((xo.grad - xr.grad).abs().max()) <= 2e-5 * scale
    assert float((own.weight.grad - ref.weight.grad).abs().max()) <= 2e-5 * max(1.0, float(ref.weight.grad.abs().max()))


def test_small_planes_ignore_the_strip_knob():
    """planes of <= 256 positions always run the one-wavefront-per-plane depthwise-conv kernels (one row of partial
    sums per plane); mm_dwconv_silu_cross_strips, which sizes the caller's workspace, must say 1 for them."""
    from medmamba_amd import _lib
    lib = _lib.lib()
    for H, W in [(14, 14), (7, 7), (16, 16), (1, 200), (4, 64)]:
        assert lib.mm_dwconv_silu_cross_strips(H, W) == 1
    assert lib.mm_dwconv_silu_cross_strips(56, 56) >= 1 and lib.mm_dwconv_silu_cross_strips(96, 96) == 3


@pytest.mark.parametrize("shape", [(64, 192, 192, 14, 14), (2, 5, 7, 6, 9), (3, 48, 48, 56, 56), (5, 384, 384, 7, 7), (1, 8, 64, 1, 1),
                                   (2, 70, 132, 11, 3), (2, 20, 8, 28, 28), (1, 16, 32, 5, 100)])
@pytest.mark.parametrize("mode", ["plain", "affine_relu", "nobias"])
@pytest.mark.parametrize("version", [1, 2])
def test_own_conv3x3_forward_vs_fp64(shape, mode, version):
    """csrc/conv.hip (fp32 MFMA implicit-GEMM 3x3 conv, MedMamba.py:339, 342): output against an fp64 convolution, the optional
    input affine + ReLU (a BatchNorm folded in front, zero padding AFTER it), and the (count, mean, M2) partials of the epilogue
    merged by mm_bn_relu_fwd_stats against torch.nn.BatchNorm2d on the same conv output."""
    from medmamba_amd import _lib
    B, C, K, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(B, C, H, W, device=DEV, generator=g)
    w = torch.randn(K, C, 3, 3, device=DEV, generator=g) / (3.0 * C ** 0.5)
    b = None if mode == "nobias" else torch.randn(K, device=DEV, generator=g)
    aff = torch.cat([torch.rand(C, device=DEV, generator=g) + 0.5, torch.randn(C, device=DEV, generator=g) * 0.3]) if mode == "affine_relu" else None
    lib = _lib.exp_lib()          # the experiments build (lib/libmedmamba_hip_exp.so); the product library has no conv kernels
    assert lib is not None, "python -m medmamba_amd.build --experiments"
    y = torch.empty(B, K, H, W, device=DEV)
    if version == 1:
        stats = torch.empty(lib.mm_conv3x3_fwd_tiles(B, H, W), K, 3, device=DEV)
        rc = lib.mm_conv3x3_fwd(x.data_ptr(), w.data_ptr(), None if b is None else b.data_ptr(), None if aff is None else aff.data_ptr(),
                                int(aff is not None), y.data_ptr(), stats.data_ptr(), B, C, K, H, W, torch.cuda.current_stream().cuda_stream)
    else:
        if K % 4:
            pytest.skip("the second version needs K % 4 == 0")
        wt = w.permute(1, 2, 3, 0).reshape(C * 9, K).contiguous()
        stats = torch.empty(lib.mm_conv3x3_v2_tiles(B, H, W), K, 3, device=DEV)
        rc = lib.mm_conv3x3_v2_fwd(x.data_ptr(), wt.data_ptr(), None if b is None else b.data_ptr(), None if aff is None else aff.data_ptr(),
                                   int(aff is not None), y.data_ptr(), stats.data_ptr(), B, C, K, H, W, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    xin = x.double()
    if aff is not None:
        xin = torch.relu(xin * aff[:C].double()[None, :, None, None] + aff[C:].double()[None, :, None, None])
    ref = torch.nn.functional.conv2d(xin.cpu(), w.double().cpu(), None if b is None else b.double().cpu(), padding=1)
    err = float((y.cpu().double() - ref).abs().max())
    assert err <= 2e-5 * max(1.0, float(ref.abs().max())), err
    # statistics partials -> BatchNorm apply; reference: torch's training-mode BatchNorm2d + ReLU on OUR conv output
    if B * H * W >= 2:
        bn = torch.nn.BatchNorm2d(K).to(DEV).train()
        with torch.no_grad():
            bn.weight.copy_(torch.randn(K, device=DEV, generator=g)); bn.bias.copy_(torch.randn(K, device=DEV, generator=g) * 0.2)
        want = torch.relu(bn(y))
        got = torch.empty_like(y)
        mean, rstd = torch.empty(K, device=DEV), torch.empty(K, device=DEV)
        rm, rv = torch.zeros(K, device=DEV), torch.ones(K, device=DEV)
        rc = lib.mm_bn_relu_fwd_stats(y.data_ptr(), stats.data_ptr(), stats.shape[0], bn.weight.data_ptr(), bn.bias.data_ptr(), bn.eps, 0.1,
                                      rm.data_ptr(), rv.data_ptr(), got.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 1, B, K, H * W,
                                      torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        assert float((got - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))
        assert float((rm - bn.running_mean).abs().max()) <= 1e-5 * max(1.0, float(bn.running_mean.abs().max()))
        assert float((rv - bn.running_var).abs().max()) <= 1e-4 * max(1.0, float(bn.running_var.abs().max()))


def test_own_conv_branch_matches_miopen_branch(monkeypatch):
    """MM_OWN_CONV=1: a block's conv branch with our forward convs (statistics in their epilogue) against the MIOpen route,
    forward and all gradients, on the reference block fixture's weights."""
    from medmamba_amd import modules, ops
    from conftest import load_golden, split_sd
    fx = load_golden("block_c16.npz")
    blk = modules.SS_Conv_SSM(hidden_dim=16, drop_path=0.0, norm_layer=torch.nn.LayerNorm)
    blk.load_state_dict(split_sd(fx))
    blk.to(DEV).train()
    x = torch.from_numpy(fx["x"]).to(DEV)
    res = {}
    for own in (False, True):
        monkeypatch.setattr(ops, "_OWN_CONV", own)
        blk.load_state_dict({k: v.to(DEV) for k, v in split_sd(fx).items()})        # same running statistics at the start
        blk.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_()
        y = blk(xi)
        (y * y).mean().backward()
        res[own] = (y.detach().clone(), xi.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()},
                    {k: v.clone() for k, v in blk.state_dict().items() if "running" in k})
    assert float((res[True][0] - res[False][0]).abs().max()) <= 2e-5 * max(1.0, float(res[False][0].abs().max()))
    assert float((res[True][1] - res[False][1]).abs().max()) <= 1e-4 * max(1e-3, float(res[False][1].abs().max()))
    for k, v in res[False][2].items():
        if k.endswith("conv33conv33conv11.1.bias") or k.endswith("conv33conv33conv11.4.bias"):
            continue                                                             # exact-zero true gradient: rounding noise
        assert float((res[True][2][k] - v).abs().max()) <= 2e-4 * max(1e-3, float(v.abs().max())), k
    for k, v in res[False][3].items():
        assert float((res[True][3][k] - v).abs().max()) <= 1e-5 * max(1.0, float(v.abs().max())), k


@pytest.mark.parametrize("shape", [(3, 5, 7, 9), (2, 48, 56, 56), (4, 16, 1, 1), (1, 3, 2, 300), (180, 384, 2, 2)])
def test_im2col3x3_is_unfold(shape):
    """mm_im2col3x3 (every image in one launch) == torch.nn.functional.unfold(x, 3, padding=1) bit for bit.  The last shape has
    batch * C = 69120 planes, beyond HIP's 65535 limit of grid.y (384 channels from 171 images on)."""
    from medmamba_amd import _lib
    B, C, H, W = shape
    x = torch.randn(*shape, device=DEV)
    cols = torch.full((B, 9 * C, H * W), float("nan"), device=DEV)
    _lib.check(_lib.lib().mm_im2col3x3(x.data_ptr(), cols.data_ptr(), B, C, H, W, 1, _lib.raw_stream()), "mm_im2col3x3")
    ref = torch.nn.functional.unfold(x, 3, padding=1)
    assert torch.equal(cols, ref)
    for gs in (2, 4):                        # gs images side by side: (B/gs, 9C, gs*HW)
        if B % gs == 0:
            grouped = torch.full((B // gs, 9 * C, gs * H * W), float("nan"), device=DEV)
            _lib.check(_lib.lib().mm_im2col3x3(x.data_ptr(), grouped.data_ptr(), B, C, H, W, gs, _lib.raw_stream()), "mm_im2col3x3")
            want = ref.view(B // gs, gs, 9 * C, H * W).transpose(1, 2).reshape(B // gs, 9 * C, gs * H * W)
            assert torch.equal(grouped, want), gs
    assert _lib.lib().mm_im2col3x3(x.data_ptr(), cols.data_ptr(), B, C, H, W, 0, None) == -2


@pytest.mark.parametrize("CL", [(44, 196), (43, 49)], ids=["vector", "scalar"])
@pytest.mark.parametrize("cm", [False, True])
def test_sum_lead_chunks_fills_the_bc_rows_of_dx_dbl(cm, CL):
    """mm_sum_lead_chunks: the per-workgroup partial dB / dC planes of the backward scan summed straight into rows R.. of every
    direction of d(x_dbl) (batch-major: (B, 4, C, L); channel-major: (4, C, B*L)) — same bits as mm_sum_lead on a dense copy, the dt
    rows in between untouched."""
    from medmamba_amd import _lib, selective_scan_interface as ssi
    (C, L), B, W = CL, 5, 6
    R = C - 32              # (an odd chunk stride takes the dword path of the kernel)
    g = torch.Generator(device=DEV).manual_seed(3)
    if cm:
        dx_dbl = torch.full((4, C, B * L), 7.0, device=DEV)
        dst = dx_dbl.view(4, C, B, L).permute(2, 0, 1, 3)[:, :, R:]
    else:
        dx_dbl = torch.full((B, 4, C, L), 7.0, device=DEV)
        dst = dx_dbl[:, :, R:]
    planes = ssi._like_strided(dst, W)
    planes.copy_(torch.randn(W, *dst.shape, device=DEV, generator=g))
    ch = ssi._chunks_of(dst)
    assert ch == ((4, 32 * B * L, C * B * L) if cm else (B * 4, 32 * L, C * L))
    _lib.check(_lib.lib().mm_sum_lead_chunks(planes.data_ptr(), dst.data_ptr(), W, ch[0], ch[1], ch[2], _lib.raw_stream()), "mm_sum_lead_chunks")
    from medmamba_amd import ops
    dense = ops.sum_lead(planes.permute(0, *[i + 1 for i in sorted(range(dst.dim()), key=lambda i: -dst.stride(i))]).contiguous())
    want = planes.double().sum(0)
    assert float((dst.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    order = sorted(range(dst.dim()), key=lambda i: -dst.stride(i))
    assert torch.equal(dst.permute(*order).contiguous(), dense)                # the same bits as the dense sum
    rest = dx_dbl.view(4, C, B, L)[:, :R] if cm else dx_dbl[:, :, :R]
    assert bool((rest == 7.0).all())                                           # the dt rows are not written
    assert _lib.lib().mm_sum_lead_chunks(planes.data_ptr(), dst.data_ptr(), W, 0, 8, 8, None) == -2


@pytest.mark.parametrize("shape", [(64, 4, 35, 96), (2, 7), (3, 1), (64, 768, 384), (5, 13), (17, 2, 48, 48), (130, 10), (1, 8), (300, 8), (256, 2 * 96), (200, 1536), (33, 4, 3), (1024, 96), (4096, 768), (3136, 6), (5000, 16)])
def test_sum_lead_matches_torch_sum_and_is_reproducible(shape):
    """mm_sum_lead (ops.sum_lead / csrc_host sum_lead): t.sum(0) of dense fp32 tensors — the batch sums behind the batched
    weight-gradient GEMMs and the partial rows — vector and scalar form, into a fresh tensor and into a given one."""
    from medmamba_amd import ops
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    t = torch.randn(*shape, device=DEV, generator=g)
    want = t.double().sum(0)
    got = ops.sum_lead(t)
    assert got.shape == want.shape
    assert float((got.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    out = torch.full(shape[1:], float("nan"), device=DEV)
    assert ops.sum_lead(t, out=out) is out and torch.equal(out, got)          # same bits on every call: fixed summation order
    # a view that is not dense takes torch's path and still gives the sum
    if len(shape) >= 2 and shape[-1] > 1:
        tv = t[..., ::2]
        assert torch.allclose(ops.sum_lead(tv), tv.sum(0), rtol=1e-5, atol=1e-5)
