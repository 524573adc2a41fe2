"""GPU: d_state other than the 16 the kernels hold per channel (SS2D's d_state argument, MedMamba.py:187,329,457; VSSM passes
it through to every block).  selective_scan_fn runs them as 16-state slices of the same HIP kernels (zero B/C rows fill a
short slice); the blocks then take the reference's op chain around it.  Checked against the CPU oracle, which handles any N.
Tolerances as in test_scan_parity.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FWD_RTOL = 2e-5
BWD_RTOL = 2e-4


def _inputs(batch, G, H, L, N, R=2, seed=0):
    g = torch.Generator().manual_seed(seed)
    dim = G * H
    u = torch.randn(batch, dim, L, generator=g)
    delta = torch.randn(batch, dim, L, generator=g)
    A = -torch.exp(torch.randn(dim, N, generator=g) * 0.5)
    x_dbl = torch.randn(batch, G, R + 2 * N, L, generator=g)
    D = torch.randn(dim, generator=g)
    bias = torch.randn(dim, generator=g) * 0.3
    dout = torch.randn(batch, dim, L, generator=g)
    return u, delta, A, x_dbl, D, bias, dout, R


@pytest.mark.parametrize("N", [1, 4, 8, 15, 17, 24, 32, 40])
@pytest.mark.parametrize("shape", [(2, 4, 8, 37), (1, 2, 5, 196)])
def test_selective_scan_any_d_state(N, shape):
    from medmamba_amd import selective_scan_fn
    from oracle.scan_ref import c_scan_bwd, c_scan_fwd
    dev = torch.device("cuda:0")
    u, delta, A, x_dbl, D, bias, dout, R = _inputs(*shape, N, seed=N + sum(shape))
    Bh, Ch = x_dbl[:, :, R:R + N], x_dbl[:, :, R + N:]
    ins = [t.to(dev).requires_grad_() for t in (u, delta, A, x_dbl, D, bias)]
    ud, dd, Ad, xd, Dd, bd = ins
    out = selective_scan_fn(ud, dd, Ad, xd[:, :, R:R + N], xd[:, :, R + N:], Dd, z=None, delta_bias=bd, delta_softplus=True)
    assert out.shape == u.shape and out.dtype == torch.float32
    o64 = c_scan_fwd(u, delta, A, Bh, Ch, D, bias, True, f64=True)
    o32 = c_scan_fwd(u, delta, A, Bh, Ch, D, bias, True, f64=False)
    scale = np.abs(o64).max()
    err = np.abs(out.detach().cpu().numpy() - o64).max()
    assert err <= 2 * np.abs(o32 - o64).max() + FWD_RTOL * scale, (err, scale)
    out.backward(dout.to(dev))
    r = c_scan_bwd(u, delta, A, Bh, Ch, D, bias, dout, True)
    gx = xd.grad.cpu().numpy()
    got = dict(du=ud.grad, ddelta=dd.grad, dA=Ad.grad, dD=Dd.grad, ddelta_bias=bd.grad)
    got = {k: v.cpu().numpy() for k, v in got.items()}
    got["dB"], got["dC"] = gx[:, :, R:R + N], gx[:, :, R + N:]
    assert np.abs(gx[:, :, :R]).max() == 0.0                      # the dt rows of x_dbl are not an input of the scan
    for k, g in got.items():
        want = r[k]
        assert g.shape == want.shape, k
        e = np.abs(g - want).max() / max(1.0, np.abs(want).max())
        assert e <= BWD_RTOL, (k, N, e)


def test_d_state_without_D_and_bias():
    from medmamba_amd import selective_scan_fn
    from oracle.scan_ref import c_scan_fwd
    dev = torch.device("cuda:0")
    u, delta, A, x_dbl, _, _, _, R = _inputs(1, 4, 6, 50, 20, seed=5)
    Bh, Ch = x_dbl[:, :, R:R + 20].contiguous(), x_dbl[:, :, R + 20:].contiguous()
    out = selective_scan_fn(u.to(dev), delta.to(dev), A.to(dev), Bh.to(dev), Ch.to(dev), None, delta_softplus=False)
    o64 = c_scan_fwd(u, delta, A, Bh, Ch, None, None, False, f64=True)
    o32 = c_scan_fwd(u, delta, A, Bh, Ch, None, None, False, f64=False)
    assert np.abs(out.cpu().numpy() - o64).max() <= 2 * np.abs(o32 - o64).max() + FWD_RTOL * np.abs(o64).max()


@pytest.mark.parametrize("d_state", [8, 24])
def test_tiny_vssm_with_other_d_state_vs_oracle(d_state):
    """VSSM(d_state=...) end to end (logits, loss gradients) against the oracle model built from the same state dict."""
    import torch.nn.functional as F
    from medmamba_amd.modules import VSSM
    from oracle import model_ref
    from oracle.scan_ref import c_selective_scan_fn
    torch.manual_seed(d_state)
    depths, dims = [1, 1], [16, 32]
    net = VSSM(patch_size=4, in_chans=3, num_classes=5, depths=depths, dims=dims, d_state=d_state, drop_path_rate=0.0)
    assert net.layers[0].blocks[0].self_attention.A_logs.shape[1] == d_state
    x = torch.randn(2, 3, 32, 32)
    y = torch.tensor([1, 3])
    learnable = {k for k, _ in net.named_parameters()}
    p = {k: v.detach().clone().requires_grad_(k in learnable) for k, v in net.state_dict().items()}
    bn_updates = {}
    ref_logits = model_ref.vssm_forward(p, x, depths, scan=c_selective_scan_fn, training=True, bn_updates=bn_updates)
    ref_loss = F.cross_entropy(ref_logits, y)
    ref_loss.backward()

    net = net.cuda().train()
    logits = net(x.cuda())
    loss = F.cross_entropy(logits, y.cuda())
    loss.backward()
    scale = ref_logits.detach().abs().max().item()
    assert (logits.detach().cpu() - ref_logits.detach()).abs().max().item() <= 2e-4 * max(1.0, scale)
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * max(1.0, abs(ref_loss.item()))
    checked = 0
    for name, prm in net.named_parameters():
        want = p[name].grad
        if want is None:
            continue
        e = (prm.grad.cpu() - want).abs().max().item() / max(1.0, want.abs().max().item())
        assert e <= 5e-4, (name, e)
        checked += 1
    assert checked >= 40

    net.eval()
    with torch.no_grad():
        ev = net(x.cuda())
        ref_ev = model_ref.vssm_forward({k: v.detach() for k, v in net.cpu().state_dict().items()}, x, depths,
                                        scan=c_selective_scan_fn, training=False)
    assert (ev.cpu() - ref_ev).abs().max().item() <= 2e-4 * max(1.0, ref_ev.abs().max().item())
