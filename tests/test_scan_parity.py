"""GPU: HIP selective scan (through the C ABI) vs the CPU oracle and the committed golden vectors.

Tolerances (fp32 path; north_star: "match ... to a stated fp32 tolerance"):
  forward : max|hip - f64 oracle| <= 2 * max|f32 oracle - f64 oracle| + 2e-5 * max|ref|
  backward: max|hip - f64 oracle| <= 2e-4 * max(1, max|ref|)   per gradient tensor
(the HIP kernels use v_exp_f32 on A*log2(e)-prescaled arguments and a different summation order
than the PyTorch loop; the fp64 run of the oracle arbitrates — SURVEY §8c.)
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

FWD_RTOL = 2e-5
BWD_RTOL = 2e-4


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _make(batch, G, H, L, R=3, seed=0, contiguous_bc=False):
    g = torch.Generator().manual_seed(seed)
    N, dim = 16, G * H
    u = torch.randn(batch, dim, L, generator=g)
    delta = torch.randn(batch, dim, L, generator=g)
    A = -torch.exp(torch.randn(dim, N, generator=g) * 0.5)
    x_dbl = torch.randn(batch, G, R + 2 * N, L, generator=g)
    Bs, Cs = x_dbl[:, :, R:R + N], x_dbl[:, :, R + N:]
    if contiguous_bc:
        Bs, Cs = Bs.contiguous(), Cs.contiguous()
    D = torch.randn(dim, generator=g)
    dt = torch.exp(torch.rand(dim, generator=g) * (np.log(0.1) - np.log(1e-3)) + np.log(1e-3))
    bias = dt + torch.log(-torch.expm1(-dt))
    dout = torch.randn(batch, dim, L, generator=g)
    return u, delta, A, Bs, Cs, D, bias, dout


def _hip_fwd(u, delta, A, B, C, D, bias, softplus=True, variant=0, grad=False):
    from medmamba_amd.selective_scan_interface import SelectiveScanFn
    dev = _dev()
    xd = None
    mv = lambda t: None if t is None else t.to(dev)
    # keep B/C as views of one device buffer when they are views on the host
    if B.untyped_storage().data_ptr() == C.untyped_storage().data_ptr() and not B.is_contiguous():
        base = torch.empty(B.untyped_storage().size() // 4, dtype=torch.float32)
        base.set_(B.untyped_storage(), 0, (B.untyped_storage().size() // 4,))
        xd = base.to(dev)
        Bd = torch.as_strided(xd, B.shape, B.stride(), B.storage_offset())
        Cd = torch.as_strided(xd, C.shape, C.stride(), C.storage_offset())
    else:
        Bd, Cd = mv(B), mv(C)
    ins = [mv(u), mv(delta), mv(A), Bd, Cd, mv(D), mv(bias)]
    if grad:
        ins = [None if t is None else t.detach().requires_grad_() for t in ins]
    out = SelectiveScanFn.apply(*ins, softplus, variant)
    return out, ins


def _check_fwd(args, softplus=True, variant=0):
    from oracle.scan_ref import c_scan_fwd
    u, delta, A, B, C, D, bias, _ = args
    out, _ = _hip_fwd(u, delta, A, B, C, D, bias, softplus, variant)
    torch.cuda.synchronize()
    o64 = c_scan_fwd(u, delta, A, B, C, D, bias, softplus, f64=True)
    o32 = c_scan_fwd(u, delta, A, B, C, D, bias, softplus, f64=False)
    scale = np.abs(o64).max()
    err = np.abs(out.cpu().numpy() - o64).max()
    ref_err = np.abs(o32 - o64).max()
    assert np.isfinite(out.cpu().numpy()).all()
    assert err <= 2 * ref_err + FWD_RTOL * scale, (err, ref_err, scale)
    return err / scale


def _check_bwd(args, softplus=True, variant=0):
    from oracle.scan_ref import c_scan_bwd
    u, delta, A, B, C, D, bias, dout = args
    out, ins = _hip_fwd(u, delta, A, B, C, D, bias, softplus, variant, grad=True)
    out.backward(dout.to(out.device))
    torch.cuda.synchronize()
    r = c_scan_bwd(u, delta, A, B, C, D, bias, dout, softplus)
    names = ["du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias"]
    worst = {}
    for k, t in zip(names, ins):
        if t is None:
            continue
        got, want = t.grad.cpu().numpy(), r[k]
        assert got.shape == want.shape, k
        e = np.abs(got - want).max() / max(1.0, np.abs(want).max())
        worst[k] = e
        assert e <= BWD_RTOL, (k, e)
    return worst


@pytest.mark.parametrize("name", ["scan_small.npz", "scan_mid.npz", "scan_long.npz"])
def test_golden_forward_and_backward(name):
    fx = load_golden(name)
    R, N = int(fx["R"]), 16
    xd = torch.from_numpy(fx["x_dbl"])
    t = lambda k: torch.from_numpy(fx[k])
    B, C = xd[:, :, R:R + N], xd[:, :, R + N:]
    out, ins = _hip_fwd(t("u"), t("delta"), t("A"), B, C, t("D"), t("delta_bias"), True, 0, grad=True)
    scale = np.abs(fx["out"]).max()
    assert np.abs(out.detach().cpu().numpy() - fx["out"]).max() <= 5e-5 * scale     # golden is the fp32 loop
    out.backward(t("dout").to(out.device))
    want = dict(du=fx["du"], ddelta=fx["ddelta"], dA=fx["dA"], dB=fx["dx_dbl"][:, :, R:R + N],
                dC=fx["dx_dbl"][:, :, R + N:], dD=fx["dD"], ddelta_bias=fx["ddelta_bias"])
    for k, tin in zip(["du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias"], ins):
        e = np.abs(tin.grad.cpu().numpy() - want[k]).max() / max(1.0, np.abs(want[k]).max())
        assert e <= 3e-4, (k, e)


@pytest.mark.parametrize("variant", [1, 2, 4, 32])
@pytest.mark.parametrize("shape", [(2, 4, 8, 37), (1, 4, 24, 196), (2, 4, 96, 64), (1, 2, 5, 130), (3, 4, 32, 49),
                                   (1, 4, 16, 257), (1, 1, 16, 1), (2, 4, 20, 452), (1, 4, 50, 16)])
def test_forward_vs_oracle(shape, variant):
    """variant = states per lane of the general kernel (1, 2, 4); 32 = the workgroup-cooperative kernel (2 states per lane, B / C
    staged once per workgroup and read through DPP quad broadcasts) — vector path only: the shapes with L % 4 != 0 then take the
    general kernel's plan."""
    _check_fwd(_make(*shape, seed=sum(shape)), True, variant)


@pytest.mark.parametrize("fwd_variant", [32, 4])
@pytest.mark.parametrize("name", ["scan_mid.npz", "scan_long.npz"])
def test_forced_forward_kernels_golden_checkpoints_and_cross_scan(name, fwd_variant, monkeypatch):
    """The workgroup-cooperative (32) / the general 4-states-per-lane (4) forward kernel forced as the process-wide plan
    (MM_FWD_VARIANT): golden vectors forward AND backward (the backward kernel consumes the state checkpoints this forward
    wrote), and the shared-u / reversed-direction call of SS2D — whichever kernel the default plan picks, both stay covered."""
    from medmamba_amd import selective_scan_interface as ssi
    monkeypatch.setattr(ssi, "_FWD_VARIANT", fwd_variant)
    test_golden_forward_and_backward(name)
    test_cross_scan_matches_explicit_flips_and_oracle()
    test_forward_default_plan_real_stage_shapes()


def test_forward_default_plan_real_stage_shapes():
    # MedMamba-T/S per-stage (K*D, L) at batch 2: default variant planning
    for G, H, L in [(4, 96, 3136), (4, 192, 784), (4, 384, 196), (4, 768, 49)]:
        _check_fwd(_make(2, G, H, L, seed=L), True, 0)


def test_long_sequence_config5_shape():
    """BASELINE config 5 (MedMamba-B, 384^2): stage-1 sequence length L = 9216, forward and backward."""
    args = _make(1, 4, 16, 9216, R=4, seed=11)
    _check_fwd(args, True, 0)
    _check_bwd(args)


def test_forward_options():
    base = _make(2, 4, 16, 100, seed=5)
    u, delta, A, B, C, D, bias, dout = base
    _check_fwd((u, delta, A, B, C, None, bias, dout))            # D = None
    _check_fwd((u, delta, A, B, C, D, None, dout))               # no bias
    _check_fwd((u, delta.abs() * 0.1, A, B, C, D, bias.abs() * 0.01, dout), softplus=False)   # delta' must stay > 0
    _check_fwd(_make(2, 4, 16, 100, seed=6, contiguous_bc=True))
    # softplus threshold branch (x > 20) and very negative raw delta (log1p series branch)
    d2 = delta.clone(); d2[:, :, ::7] = 25.0; d2[:, :, 3::11] = -30.0
    _check_fwd((u, d2 * 1.0, A * 0.01, B, C, D, bias, dout))


@pytest.mark.parametrize("shape", [(2, 4, 8, 37), (1, 4, 24, 196), (2, 4, 96, 64), (1, 2, 5, 130), (3, 4, 32, 49),
                                   (1, 4, 200, 80)])
@pytest.mark.parametrize("bwd_variant", [0, 1 << 24, 1 << 25, (1 << 24) | (3 << 16), (1 << 24) | (12 << 16), (1 << 25) | (2 << 16),
                                         (1 << 25) | (1 << 16) | (3 << 8), (1 << 24) | (2 << 16) | (2 << 8), (1 << 25) | (1 << 16) | (1 << 8),
                                         (1 << 25) | (1 << 16) | (255 << 8)],
                         ids=["plan", "2states", "4states", "2states_3waves", "2states_12waves", "4states_2waves",
                              "4states_1wave_3passes", "2states_2waves_2passes", "4states_1wave_1pass", "4states_1wave_allpasses"])
def test_backward_vs_oracle(shape, bwd_variant, monkeypatch):
    """Both lane mappings of scan_bwd.hip (2 and 4 states per lane: variant bits 24 / 25), several workgroup widths (variant
    bits 16-23) and pass counts (bits 8-15: channel tiles a workgroup walks in turn; with every tile of a direction in ONE
    workgroup dB/dC are plain stores + read-modify-writes, with several workgroups per direction they are partial planes)."""
    from medmamba_amd import selective_scan_interface as ssi
    monkeypatch.setattr(ssi, "_BWD_VARIANT", bwd_variant)
    _check_bwd(_make(*shape, seed=1 + sum(shape)))


@pytest.mark.parametrize("bwd_variant", [1 << 24, 1 << 25])
def test_backward_options(bwd_variant, monkeypatch):
    from medmamba_amd import selective_scan_interface as ssi
    monkeypatch.setattr(ssi, "_BWD_VARIANT", bwd_variant)
    u, delta, A, B, C, D, bias, dout = _make(2, 4, 16, 100, seed=8)
    _check_bwd((u, delta, A, B, C, None, bias, dout))
    _check_bwd((u, delta, A, B, C, D, None, dout))
    _check_bwd((u, delta.abs() * 0.1, A, B, C, D, bias.abs() * 0.01, dout), softplus=False)


def test_full_size_properties_stage1():
    """BASELINE config-3 size (S, Bz=64, stage 1: 64 x 384 x 3136): properties instead of a full oracle run.
    (a) sampled rows against the fp64 oracle, (b) linearity in u: scan(a*u) == a*scan(u),
    (c) determinism of the forward (no atomics): two launches are bitwise equal."""
    from oracle.scan_ref import c_scan_fwd
    from medmamba_amd import selective_scan_fn
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(0)
    Bz, G, H, L, N, R = 64, 4, 96, 3136, 16, 3
    dim = G * H
    u = torch.randn(Bz, dim, L, device=dev, generator=g)
    delta = torch.randn(Bz, dim, L, device=dev, generator=g)
    A = -torch.exp(torch.randn(dim, N, device=dev, generator=g) * 0.5)
    x_dbl = torch.randn(Bz, G, R + 2 * N, L, device=dev, generator=g)
    Bs, Cs = x_dbl[:, :, R:R + N], x_dbl[:, :, R + N:]
    D = torch.randn(dim, device=dev, generator=g)
    bias = torch.randn(dim, device=dev, generator=g) - 4.0
    out = selective_scan_fn(u, delta, A, Bs, Cs, D, None, bias, True)
    out2 = selective_scan_fn(u, delta, A, Bs, Cs, D, None, bias, True)
    assert torch.equal(out, out2)
    out3 = selective_scan_fn(u * 2.0, delta, A, Bs, Cs, D, None, bias, True)
    assert torch.allclose(out3, out * 2.0, rtol=1e-5, atol=1e-5)
    for b in (0, 37, 63):
        for grp in (0, 3):
            sl = slice(grp * H + 5, grp * H + 9)
            o64 = c_scan_fwd(u[b:b + 1, sl].cpu(), delta[b:b + 1, sl].cpu(), A[sl].cpu(), Bs[b:b + 1, grp:grp + 1].cpu(),
                             Cs[b:b + 1, grp:grp + 1].cpu(), D[sl].cpu(), bias[sl].cpu(), True, f64=True)
            got = out[b:b + 1, sl].cpu().numpy()
            assert np.abs(got - o64).max() <= 5e-5 * max(1.0, np.abs(o64).max())


def test_cross_scan_matches_explicit_flips_and_oracle():
    """cross_scan_fn (shared u blocks + reversed directions, nothing flipped in memory) == the reference's
    materialised cross-scan (stack / flip) pushed through the oracle, forward and backward."""
    from oracle.scan_ref import c_scan_bwd, c_scan_fwd
    from medmamba_amd import cross_scan_fn
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    Bz, D, L, N = 2, 24, 100, 16
    u2 = torch.randn(Bz, 2, D, L, generator=g)
    delta = torch.randn(Bz, 4, D, L, generator=g)
    A = -torch.exp(torch.randn(4 * D, N, generator=g) * 0.5)
    Bm, Cm = torch.randn(Bz, 4, N, L, generator=g), torch.randn(Bz, 4, N, L, generator=g)
    Dp, bias = torch.randn(4 * D, generator=g), torch.randn(4 * D, generator=g) - 3
    dy2 = torch.randn(Bz, 2, D, L, generator=g)
    # oracle on explicitly time-ordered tensors: direction g = (block g//2, reversed if g odd)
    tm = lambda t, gi: t.flip(-1) if gi % 2 else t
    u4 = torch.stack([tm(u2[:, gi // 2], gi) for gi in range(4)], 1).reshape(Bz, 4 * D, L)
    d4 = torch.stack([tm(delta[:, gi], gi) for gi in range(4)], 1).reshape(Bz, 4 * D, L)
    B4 = torch.stack([tm(Bm[:, gi], gi) for gi in range(4)], 1)
    C4 = torch.stack([tm(Cm[:, gi], gi) for gi in range(4)], 1)
    o4 = torch.from_numpy(c_scan_fwd(u4, d4, A, B4, C4, Dp, bias, True, f64=True)).view(Bz, 4, D, L)
    want = torch.stack([o4[:, 0] + o4[:, 1].flip(-1), o4[:, 2] + o4[:, 3].flip(-1)], 1)
    ins = [t.to(dev).requires_grad_() for t in (u2.view(Bz, 2 * D, L), delta.view(Bz, 4 * D, L), A, Bm, Cm, Dp, bias)]
    y2 = cross_scan_fn(*ins)
    err = (y2.detach().cpu().double().view(Bz, 2, D, L) - want).abs().max().item()
    assert err <= 5e-5 * want.abs().max().item(), err
    y2.backward(dy2.view(Bz, 2 * D, L).to(dev))
    dout4 = torch.stack([tm(dy2[:, gi // 2], gi) for gi in range(4)], 1).reshape(Bz, 4 * D, L)
    r = c_scan_bwd(u4, d4, A, B4, C4, Dp, bias, dout4, True)
    du4 = torch.from_numpy(r["du"]).view(Bz, 4, D, L)
    want_du = torch.stack([du4[:, 0] + du4[:, 1].flip(-1), du4[:, 2] + du4[:, 3].flip(-1)], 1).view(Bz, 2 * D, L)
    unflip = lambda t: torch.stack([tm(t[:, gi], gi) for gi in range(4)], 1)
    wants = [want_du, unflip(torch.from_numpy(r["ddelta"]).view(Bz, 4, D, L)).reshape(Bz, 4 * D, L),
             torch.from_numpy(r["dA"]), unflip(torch.from_numpy(r["dB"])), unflip(torch.from_numpy(r["dC"])),
             torch.from_numpy(r["dD"]), torch.from_numpy(r["ddelta_bias"])]
    for name, t, w in zip(["du2", "ddelta", "dA", "dB", "dC", "dD", "dbias"], ins, wants):
        e = (t.grad.cpu().double() - w).abs().max().item() / max(1.0, w.abs().max().item())
        assert e <= BWD_RTOL, (name, e)


@pytest.mark.parametrize("bwd_variant", [1 << 24, 1 << 25], ids=["2states", "4states"])
def test_cross_scan_backward_in_both_lane_mappings(bwd_variant, monkeypatch):
    """Reversed directions / shared u blocks (mirrored LDS tiles) through both lane mappings of the backward kernel."""
    from medmamba_amd import selective_scan_interface as ssi
    monkeypatch.setattr(ssi, "_BWD_VARIANT", bwd_variant)
    test_cross_scan_matches_explicit_flips_and_oracle()


@pytest.mark.parametrize("L", [49, 130])
def test_cross_scan_unaligned_lengths(L):
    """L % 4 != 0 exercises the dword path of the reversed directions."""
    from oracle.scan_ref import c_scan_fwd
    from medmamba_amd import cross_scan_fn
    dev = _dev()
    g = torch.Generator().manual_seed(L)
    Bz, D, N = 1, 8, 16
    u2 = torch.randn(Bz, 2, D, L, generator=g); delta = torch.randn(Bz, 4, D, L, generator=g)
    A = -torch.exp(torch.randn(4 * D, N, generator=g) * 0.5)
    Bm, Cm = torch.randn(Bz, 4, N, L, generator=g), torch.randn(Bz, 4, N, L, generator=g)
    Dp, bias = torch.randn(4 * D, generator=g), torch.randn(4 * D, generator=g) - 3
    tm = lambda t, gi: t.flip(-1) if gi % 2 else t
    u4 = torch.stack([tm(u2[:, gi // 2], gi) for gi in range(4)], 1).reshape(Bz, 4 * D, L)
    d4 = torch.stack([tm(delta[:, gi], gi) for gi in range(4)], 1).reshape(Bz, 4 * D, L)
    B4 = torch.stack([tm(Bm[:, gi], gi) for gi in range(4)], 1); C4 = torch.stack([tm(Cm[:, gi], gi) for gi in range(4)], 1)
    o4 = torch.from_numpy(c_scan_fwd(u4, d4, A, B4, C4, Dp, bias, True, f64=True)).view(Bz, 4, D, L)
    want = torch.stack([o4[:, 0] + o4[:, 1].flip(-1), o4[:, 2] + o4[:, 3].flip(-1)], 1)
    with torch.no_grad():
        y2 = cross_scan_fn(*[t.to(dev) for t in (u2.view(Bz, 2 * D, L), delta.view(Bz, 4 * D, L), A, Bm, Cm, Dp, bias)])
    assert (y2.cpu().double().view(Bz, 2, D, L) - want).abs().max().item() <= 5e-5 * want.abs().max().item()


def _full_size_inputs(Bz, G, H, L, R, seed, channel_major):
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(seed)
    N, dim = 16, G * H
    mk = (lambda *shape: torch.randn(shape[1], shape[0], shape[2], device=dev, generator=g).permute(1, 0, 2)) if channel_major \
        else (lambda *shape: torch.randn(*shape, device=dev, generator=g))
    u, delta, dout = mk(Bz, dim, L), mk(Bz, dim, L), mk(Bz, dim, L)
    A = -torch.exp(torch.randn(dim, N, device=dev, generator=g) * 0.5)
    x_dbl = torch.randn(Bz, G, R + 2 * N, L, device=dev, generator=g)
    D = torch.randn(dim, device=dev, generator=g)
    bias = torch.randn(dim, device=dev, generator=g) - 4.0
    return u, delta, A, x_dbl[:, :, R:R + N], x_dbl[:, :, R + N:], D, bias, dout


def _run_bwd(args, channel_major, variant=0, monkeypatch=None):
    from medmamba_amd import selective_scan_interface as ssi
    u, delta, A, Bs, Cs, D, bias, dout = args
    if monkeypatch is not None:
        monkeypatch.setattr(ssi, "_BWD_VARIANT", variant)
    _, x_chk = ssi._launch_fwd(u, delta, A, Bs, Cs, D, bias, True, True)
    return ssi._launch_bwd(u, delta, A, Bs, Cs, D, bias, x_chk, dout, True, channel_major=channel_major)


@pytest.mark.parametrize("shape", [(64, 4, 96, 3136, 3, False), (64, 4, 384, 196, 12, True)],
                         ids=["stage1_batch_major", "stage3_channel_major"])
def test_full_size_backward_properties(shape, monkeypatch):
    """BASELINE config-3 sizes of the BACKWARD (S, Bz=64: stage 1 = 64 x 384 x 3136 batch-major planes, stage 3 =
    64 x 1536 x 196 channel-major planes): (a) du / ddelta of sampled rows and dB / dC of sampled (batch, direction) pairs
    against the fp64 oracle run on exactly that slice, (b) linearity in dout, (c) the forced 2-wave workgroups (several
    workgroups per direction -> atomic dB/dC path) agree with the default plan, (d) dA / dD / dbias against an oracle run
    over a full batch slice is too slow — their linearity and the sampled (batch, direction) slices cover them."""
    from oracle.scan_ref import c_scan_bwd
    Bz, G, H, L, R, cm = shape
    args = _full_size_inputs(Bz, G, H, L, R, seed=L, channel_major=cm)
    u, delta, A, Bs, Cs, D, bias, dout = args
    du, ddelta, dA, dB, dC, dD, dbias = _run_bwd(args, cm, 0, monkeypatch)
    torch.cuda.synchronize()
    for t in (du, ddelta, dA, dB, dC, dD, dbias):
        assert torch.isfinite(t).all()
    # (a) sampled (batch, direction) pairs: the whole direction of one image through the oracle
    for b, grp in ((0, 0), (Bz - 1, G - 1), (Bz // 2, 1)):
        sl = slice(grp * H, (grp + 1) * H)
        c = lambda t: t.contiguous().cpu()
        r = c_scan_bwd(c(u[b:b + 1, sl]), c(delta[b:b + 1, sl]), c(A[sl]), c(Bs[b:b + 1, grp:grp + 1]), c(Cs[b:b + 1, grp:grp + 1]),
                       c(D[sl]), c(bias[sl]), c(dout[b:b + 1, sl]), True)
        for name, got, want in (("du", du[b:b + 1, sl], r["du"]), ("ddelta", ddelta[b:b + 1, sl], r["ddelta"]),
                                ("dB", dB[b:b + 1, grp:grp + 1], r["dB"]), ("dC", dC[b:b + 1, grp:grp + 1], r["dC"])):
            e = np.abs(got.cpu().numpy() - want).max() / max(1.0, np.abs(want).max())
            assert e <= BWD_RTOL, (name, b, grp, e)
    # (b) linearity in dout (every output of the backward is linear in it)
    r2 = _run_bwd((u, delta, A, Bs, Cs, D, bias, dout * 2.0), cm, 0, monkeypatch)
    for name, a1, a2 in zip(["du", "ddelta", "dA", "dB", "dC", "dD", "dbias"], (du, ddelta, dA, dB, dC, dD, dbias), r2):
        scale = max(1.0, float(a1.abs().max()))
        assert float((a2 - 2.0 * a1).abs().max()) <= 2e-4 * scale, name
    del r2
    # (c) forced small workgroups: several workgroups share a direction and add dB/dC with atomics
    r3 = _run_bwd(args, cm, 2 << 16, monkeypatch)
    for name, a1, a3 in zip(["du", "ddelta", "dA", "dB", "dC", "dD", "dbias"], (du, ddelta, dA, dB, dC, dD, dbias), r3):
        scale = max(1.0, float(a1.abs().max()))
        assert float((a3 - a1).abs().max()) <= 2e-4 * scale, name
    del r3
    # (e) the other lane mapping (the plan takes 2 states per lane at stage 1, 4 at stage 3)
    for force in (1 << 24, 1 << 25):
        r4 = _run_bwd(args, cm, force, monkeypatch)
        for name, a1, a4 in zip(["du", "ddelta", "dA", "dB", "dC", "dD", "dbias"], (du, ddelta, dA, dB, dC, dD, dbias), r4):
            scale = max(1.0, float(a1.abs().max()))
            assert float((a4 - a1).abs().max()) <= 2e-4 * scale, (name, force)
        del r4


@pytest.mark.parametrize("bwd_variant", [0, (1 << 25) | (1 << 16) | (2 << 8), (1 << 24) | (2 << 16) | (1 << 8)],
                         ids=["plan", "4states_1wave_2passes", "2states_2waves_1pass"])
def test_backward_is_bitwise_reproducible(bwd_variant, monkeypatch):
    """VERDICT r2 item 7 / reference train.py:21-29 (deterministic training): the backward uses no atomics — dA / dD / dbias
    through per-batch-item partial buffers, dB / dC in place or through per-workgroup partial planes, all summed in a fixed
    order — so two runs give the same bits, whatever the plan (several workgroups per direction, several passes)."""
    from medmamba_amd import selective_scan_interface as ssi
    monkeypatch.setattr(ssi, "_BWD_VARIANT", bwd_variant)
    dev = _dev()
    args = [t.to(dev) for t in _make(3, 4, 200, 196, seed=21, contiguous_bc=True)]
    u, delta, A, Bs, Cs, D, bias, dout = args
    _, x_chk = ssi._launch_fwd(u, delta, A, Bs, Cs, D, bias, True, True)
    plan = _plan_of(u, delta, A, Bs, Cs, backward=True, variant=bwd_variant)
    if bwd_variant:
        assert plan["W"] > 1 and (plan["passes"] > 1 or (bwd_variant >> 8) & 0xff == 1), plan
    runs = []
    for _ in range(3):
        junk = torch.randn(1 << 20, device=dev)                      # move the allocator / the scheduler a little
        r = ssi._launch_bwd(u, delta, A, Bs, Cs, D, bias, x_chk, dout, True)
        torch.cuda.synchronize()
        runs.append([t.clone() for t in r])
        del junk
    for r in runs[1:]:
        for name, a, b in zip(["du", "ddelta", "dA", "dB", "dC", "dD", "dbias"], runs[0], r):
            assert torch.equal(a, b), name


def _plan_of(u, delta, A, Bs, Cs, backward, variant=0, det=True):
    import ctypes
    from medmamba_amd import _lib
    from medmamba_amd import selective_scan_interface as ssi
    a = _lib.ScanArgs()
    ssi._fill_common(a, u, delta, A, Bs, Cs, None, None, True)
    a.variant = variant
    if det:
        a.dpar_sb, a.dBC_sc = 1, 1
    out = (ctypes.c_int32 * 8)()
    assert _lib.lib().mm_scan_plan(a, int(backward), out) == 0
    return dict(ns=out[0], waves=out[1], blocks=out[2], vec=out[3], lean=out[4], det=out[5], W=out[6], passes=out[7])


def test_backward_plans_fill_the_chip():
    """The launch plans of the backward for the benchmark shapes: one 12-wave workgroup per direction at the
    56x56 stage of S (2 states per lane), 512 single-pass 4-wave workgroups elsewhere (3 / 6 / 12 per direction at the 28x28 / 14x14 / 7x7 stages: dB / dC partial planes)."""
    dev = _dev()
    mk = lambda b, d, L: torch.empty((b, d, L), device=dev)
    for (b, H, L), want in {(64, 96, 3136): dict(ns=2, waves=12, blocks=256, W=1, passes=1),
                            (64, 192, 784): dict(ns=4, waves=4, blocks=768, W=3, passes=1),
                            (64, 384, 196): dict(ns=4, waves=4, blocks=1536, W=6, passes=1),
                            (64, 768, 49): dict(ns=4, waves=4, blocks=3072, W=12, passes=1),
                            (32, 128, 9216): dict(ns=2, waves=8, blocks=256, W=2, passes=1)}.items():
        u = mk(b, 4 * H, L)
        xd = torch.empty((b, 4, 35, L), device=dev)
        p = _plan_of(u, u, torch.empty((4 * H, 16), device=dev), xd[:, :, 3:19], xd[:, :, 19:], backward=True)
        for k, v in want.items():
            assert p[k] == v, ((b, H, L), p)
        assert p["det"] == 1
