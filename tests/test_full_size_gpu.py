"""GPU: BASELINE configs 3 and 5 at FULL size as a MODEL.

Config 3: MedMamba-S, 64 x 3 x 224 x 224, forward + CE + backward (train.py:277-288 without the optimizer step).
Config 5: MedMamba-B, 32 x 3 x 384 x 384 (L = 9216 at the first stage).

The CPU oracle cannot run 64 images in seconds, so the checks are those that do not need it at full size:
  * eval mode makes images independent (BatchNorm uses running statistics): two images of the full batch are compared with the
    oracle model (oracle.model_ref.vssm_forward + the C scan) run on just those two — this pins the full-batch launch plans
    (wave counts, channel-major planes, strip counts) to the reference arithmetic;
  * train mode: the loss and every parameter gradient agree between the two-stream and the single-stream schedule and between
    batch-major and channel-major plane storage (independent code paths through GEMMs, strides and kernels), finite everywhere;
  * the scan launch plans taken at these sizes are the ones DESIGN.md describes (mm_scan_plan).
DropPath is off for parity (it draws from the device RNG).
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle_logits(net, x, pick, depths):
    from oracle import model_ref as R
    from oracle.scan_ref import c_selective_scan_fn
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        return R.vssm_forward(sd, x[pick].cpu(), depths, c_selective_scan_fn, training=False).numpy()


def _oracle_train_pass(net, x, y, depths):
    """Loss and every parameter gradient of ONE training-mode pass (BatchNorm batch statistics over exactly this batch,
    MedMamba.py:338-346; forward + CE + backward = train.py:277-286) through the CPU oracle: oracle.model_ref (pinned by the
    reference-generated fixtures) + the C scan, on the host cores."""
    import os
    from oracle import model_ref as R
    from oracle.scan_ref import _lib as _olib, build_c_oracle, c_selective_scan_fn
    build_c_oracle()
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    torch.set_num_threads(cores)
    _olib().oracle_set_threads(cores)
    p = {k: (v.detach().cpu().clone().requires_grad_() if v.dtype.is_floating_point and "running" not in k else v.detach().cpu().clone())
         for k, v in net.state_dict().items()}
    loss = torch.nn.functional.cross_entropy(R.vssm_forward(p, x.cpu(), depths, c_selective_scan_fn, training=True), y.cpu())
    loss.backward()
    names = {k for k, _ in net.named_parameters()}
    return float(loss.detach()), {k: v.grad for k, v in p.items() if k in names}


def _train_pass(net, x, y):
    net.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(net(x), y)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), {k: p.grad.detach().clone() for k, p in net.named_parameters()}


def _compare_grads(g0, g1, what, l2_tol=3e-3, max_tol=2e-2):
    worst = (0.0, None)
    seen = []
    for k in g0:
        # the bias of a conv that feeds a BatchNorm has an exactly-zero true gradient (the batch mean removes any constant
        # shift, MedMamba.py:339-343): what is computed there is rounding noise of either schedule, not a quantity to compare
        if k.endswith("conv33conv33conv11.1.bias") or k.endswith("conv33conv33conv11.4.bias"):
            continue
        a, b = g0[k].double(), g1[k].double()
        assert torch.isfinite(b).all(), (what, k)
        den = float(a.norm())
        if den < 1e-12:
            continue
        l2 = float((a - b).norm()) / den
        mx = float((a - b).abs().max()) / max(1e-6, float(a.abs().max()))
        worst = max(worst, (l2, k), key=lambda t: t[0])
        seen.append((l2, mx, k))
    seen.sort(reverse=True)
    print(f"\n[{what}] largest gradient deviations (l2 rel, max rel, tensor): " + "; ".join(f"{a:.2e} {b:.2e} {k}" for a, b, k in seen[:5]))
    for l2, mx, k in seen:
        assert l2 <= l2_tol and mx <= max_tol, (what, k, l2, mx)
    return worst


def _plan(batch, G, H, L, backward, cm=False):
    """Launch plan the library takes for a contiguous (or channel-major) scan of this shape: dict(ns, waves, blocks, vec, lean)."""
    from medmamba_amd import _lib
    a = _lib.ScanArgs()
    a.batch, a.dim, a.L, a.N, a.G, a.delta_softplus = batch, G * H, L, 16, G, 1
    if cm:
        a.u_sb, a.u_sd, a.delta_sb, a.delta_sd = L, batch * L, L, batch * L
        a.dout_sb, a.dud_sb, a.o_sd = L, L, batch * L
    else:
        a.u_sb, a.u_sd, a.delta_sb, a.delta_sd = 2 * H * L, L, G * H * L, L
    a.B_sb, a.B_sg, a.B_sn = G * 35 * L, 35 * L, L
    a.C_sb, a.C_sg, a.C_sn = G * 35 * L, 35 * L, L
    a.u_groups, a.u_map, a.rev_mask = 2, 0x1100, 0b1010
    out = (ctypes.c_int32 * 8)()
    rc = _lib.lib().mm_scan_plan(a, int(backward), out)
    assert rc == 0, rc
    return dict(ns=out[0], waves=out[1], blocks=out[2], vec=out[3], lean=out[4])


def test_config3_S_batch64_full_size(monkeypatch):
    from medmamba_amd import modules, ops
    from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
    cfg = MEDMAMBA_CONFIGS["S"]
    torch.manual_seed(42)
    net = VSSM(num_classes=6, drop_path_rate=0.0, **cfg).to(DEV)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 3, 224, 224, generator=g)
    y = torch.randint(0, 6, (64,), generator=g).to(DEV)
    xd = x.to(DEV)
    # (i) eval logits of two images of the full batch vs the oracle model on those two
    net.eval()
    with torch.no_grad():
        got = net(xd).cpu().numpy()
    pick = [3, 60]
    want = _oracle_logits(net, x, pick, cfg["depths"])
    assert np.isfinite(got).all()
    assert np.abs(got[pick] - want).max() <= 1e-3 * max(1.0, np.abs(want).max()), np.abs(got[pick] - want).max()
    # (ii) train mode at full size: schedules and layouts agree.  Under torch.backends.cudnn.deterministic — the mode the reference
    # trains in (train.py:28-29) — every kernel of the step is reproducible, so the two-stream schedule must give the SAME BITS as
    # the single-stream one, pass after pass: a missed cross-stream dependency or an early buffer reuse cannot hide in a
    # tolerance.  (Outside that mode MIOpen's forward pick for the dense convolutions is not reproducible on every box of the pool
    # — ~1e-6 per output, which flips a few ReLU masks: two identical passes then differ by 2e-5 of a gradient's norm most of the
    # time and by 1e-3 ... 1e-2 now and then, tools/run_to_run_noise.py, DESIGN.md §2 — which is why this comparison used to flake
    # at any fixed tolerance.)
    net.train()
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)
    # single-stream first: the first pass over a conv shape runs on one stream anyway (MIOpen's solver search, modules.py)
    monkeypatch.setattr(modules, "_TWO_STREAMS", False)
    l1, g1 = _train_pass(net, xd, y)
    monkeypatch.setattr(modules, "_TWO_STREAMS", True)
    l0, g0 = _train_pass(net, xd, y)
    assert np.isfinite(l0)
    assert sum(1 for k in modules._CONV_WARM if k[2] is True) >= 4        # the four stage shapes were marked by the first backward
    l0b, g0b = _train_pass(net, xd, y)
    assert l1 == l0 == l0b, (l1, l0, l0b)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), ("two-stream vs single-stream", k)
        assert torch.equal(g0[k], g0b[k]), ("two-stream, pass after pass", k)
    del g1, g0b
    for layout in ("bm", "cm"):
        monkeypatch.setattr(ops, "_LAYOUT", layout)
        l2, g2 = _train_pass(net, xd, y)
        assert abs(l2 - l0) <= 5e-6 * abs(l0), (layout, l0, l2)
        # the layouts differ in every GEMM's shape and summation order (batched vs one GEMM over batch*L columns); through 14
        # blocks with BatchNorm batch statistics and ReLU masks that is a few 1e-3 of a gradient's norm at this size
        # (7.2e-3 / 1.7e-2 measured; the oracle comparison below supports the same bound: l2 1.5e-2, max 5e-2)
        _compare_grads(g0, g2, f"auto vs {layout}", l2_tol=1.5e-2, max_tol=5e-2)
        del g2
    monkeypatch.setattr(ops, "_LAYOUT", "auto")
    # the default mode (what bench.py measures: MIOpen free to pick its atomics-based solvers): same loss, gradients within that
    # mode's own noise of the reproducible ones
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", False)
    l3, g3 = _train_pass(net, xd, y)
    assert abs(l3 - l0) <= 2e-5 * abs(l0), (l0, l3)
    _compare_grads(g0, g3, "default mode vs cudnn.deterministic", l2_tol=5e-2, max_tol=2.5e-1)
    del g3
    # (iii) the launch plans of this configuration
    p1 = _plan(64, 4, 96, 3136, backward=True)
    assert p1["vec"] == 1 and p1["ns"] == 2 and p1["waves"] == 12 and p1["blocks"] == 256, p1      # one workgroup per direction
    p3 = _plan(64, 4, 384, 196, backward=True, cm=True)
    assert p3["vec"] == 1 and p3["ns"] == 4, p3
    p4 = _plan(64, 4, 768, 49, backward=True, cm=True)
    assert p4["ns"] == 4, p4
    f1 = _plan(64, 4, 96, 3136, backward=False)
    assert f1["vec"] == 1 and f1["ns"] in (2, 4), f1


def test_config3_training_step_loss_and_gradients_match_the_oracle(monkeypatch):
    """the full-size TRAINING pass against the oracle itself — all 64 images of config 3 in train mode (BatchNorm
    statistics over the same 64 images), loss and every parameter gradient, DropPath off.  The oracle needs ~20-40 s of 16 host
    cores for this once (bench.py's cpu_baseline runs 32 images in ~7 s)."""
    from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
    cfg = MEDMAMBA_CONFIGS["S"]
    torch.manual_seed(42)
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)      # the reference's mode; reproducible on every box
    net = VSSM(num_classes=6, drop_path_rate=0.0, **cfg).to(DEV).train()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 3, 224, 224, generator=g)
    y = torch.randint(0, 6, (64,), generator=g)
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}        # running statistics before the pass
    want_loss, want = _oracle_train_pass(net, x, y, cfg["depths"])
    net.load_state_dict(sd0)
    loss, got = _train_pass(net, x.to(DEV), y.to(DEV))
    assert abs(loss - want_loss) <= 1e-5 * abs(want_loss), (loss, want_loss)
    assert set(got) == set(want)
    # tolerance: fp32 against fp32 through 14 blocks with BatchNorm batch statistics and ReLU masks.  Measured on MI355X: the worst
    # tensors (x_proj / conv-branch weights of the first stages, patch embedding) deviate by 5e-3 ... 8e-3 of their norm — the SAME
    # size as batch-major vs channel-major storage of this library against itself (5.9e-3 ... 7.2e-3 below), i.e. summation-order
    # noise of this depth, not a modelling difference (a wrong term shows up as >= 1e-1); the loss agrees to 1e-5
    worst = _compare_grads({k: v.to(DEV) for k, v in want.items()}, got, "HIP step vs oracle, config 3", l2_tol=1.5e-2, max_tol=5e-2)
    print("config 3 worst gradient deviation from the oracle (l2 rel, tensor):", worst)


def test_config5_training_step_matches_the_oracle_at_8_images(monkeypatch):
    """Config 5's shapes (MedMamba-B, 384 x 384: L = 9216 / 2304 / 576 / 144, 128 ... 1024 channels) at a batch the host can afford:
    8 images in train mode against the oracle — loss and every parameter gradient.  The launch plans at L = 9216 are those of
    the 32-image configuration except for the wave count."""
    from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
    cfg = MEDMAMBA_CONFIGS["B"]
    torch.manual_seed(42)
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)      # the reference's mode; reproducible on every box
    net = VSSM(num_classes=6, drop_path_rate=0.0, **cfg).to(DEV).train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 3, 384, 384, generator=g)
    y = torch.randint(0, 6, (8,), generator=g)
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    want_loss, want = _oracle_train_pass(net, x, y, cfg["depths"])
    net.load_state_dict(sd0)
    loss, got = _train_pass(net, x.to(DEV), y.to(DEV))
    assert abs(loss - want_loss) <= 1e-5 * abs(want_loss), (loss, want_loss)
    # (8 images: BatchNorm statistics over 8 x 12 x 12 positions at the last stage, gradients of smaller norm: 1.1e-2 of a tensor's norm
    # measured for the worst tensor; single elements of the last stage's conv weights — ReLU masks that flip under rounding — were
    # seen 0.14 of the tensor's largest element apart while the tensor's l2 deviation stayed at 8e-3)
    worst = _compare_grads({k: v.to(DEV) for k, v in want.items()}, got, "HIP step vs oracle, config 5 shapes", l2_tol=2e-2, max_tol=2.5e-1)
    print("config 5 (8 images) worst gradient deviation from the oracle (l2 rel, tensor):", worst)


def test_config5_B_batch32_384_full_size():
    from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
    cfg = MEDMAMBA_CONFIGS["B"]
    torch.manual_seed(42)
    net = VSSM(num_classes=6, drop_path_rate=0.0, **cfg).to(DEV)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(32, 3, 384, 384, generator=g)
    y = torch.randint(0, 6, (32,), generator=g).to(DEV)
    xd = x.to(DEV)
    net.eval()
    with torch.no_grad():
        got = net(xd).cpu().numpy()
    pick = [0, 17]
    want = _oracle_logits(net, x, pick, cfg["depths"])
    assert np.isfinite(got).all()
    assert np.abs(got[pick] - want).max() <= 1e-3 * max(1.0, np.abs(want).max()), np.abs(got[pick] - want).max()
    net.train()
    loss, grads = _train_pass(net, xd, y)
    assert np.isfinite(loss)
    for k, v in grads.items():
        assert torch.isfinite(v).all(), k
        assert float(v.abs().max()) > 0.0, k
    # the long-sequence stage (L = 9216, 128 channels per direction): half-width waves (2 states per lane)
    p1 = _plan(32, 4, 128, 9216, backward=True)
    assert p1["vec"] == 1 and p1["ns"] == 2, p1
