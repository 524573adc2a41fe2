"""GPU: the nn.Module surface running on the HIP kernels vs the committed reference fixtures
(produced by executing the real reference; tools/gen_golden.py) — forward and backward."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, split_sd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, want, rtol, what=""):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    err = np.abs(got - want).max()
    assert err <= rtol * max(1.0, np.abs(want).max()), (what, err)


@pytest.mark.parametrize("name", ["ss2d_d8.npz", "ss2d_d48.npz"])
def test_ss2d_forward_backward(name):
    from medmamba_amd.modules import SS2D
    fx = load_golden(name)
    m = SS2D(d_model=fx["x"].shape[-1])
    m.load_state_dict(split_sd(fx))
    m.to(DEV)
    x = torch.from_numpy(fx["x"]).to(DEV).requires_grad_()
    y = m(x)
    _close(y, fx["y"], 5e-5, "y")
    y.backward(torch.from_numpy(fx["dy"]).to(DEV))
    _close(x.grad, fx["dx"], 3e-4, "dx")
    for k, p in m.named_parameters():
        _close(p.grad, fx["grad/" + k], 5e-4, k)


def test_block_eval_and_train():
    from medmamba_amd.modules import SS_Conv_SSM
    fx = load_golden("block_c16.npz")
    blk = SS_Conv_SSM(hidden_dim=16, drop_path=0.0, norm_layer=torch.nn.LayerNorm)
    blk.load_state_dict(split_sd(fx))
    blk.to(DEV)
    x = torch.from_numpy(fx["x"]).to(DEV)
    blk.eval()
    _close(blk(x), fx["y_eval"], 5e-5)
    blk.train()
    xt = x.clone().requires_grad_()
    y = blk(xt)
    _close(y, fx["y_train"], 1e-4)
    y.backward(torch.from_numpy(fx["dy"]).to(DEV))
    _close(xt.grad, fx["dx"], 5e-4, "dx")
    for k, p in blk.named_parameters():
        _close(p.grad, fx["grad/" + k], 2e-3, k)


def test_tiny_vssm_logits_loss_and_grads():
    from medmamba_amd.modules import VSSM
    fx = load_golden("vssm_tiny.npz")
    net = VSSM(num_classes=3, depths=[int(v) for v in fx["depths"]], dims=[int(v) for v in fx["dims"]],
               drop_path_rate=0.0)
    net.load_state_dict(split_sd(fx))
    net.to(DEV)
    x = torch.from_numpy(fx["x"]).to(DEV)
    net.eval()
    _close(net(x), fx["logits_eval"], 1e-4, "logits_eval")
    net.train()
    logits = net(x)
    _close(logits, fx["logits_train"], 2e-4, "logits_train")
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(fx["labels"]).to(DEV))
    assert abs(float(loss) - float(fx["loss"])) <= 1e-4
    loss.backward()
    worst = 0.0
    for k, p in net.named_parameters():
        w = fx["grad/" + k]
        worst = max(worst, np.abs(p.grad.cpu().numpy() - w).max() / max(1e-3, np.abs(w).max()))
    assert worst <= 5e-3, worst      # fp32 training-mode grads through 4 blocks incl. BatchNorm batch statistics


@pytest.mark.parametrize("size", ["T", "S", "Te", "B"])
def test_seed_kat_full_model_logits(size):
    """BASELINE config 1 (T, S at 224x224) and the config-5 model (B at 384x384: 96x96 planes, D up to 1024, R up to 32):
    seed 42 -> same init -> logits of the reference (CPU, restated scan)."""
    from medmamba_amd.modules import VSSM
    kat = json.load(open(os.path.join(GOLDEN, "kat_seed42.json")))[size]
    torch.manual_seed(42)
    net = VSSM(num_classes=6, depths=kat["depths"], dims=kat["dims"]).eval()
    x = torch.randn(1, 3, kat["res"], kat["res"])
    with torch.no_grad():
        logits = net.to(DEV)(x.to(DEV))
    _close(logits[0], np.array(kat["logits"]), 1e-3, "logits")     # logits atol 1e-3


def test_two_stream_blocks_match_single_stream(monkeypatch):
    """The conv branch runs on a side HIP stream (modules.SS_Conv_SSM.forward): same loss and grads as the
    single-stream schedule, run after run with the allocator churned in between (a missed cross-stream dependency or
    an early buffer reuse shows here).  Run under cudnn.deterministic (the reference's mode, train.py:28-29), where every kernel of
    the step is reproducible: the schedules must agree BIT FOR BIT (outside that mode MIOpen's picks add run-to-run noise of 1e-5 ...
    1e-2 of a gradient's norm on this stack — DESIGN.md §2 — and a tolerance would have to be that loose)."""
    from medmamba_amd import modules
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)
    torch.manual_seed(3)
    net = modules.VSSM(num_classes=5, depths=[2, 2, 2, 2], dims=[32, 64, 128, 256], drop_path_rate=0.0).to(DEV).train()
    x = torch.randn(8, 3, 128, 128, device=DEV)
    y = torch.randint(0, 5, (8,), device=DEV)

    def run(two, it):
        monkeypatch.setattr(modules, "_TWO_STREAMS", two)
        monkeypatch.setattr(modules, "_LATE_SIDE_MIN_L", 0 if it % 2 else 1 << 30)   # late / early start of the side stream
        net.zero_grad(set_to_none=True)
        junk = [torch.empty(1 << (16 + (it + j) % 6), device=DEV).normal_() for j in range(4)]   # churn the allocator
        loss = torch.nn.functional.cross_entropy(net(x), y)
        loss.backward()
        del junk
        return float(loss.detach()), {k: p.grad.clone() for k, p in net.named_parameters()}

    l0, g0 = run(False, 0)
    for it, two in enumerate([True, True, False, True, True, True], start=1):
        l, g = run(two, it)
        assert l == l0, (it, l, l0)
        for k in g0:
            assert torch.equal(g0[k], g[k]), (it, two, k)


@pytest.mark.parametrize("layout", ["bm", "cm"])
def test_block_and_tiny_model_in_both_plane_layouts(layout, monkeypatch):
    """SS_Conv_SSM and the tiny VSSM, fwd + bwd, with the SS2D planes forced batch-major / channel-major
    (ops.channel_major) against the committed reference fixtures — the storage layout must not change any value."""
    from medmamba_amd import modules, ops
    monkeypatch.setattr(ops, "_LAYOUT", layout)
    fx = load_golden("block_c16.npz")
    blk = modules.SS_Conv_SSM(hidden_dim=16, drop_path=0.0, norm_layer=torch.nn.LayerNorm)
    blk.load_state_dict(split_sd(fx))
    blk.to(DEV).train()
    x = torch.from_numpy(fx["x"]).to(DEV).requires_grad_()
    y = blk(x)
    _close(y, fx["y_train"], 1e-4, "y_train")
    y.backward(torch.from_numpy(fx["dy"]).to(DEV))
    _close(x.grad, fx["dx"], 5e-4, "dx")
    for k, p in blk.named_parameters():
        _close(p.grad, fx["grad/" + k], 2e-3, k)
    fx = load_golden("vssm_tiny.npz")
    net = modules.VSSM(num_classes=3, depths=[int(v) for v in fx["depths"]], dims=[int(v) for v in fx["dims"]],
                       drop_path_rate=0.0)
    net.load_state_dict(split_sd(fx))
    net.to(DEV).train()
    logits = net(torch.from_numpy(fx["x"]).to(DEV))
    _close(logits, fx["logits_train"], 2e-4, "logits_train")
    torch.nn.functional.cross_entropy(logits, torch.from_numpy(fx["labels"]).to(DEV)).backward()
    worst = max(np.abs(p.grad.cpu().numpy() - fx["grad/" + k]).max() / max(1e-3, np.abs(fx["grad/" + k]).max())
                for k, p in net.named_parameters())
    assert worst <= 5e-3, worst


def test_hooked_block_runs_module_by_module_on_hip():
    """A forward hook on `conv33conv33conv11[-2]` (Grad-CAM, test.py:101-108) switches the block to the module-by-module
    path (HIP scan operator inside); same values as the fused path and the reference fixture."""
    from medmamba_amd import modules
    fx = load_golden("block_c16.npz")
    blk = modules.SS_Conv_SSM(hidden_dim=16, drop_path=0.0, norm_layer=torch.nn.LayerNorm)
    blk.load_state_dict(split_sd(fx))
    blk.to(DEV).train()
    seen = []
    h = blk.conv33conv33conv11[-2].register_forward_hook(lambda m, i, o: seen.append(o.detach()))
    x = torch.from_numpy(fx["x"]).to(DEV).requires_grad_()
    y = blk(x)
    y.backward(torch.from_numpy(fx["dy"]).to(DEV))
    assert len(seen) == 1 and seen[0].shape[1] == 8
    _close(y, fx["y_train"], 1e-4, "y_train")
    _close(x.grad, fx["dx"], 5e-4, "dx")
    for k, p in blk.named_parameters():
        _close(p.grad, fx["grad/" + k], 2e-3, k)
    h.remove()


def test_tuned_gemm_selection_keeps_the_numbers():
    """medmamba_amd.tuning: recorded rocBLAS / hipBLASLt solutions for the projection GEMMs; same values as the default
    heuristic's kernels (fp32 summation order aside) on a shape from the table."""
    from medmamba_amd.tuning import enable_tuned_gemms
    g = torch.Generator(device=DEV).manual_seed(0)
    w = torch.randn(96, 3, device=DEV, generator=g).unsqueeze(0).expand(256, -1, -1)
    x = torch.randn(256, 3, 3136, device=DEV, generator=g)
    ref = torch.bmm(w, x)
    try:
        assert enable_tuned_gemms() is not None
        assert torch.cuda.tunable.is_enabled() and not torch.cuda.tunable.tuning_is_enabled()
        got = torch.bmm(w, x)
    finally:
        torch.cuda.tunable.enable(False)
    assert (got - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


def test_odd_and_non_square_images_same_in_both_layouts(monkeypatch, capsys):
    """72 x 40 input: stage grids 18x10, 9x5, 4x2 (PatchMerging crops the odd 9x5, MedMamba.py:97-111), 2x1 — non-square,
    odd, unaligned (L % 4 != 0) planes through every kernel; batch-major and channel-major storage must agree, and the
    single-stream schedule with them."""
    from medmamba_amd import modules, ops
    torch.manual_seed(11)
    net = modules.VSSM(num_classes=4, depths=[1, 1, 1, 1], dims=[16, 32, 64, 128], drop_path_rate=0.0).to(DEV).train()
    x = torch.randn(3, 3, 72, 40, device=DEV)
    y = torch.randint(0, 4, (3,), device=DEV)

    def run(layout, two):
        monkeypatch.setattr(ops, "_LAYOUT", layout)
        monkeypatch.setattr(modules, "_TWO_STREAMS", two)
        net.zero_grad(set_to_none=True)
        logits = net(x)
        torch.nn.functional.cross_entropy(logits, y).backward()
        return logits.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}

    l0, g0 = run("bm", False)
    for layout, two in (("cm", False), ("auto", True), ("cm", True)):
        l1, g1 = run(layout, two)
        assert (l1 - l0).abs().max().item() <= 1e-4 * max(1.0, l0.abs().max().item()), layout
        for k in g0:
            scale = max(1e-4, g0[k].abs().max().item())
            assert (g1[k] - g0[k]).abs().max().item() <= 2e-3 * scale, (layout, two, k)
    capsys.readouterr()         # the crop warning of PatchMerging2D


def test_drop_path_factors_are_drawn_once_per_step(monkeypatch):
    """Training mode: VSSM draws every block's DropPath factor (mask / keep_prob, MedMamba.py:335, 353) in one go; the blocks
    consume them through shuffle_residual and none falls back to its own draw."""
    from medmamba_amd import modules
    torch.manual_seed(2)
    net = modules.VSSM(num_classes=3, depths=[1, 1, 2, 1], dims=[16, 32, 64, 128], drop_path_rate=0.5).to(DEV).train()
    x = torch.randn(16, 3, 32, 32, device=DEV)
    seen = {}
    orig = modules.shuffle_residual

    def spy(left, ssm, inp, channel_first=False, ssm_scale=None, left_relu=False, left_bias=None):
        seen[len(seen)] = None if ssm_scale is None else ssm_scale.detach().clone()
        return orig(left, ssm, inp, channel_first=channel_first, ssm_scale=ssm_scale, left_relu=left_relu, left_bias=left_bias)

    monkeypatch.setattr(modules, "shuffle_residual", spy)
    def no_own_draw(self, x):
        assert self.drop_prob == 0.0, "per-block draw"
        return None

    monkeypatch.setattr(modules.DropPath, "factor", no_own_draw)
    net(x).sum().backward()
    probs = [b.drop_path.drop_prob for layer in net.layers for b in layer.blocks]
    assert len(seen) == 5 and seen[0] is None and probs[0] == 0.0          # first block: drop_prob 0 -> no factor
    for i in range(1, 5):
        keep = 1.0 - probs[i]
        f = seen[i].cpu()
        assert f.shape == (16,) and bool(((f == 0) | ((f - 1.0 / keep).abs() < 1e-6)).all())
    assert all(getattr(b, "_dp_factor", None) is None for layer in net.layers for b in layer.blocks)
    assert any(bool((seen[i] == 0).any()) for i in range(1, 5)) and any(bool((seen[i] != 0).any()) for i in range(1, 5))


def test_config2_S_batch32_inference_vs_oracle_slice():
    """BASELINE config 2: MedMamba-S, 32 x 3 x 224 x 224, forward only (no_grad, eval: the x_chk = NULL / need_grad = False
    kernels).  eval-mode BatchNorm makes images independent, so two images of the batch are checked against the CPU oracle
    model (oracle.model_ref.vssm_forward with the C scan) run on just those two."""
    from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
    from oracle import model_ref as R
    from oracle.scan_ref import c_selective_scan_fn
    torch.manual_seed(42)
    cfg = MEDMAMBA_CONFIGS["S"]
    net = VSSM(num_classes=6, **cfg).eval()
    x = torch.randn(32, 3, 224, 224)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        got = net.to(DEV)(x.to(DEV)).cpu()
        pick = [5, 31]
        want = R.vssm_forward(sd, x[pick], cfg["depths"], c_selective_scan_fn, training=False)
    assert got.shape == (32, 6) and torch.isfinite(got).all()
    _close(got[pick], want.numpy(), 1e-3, "logits of images 5 and 31")
    # batch independence in eval mode: the same two images alone give the same logits
    with torch.no_grad():
        alone = net(x[pick].to(DEV)).cpu()
    _close(alone, got[pick].numpy(), 2e-4, "batch of 2 vs slice of the batch of 32")


def test_activation_checkpointing_with_drop_path_matches_plain_run(monkeypatch):
    """use_checkpoint=True (MedMamba.py:414-415) in training mode with DropPath: the recomputed blocks must see the mask of
    their first run and the side-stream schedule must survive the recompute inside backward -> same loss and gradients as
    the plain run with the same RNG state."""
    from medmamba_amd import modules
    monkeypatch.setattr(modules.VSSM, "_draw_drop_path", lambda self, batch, device: [])    # per-block draws in both runs
    x = torch.randn(6, 3, 64, 64, device=DEV)
    y = torch.randint(0, 3, (6,), device=DEV)
    res = {}
    for ck in (False, True):
        torch.manual_seed(5)
        net = modules.VSSM(num_classes=3, depths=[1, 2, 1, 1], dims=[16, 32, 64, 128], drop_path_rate=0.4,
                           use_checkpoint=ck).to(DEV).train()
        torch.manual_seed(77)
        torch.cuda.manual_seed(77)
        loss = torch.nn.functional.cross_entropy(net(x), y)
        loss.backward()
        res[ck] = (float(loss), {k: p.grad.clone() for k, p in net.named_parameters()})
    assert abs(res[True][0] - res[False][0]) <= 1e-6 * abs(res[False][0])
    for k, g0 in res[False][1].items():
        scale = max(1e-4, float(g0.abs().max()))
        assert float((res[True][1][k] - g0).abs().max()) <= 2e-3 * scale, k


def test_frozen_block_and_full_drop_rate(monkeypatch):
    """a block put into eval() inside a training model drops nothing (no pre-drawn factor reaches it), and
    drop_prob = 1.0 yields zeros, not NaN."""
    from medmamba_amd import modules
    torch.manual_seed(2)
    net = modules.VSSM(num_classes=3, depths=[1, 1, 2, 1], dims=[16, 32, 64, 128], drop_path_rate=0.5).to(DEV).train()
    frozen = net.layers[2].blocks[1]
    frozen.eval()
    net.layers[3].blocks[0].drop_path.drop_prob = 1.0
    seen = []
    orig = modules.shuffle_residual

    def spy(left, ssm, inp, channel_first=False, ssm_scale=None, left_relu=False, left_bias=None):
        seen.append(None if ssm_scale is None else ssm_scale.detach().clone())
        return orig(left, ssm, inp, channel_first=channel_first, ssm_scale=ssm_scale, left_relu=left_relu, left_bias=left_bias)

    monkeypatch.setattr(modules, "shuffle_residual", spy)
    out = net(torch.randn(8, 3, 32, 32, device=DEV))
    assert torch.isfinite(out).all()
    assert len(seen) == 5
    assert seen[3] is None                                   # the frozen block: no DropPath factor at all
    assert seen[4] is not None and bool((seen[4] == 0).all()) and bool(torch.isfinite(seen[4]).all())


def test_large_planes_run_in_row_strips():
    """ADVICE r1 / VERDICT weak #9: a 120 x 120 feature map (a 480^2 input) does not fit the LDS as a whole plane; the fused
    depthwise-conv kernels cut it into 32-row strips with halos, so SS2D runs at any resolution like the reference
    (MedMamba.py:288-305) — forward and backward agree with the module-by-module path (MIOpen conv + HIP scan operator)."""
    from medmamba_amd import _lib
    from medmamba_amd.modules import SS2D
    lib = _lib.lib()
    assert lib.mm_dwconv_silu_cross_strips(28, 28) == 1 and lib.mm_dwconv_silu_cross_strips(56, 56) == 2 and lib.mm_dwconv_silu_cross_strips(120, 120) == 4
    assert lib.mm_dwconv_silu_cross_supported(120, 120) == 1 and lib.mm_dwconv_silu_cross_supported(8, 100000) == 0
    torch.manual_seed(4)
    m = SS2D(d_model=2).to(DEV)
    x = torch.randn(1, 120, 120, 2, device=DEV, requires_grad=True)
    y = m(x)
    y.square().mean().backward()
    g = x.grad.clone(); x.grad = None
    gp = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    y2 = m.forward_modules(x)
    y2.square().mean().backward()
    _close(y, y2.detach().cpu().numpy(), 1e-4, "y")
    _close(g, x.grad.cpu().numpy(), 1e-3, "dx")
    for k, p in m.named_parameters():
        w = p.grad.cpu().numpy()
        assert np.abs(gp[k].cpu().numpy() - w).max() <= 2e-3 * max(1e-4, np.abs(w).max()), k


def test_wide_block_beyond_block_split_limit():
    """hidden_dim = 2048 (C/2 = 1024 > 512): the block prologue falls back to the reference op chain instead of raising."""
    from medmamba_amd.modules import SS_Conv_SSM
    torch.manual_seed(6)
    blk = SS_Conv_SSM(hidden_dim=2048, drop_path=0.0, norm_layer=torch.nn.LayerNorm).to(DEV).train()
    x = torch.randn(1, 3, 4, 2048, device=DEV, requires_grad=True)
    y = blk(x)
    y.mean().backward()
    y2 = blk.forward_modules(x.detach())
    _close(y, y2.detach().cpu().numpy(), 2e-4, "y")
    assert torch.isfinite(x.grad).all()


def test_eval_path_with_folded_batchnorm(monkeypatch):
    """eval() + no_grad folds the three BatchNorm2d of the conv branch (MedMamba.py:338-346) into the block
    prologue / the conv weights (fp64, cast once).  Same values as the reference fixture and as the unfolded path; a hooked
    block (Grad-CAM, test.py:101) and a grad-enabled eval pass keep the module-by-module / unfolded routes; changed
    statistics rebuild the fold."""
    from medmamba_amd import modules
    fx = load_golden("block_c16.npz")
    blk = modules.SS_Conv_SSM(hidden_dim=16, drop_path=0.0, norm_layer=torch.nn.LayerNorm)
    blk.load_state_dict(split_sd(fx))
    blk.to(DEV).eval()
    x = torch.from_numpy(fx["x"]).to(DEV)
    calls = []
    orig = modules.SS_Conv_SSM._forward_infer
    monkeypatch.setattr(modules.SS_Conv_SSM, "_forward_infer", lambda self, inp: (calls.append(1), orig(self, inp))[1])
    with torch.no_grad():
        y = blk(x)
    assert calls == [1]
    _close(y, fx["y_eval"], 5e-5, "folded vs reference fixture")
    y_unfolded = blk(x)                                   # grad enabled -> the unfolded autograd path
    assert calls == [1]
    _close(y, y_unfolded.detach().cpu().numpy(), 2e-5, "folded vs unfolded")
    # statistics change (a training pass) -> the cached fold is rebuilt
    blk.train()
    blk(torch.randn_like(x) * 3 + 1)
    blk.eval()
    with torch.no_grad():
        y2 = blk(x)
        monkeypatch.setattr(modules, "_FOLD_BN", False)
        y2_ref = blk(x)
    assert float((y2 - y).abs().max()) > 1e-4            # the running statistics really moved
    _close(y2, y2_ref.cpu().numpy(), 2e-5, "refolded vs unfolded")
    monkeypatch.setattr(modules, "_FOLD_BN", True)
    # hooks: module-by-module path, hook fires, same values
    seen = []
    h = blk.conv33conv33conv11[-2].register_forward_hook(lambda m, i, o: seen.append(o.shape))
    with torch.no_grad():
        y3 = blk(x)
    h.remove()
    assert len(seen) == 1 and len(calls) == 2
    _close(y3, y2_ref.cpu().numpy(), 2e-5, "hooked vs unfolded")
    # the whole tiny model in inference mode against its fixture
    fxm = load_golden("vssm_tiny.npz")
    net = modules.VSSM(num_classes=3, depths=[int(v) for v in fxm["depths"]], dims=[int(v) for v in fxm["dims"]], drop_path_rate=0.0)
    net.load_state_dict(split_sd(fxm))
    net.to(DEV).eval()
    with torch.no_grad():
        _close(net(torch.from_numpy(fxm["x"]).to(DEV)), fxm["logits_eval"], 1e-4, "tiny VSSM logits, folded")
    assert len(calls) > 2


def test_graphed_inference_matches_eager():
    """medmamba_amd.graphs.GraphedInference: the eval forward recorded into one hipGraph (library kernels through the C ABI,
    GEMMs, MIOpen convolutions, both streams) replays to the same logits as eager execution, also for new input values."""
    from medmamba_amd import modules
    from medmamba_amd.graphs import GraphedInference
    torch.manual_seed(9)
    net = modules.VSSM(num_classes=4, depths=[1, 1, 2, 1], dims=[16, 32, 64, 128]).to(DEV).eval()
    x0 = torch.randn(4, 3, 64, 64, device=DEV)
    g = GraphedInference(net, x0)
    for seed in (1, 2):
        x = torch.randn(4, 3, 64, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(seed))
        with torch.no_grad():
            want = net(x)
        got = g(x).clone()
        assert float((got - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    with pytest.raises(RuntimeError):
        g(torch.randn(2, 3, 64, 64, device=DEV))
    # weights change in place (an optimizer step / load_state_dict): the folded constants inside the graph are stale — the
    # replay notices the version counters and recaptures instead of mixing two sets of weights
    assert g.captures == 1
    sd = {k: (v * 1.5 + 0.01 if v.dtype.is_floating_point else v) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    x = torch.randn(4, 3, 64, 64, device=DEV)
    got = g(x).clone()
    assert g.captures == 2
    with torch.no_grad():
        want = net(x)
    assert float((got - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    g(x)
    assert g.captures == 2


def test_bn_recalibration_invalidates_the_fold():
    """only the BatchNorm modules are switched to train() and back (the block's own train() is never called); our BN
    kernels update the running statistics through raw pointers and must still invalidate the block's folded constants."""
    from medmamba_amd import modules
    torch.manual_seed(4)
    blk = modules.SS_Conv_SSM(hidden_dim=32, drop_path=0.0, norm_layer=torch.nn.LayerNorm).to(DEV).eval()
    x = torch.randn(2, 12, 12, 32, device=DEV)
    with torch.no_grad():
        y0 = blk(x)                                               # builds the fold
        bns = [m for m in blk.conv33conv33conv11 if isinstance(m, torch.nn.BatchNorm2d)]
        for bn in bns:
            bn.train()
        for _ in range(3):
            blk(torch.randn_like(x) * 2 + 1)                      # recalibration passes: statistics move
        for bn in bns:
            bn.eval()
        y1 = blk(x)
    prev = modules._FOLD_BN
    modules._FOLD_BN = False
    try:
        with torch.no_grad():
            y1_ref = blk(x)
    finally:
        modules._FOLD_BN = prev
    assert float((y1 - y0).abs().max()) > 1e-4
    assert float((y1 - y1_ref).abs().max()) <= 2e-5 * max(1.0, float(y1_ref.abs().max()))


def test_graphed_blocks_match_eager(monkeypatch):
    """MM_GRAPH_BLOCK_MAX_L (experimental): whole blocks replayed from hipGraphs in training — same loss, gradients and
    BatchNorm statistics as eager execution, over several steps (static input / gradient buffers are reused)."""
    from medmamba_amd import modules
    x = torch.randn(4, 3, 64, 64, device=DEV)
    y = torch.randint(0, 3, (4,), device=DEV)
    res = {}
    for graphed in (False, True):
        monkeypatch.setattr(modules, "_GRAPH_BLOCK_MAX_L", (1 << 20) if graphed else 0)
        torch.manual_seed(5)
        net = modules.VSSM(num_classes=3, depths=[1, 2, 1, 1], dims=[16, 32, 64, 128], drop_path_rate=0.0).to(DEV).train()
        opt = torch.optim.SGD(net.parameters(), lr=1e-2)
        losses = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            loss = torch.nn.functional.cross_entropy(net(x), y)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        res[graphed] = (losses, {k: v.clone() for k, v in net.state_dict().items()})
    for a, b in zip(res[True][0], res[False][0]):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(b)), (res[True][0], res[False][0])
    for k, v in res[False][1].items():
        w = res[True][1][k]
        if v.dtype.is_floating_point:
            assert float((v - w).abs().max()) <= 1e-4 * max(1e-2, float(v.abs().max())), k
        else:
            assert torch.equal(v, w), k


@pytest.mark.parametrize("kw", [dict(d_conv=5), dict(conv_bias=False), dict(bias=True), dict(expand=1), dict(dt_rank=5),
                                dict(expand=3, dt_rank=2, bias=True, conv_bias=False), dict(dropout=0.0, dt_init="constant")],
                         ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()))
def test_ss2d_constructor_variants_fused_vs_module_path(kw):
    """SS2D's constructor surface (MedMamba.py:124-142) beyond the defaults MedMamba uses: other depthwise kernel sizes (MIOpen
    conv + fused core), no conv bias, projection biases, other expansion factors and dt ranks.  The fused channel-first path
    must agree with the module-by-module path (the reference's own op chain around the HIP scan operator), forward and backward."""
    from medmamba_amd.modules import SS2D
    torch.manual_seed(11)
    m = SS2D(d_model=12, **kw).to(DEV)
    x = torch.randn(2, 6, 10, 12, device=DEV, requires_grad=True)
    dy = torch.randn(2, 6, 10, 12, device=DEV)
    y = m(x)
    y.backward(dy)
    g = x.grad.clone(); x.grad = None
    gp = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    y2 = m.forward_modules(x)
    y2.backward(dy)
    _close(y, y2.detach().cpu().numpy(), 5e-5, "y")
    _close(g, x.grad.cpu().numpy(), 2e-4, "dx")
    assert set(gp) == {k for k, p in m.named_parameters() if p.grad is not None}
    for k, p in m.named_parameters():
        w = p.grad.cpu().numpy()
        assert np.abs(gp[k].cpu().numpy() - w).max() <= 5e-4 * max(1e-3, np.abs(w).max()), (k, kw)


@pytest.mark.parametrize("kw", [dict(patch_norm=False), dict(attn_drop_rate=0.3), dict(dims=24, depths=[1, 1, 1]),
                                dict(patch_size=2, in_chans=1), dict(norm_layer=torch.nn.Identity)],
                         ids=lambda k: ",".join(f"{a}" for a in k))
def test_vssm_constructor_variants_fused_vs_hooked_path(kw):
    """VSSM's constructor surface (MedMamba.py:423-470) beyond the defaults: the fused paths against the module-by-module path
    that a forward hook on any sub-module selects (eval mode: Dropout / DropPath inactive), logits and input gradient."""
    from medmamba_amd.modules import VSSM
    torch.manual_seed(5)
    args = dict(num_classes=4, depths=[1, 1], dims=[16, 32], drop_path_rate=0.0)
    args.update(kw)
    if kw.get("norm_layer") is torch.nn.Identity:
        args["norm_layer"] = lambda d: torch.nn.Identity()
    net = VSSM(**args).to(DEV).eval()
    cin, ps = args.get("in_chans", 3), args.get("patch_size", 4)
    x = torch.randn(2, cin, 8 * ps, 8 * ps, device=DEV, requires_grad=True)
    y = net(x)
    y.square().sum().backward()
    g = x.grad.clone(); x.grad = None
    handles = [m.register_forward_hook(lambda mod, inp, out: None) for m in net.modules() if not list(m.children())]
    y2 = net(x)
    y2.square().sum().backward()
    for h in handles:
        h.remove()
    assert y.shape == (2, 4)
    _close(y, y2.detach().cpu().numpy(), 1e-4, "logits")
    _close(g, x.grad.cpu().numpy(), 1e-3, "dx")


@pytest.mark.parametrize("layout", ["bm", "cm"])
def test_cpp_sequenced_ss2d_branch_is_the_python_route_bit_for_bit(layout, monkeypatch):
    """csrc_host/ss2d_host.cpp issues the launches of InProjFn + SS2DCoreFn + OutProjFn from C++: same kernels, same GEMM shapes,
    same allocations — output, input gradient and all eleven parameter gradients are identical bits in both storage layouts."""
    from medmamba_amd import _host, modules, ops
    from medmamba_amd.selective_scan_interface import KERNEL_TIMER
    assert _host.module() is not None, "lib/_mm_host.so is missing: python -m medmamba_amd.build"
    torch.manual_seed(21)
    ss = modules.SS2D(d_model=24, d_state=16, expand=2).to(DEV).train()
    monkeypatch.setattr(ops, "_LAYOUT", layout)
    x = torch.randn(3, 12, 8, 24, device=DEV, requires_grad=True)
    g = torch.randn(3, 24, 96, device=DEV)
    calls = []
    real = ops.SS2DBranchFn.apply
    monkeypatch.setattr(ops.SS2DBranchFn, "apply", lambda *a: (calls.append(1), real(*a))[1])

    def run(native):
        if not native:
            monkeypatch.setattr(ops, "ss2d_branch_native_ok", lambda *a: False)
        ss.zero_grad(set_to_none=True)
        x.grad = None
        out = ss.forward_cf(x)
        out.backward(g)
        return out.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in ss.named_parameters()}

    KERNEL_TIMER.records.clear()
    KERNEL_TIMER.enabled = True
    try:
        o1, dx1, g1 = run(True)
        torch.cuda.synchronize()
        tags = [r[0] for r in KERNEL_TIMER.records]
        assert tags == ["scan_fwd", "scan_bwd"] and all(s.elapsed_time(e) > 0 for _, s, e, *_ in KERNEL_TIMER.records)
    finally:
        KERNEL_TIMER.enabled = False
        KERNEL_TIMER.records.clear()
    assert calls == [1]
    o0, dx0, g0 = run(False)
    assert calls == [1]
    assert torch.equal(o1, o0) and torch.equal(dx1, dx0)
    assert set(g1) == set(g0) and len(g0) == 11
    for k in g0:
        assert torch.equal(g1[k], g0[k]), k


@pytest.mark.parametrize("hw", [(8, 8), (24, 28)])
def test_cpp_sequenced_conv_branch_is_the_python_route_bit_for_bit(hw, monkeypatch):
    """csrc_host conv_branch_fwd / _bwd against modules._conv_branch (BNReluFn, ConvBiasFn, PointwiseConvFn node by node): block
    output, input gradient, every parameter gradient, the BatchNorm running statistics and step counters are identical bits
    (planes of 64 positions take the one-kernel BatchNorm with the folded bias gradient, 672 positions the two-kernel form)."""
    from medmamba_amd import _host, modules, ops
    assert _host.module() is not None
    torch.manual_seed(8)
    blk = modules.SS_Conv_SSM(hidden_dim=128, drop_path=0.0, norm_layer=torch.nn.LayerNorm).to(DEV).train()
    state0 = {k: v.clone() for k, v in blk.state_dict().items()}
    x = torch.randn(4, hw[0], hw[1], 128, device=DEV, requires_grad=True)
    g = torch.randn(4, hw[0], hw[1], 128, device=DEV)
    calls = []
    real = ops.ConvBranchFn.apply
    monkeypatch.setattr(ops.ConvBranchFn, "apply", lambda *a: (calls.append(1), real(*a))[1])

    def run(native):
        blk.load_state_dict(state0)
        if not native:
            monkeypatch.setattr(ops, "conv_branch_native", lambda *a: None)
        outs = []
        for _ in range(2):                 # two passes: running statistics accumulate, the second pass overlaps the two streams
            blk.zero_grad(set_to_none=True)
            x.grad = None
            out = blk(x)
            out.backward(g)
            outs.append((out.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()}))
        torch.cuda.synchronize()
        return outs, {k: v.clone() for k, v in blk.state_dict().items() if "running" in k or "num_batches" in k}

    r1, b1 = run(True)
    assert calls == [1, 1]
    r0, b0 = run(False)
    assert calls == [1, 1]
    for (o1, dx1, g1), (o0, dx0, g0) in zip(r1, r0):
        assert torch.equal(o1, o0) and torch.equal(dx1, dx0)
        for k in g0:
            if k in ("conv33conv33conv11.1.weight", "conv33conv33conv11.4.weight"):
                # MIOpen's weight-gradient kernels accumulate with atomics at some shapes: run-to-run noise in either route
                assert torch.allclose(g1[k], g0[k], rtol=1e-4, atol=1e-4 * float(g0[k].abs().max())), k
            else:
                assert torch.equal(g1[k], g0[k]), k
    assert len(b0) == 9
    for k in b0:
        assert torch.equal(b1[k], b0[k]), k
    assert int(b1["conv33conv33conv11.0.num_batches_tracked"]) == int(state0["conv33conv33conv11.0.num_batches_tracked"]) + 2


def test_eval_mode_with_gradients_takes_the_same_values_on_both_routes(monkeypatch):
    """Saliency-style use (test.py:101-108 without hooks): eval() but gradients w.r.t. the input wanted.  The SS2D branch then runs
    the C++-sequenced training form (checkpoints, no fused dt projection), the conv branch its module chain with running
    statistics; logits and input gradient agree with the all-Python route."""
    from medmamba_amd import modules, ops
    torch.manual_seed(12)
    net = modules.VSSM(num_classes=4, depths=[1, 1, 1, 1], dims=[16, 32, 64, 128], drop_path_rate=0.0).to(DEV).eval()
    x = torch.randn(2, 3, 64, 64, device=DEV, requires_grad=True)

    def run():
        x.grad = None
        logits = net(x)
        logits[:, 1].sum().backward()
        return logits.detach().clone(), x.grad.clone()

    calls = []
    real = ops.SS2DBranchFn.apply
    monkeypatch.setattr(ops.SS2DBranchFn, "apply", lambda *a: (calls.append(1), real(*a))[1])
    l1, g1 = run()
    assert len(calls) == 4
    monkeypatch.setattr(ops, "ss2d_branch_native_ok", lambda *a: False)
    l0, g0 = run()
    assert len(calls) == 4
    assert torch.allclose(l1, l0, rtol=1e-5, atol=1e-6)
    assert float((g1 - g0).abs().max()) <= 1e-4 * max(1e-6, float(g0.abs().max()))
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


def test_deterministic_mode_weight_gradient_of_the_dense_convs(monkeypatch):
    """torch.backends.cudnn.deterministic (the reference's set_seed, train.py:28-29): the 3x3 convs' weight gradient is one im2col +
    one batched GEMM + an ordered sum over the batch instead of MIOpen's per-image solver — the same values as MIOpen's default
    weight gradient, identical bits run after run, on the C++-sequenced route and on the Python route."""
    from medmamba_amd import modules, ops
    torch.manual_seed(9)
    blk = modules.SS_Conv_SSM(hidden_dim=96, drop_path=0.0, norm_layer=torch.nn.LayerNorm).to(DEV).train()
    state0 = {k: v.clone() for k, v in blk.state_dict().items()}
    x = torch.randn(8, 28, 28, 96, device=DEV, requires_grad=True)
    g = torch.randn(8, 28, 28, 96, device=DEV)

    def run():
        blk.load_state_dict(state0)
        blk.zero_grad(set_to_none=True)
        x.grad = None
        blk(x).backward(g)
        return x.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()}

    run()                                             # first pass over the shapes (solver search)
    dx0, g0 = run()                                   # MIOpen's default weight gradient
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)
    dx1, g1 = run()
    dx2, g2 = run()
    monkeypatch.setattr(ops, "conv_branch_native", lambda *a: None)
    dx3, g3 = run()
    assert torch.equal(dx1, dx2) and torch.equal(dx1, dx3)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k           # run to run
        assert torch.equal(g1[k], g3[k]), k           # C++ route vs Python route
        scale = max(1e-4, float(g0[k].abs().max()))
        assert float((g1[k] - g0[k]).abs().max()) <= 2e-4 * scale, k


@pytest.mark.parametrize("native", [True, False])
def test_no_grad_forward_with_trainable_parameters_takes_the_inference_form(native, monkeypatch):
    """ctx.needs_input_grad ignores torch.no_grad(), so eval() + no_grad() with requires_grad parameters (bench config 2,
    trainer.evaluate, GraphedInference) used to write state checkpoints and run the dt GEMM.  The wrappers now pass grad mode."""
    from medmamba_amd import modules, ops, selective_scan_interface as ssi
    torch.manual_seed(4)
    ss = modules.SS2D(d_model=24, d_state=16, expand=2).to(DEV).eval()
    assert all(p.requires_grad for p in ss.parameters())
    x = torch.randn(2, 8, 8, 24, device=DEV)
    seen = []
    real_apply = ops.SS2DBranchFn.apply
    monkeypatch.setattr(ops.SS2DBranchFn, "apply", lambda *a: (seen.append(("branch", a[-1])), real_apply(*a))[1])
    real_launch = ssi._launch_fwd
    monkeypatch.setattr(ssi, "_launch_fwd", lambda *a, **k: (seen.append(("launch", a[8], k.get("dt") is not None)), real_launch(*a, **k))[1])
    if not native:
        monkeypatch.setattr(ops, "ss2d_branch_native_ok", lambda *a: False)
    with torch.no_grad():
        y0 = ss.forward_cf(x)
    if native:
        assert seen == [("branch", False)]
    else:
        assert seen == [("launch", False, True)], seen      # no checkpoints, dt projection fused (rank 2 <= mm_scan_dt_max)
    seen.clear()
    y1 = ss.forward_cf(x)                                   # grad mode on: the training form
    assert seen == ([("branch", True)] if native else [("launch", True, False)]), seen
    assert y1.requires_grad and torch.allclose(y0, y1, rtol=1e-5, atol=1e-6)
