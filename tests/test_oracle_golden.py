"""CPU: the oracle (oracle/*.py, oracle/selective_scan_ref.c) against the committed golden vectors.

The vectors were produced by executing the real reference (tools/gen_golden.py); the scan-level
ones come from the restated selective_scan_ref loop (temp.py:57-139) and torch autograd of it.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, split_sd
from oracle import model_ref as R
from oracle.scan_ref import c_scan_bwd, c_scan_fwd, c_selective_scan_fn, selective_scan_ref

SCAN_CASES = ["scan_small.npz", "scan_mid.npz", "scan_long.npz"]


def _scan_inputs(fx):
    Rk, N = int(fx["R"]), fx["A"].shape[1]
    xd = torch.from_numpy(fx["x_dbl"])
    t = lambda k: torch.from_numpy(fx[k])
    return t("u"), t("delta"), t("A"), xd[:, :, Rk:Rk + N], xd[:, :, Rk + N:], t("D"), t("delta_bias")


@pytest.mark.parametrize("name", SCAN_CASES)
def test_c_oracle_forward_matches_golden(name):
    fx = load_golden(name)
    u, delta, A, B, C, D, bias = _scan_inputs(fx)
    assert not B.is_contiguous()            # the reference passes views of x_dbl
    o32 = c_scan_fwd(u, delta, A, B, C, D, bias, True)
    o64 = c_scan_fwd(u, delta, A, B, C, D, bias, True, f64=True)
    scale = np.abs(fx["out"]).max()
    # fp32 C vs fp32 torch loop: same algorithm, different rounding of expf / einsum order
    assert np.abs(o32 - fx["out"]).max() <= 2e-5 * scale
    assert np.abs(o64 - fx["out"]).max() <= 2e-5 * scale


@pytest.mark.parametrize("name", SCAN_CASES)
def test_c_oracle_backward_matches_autograd_golden(name):
    fx = load_golden(name)
    u, delta, A, B, C, D, bias = _scan_inputs(fx)
    r = c_scan_bwd(u, delta, A, B, C, D, bias, fx["dout"], True)
    Rk, N = int(fx["R"]), A.shape[1]
    dx = fx["dx_dbl"]
    want = dict(du=fx["du"], ddelta=fx["ddelta"], dA=fx["dA"], dB=dx[:, :, Rk:Rk + N], dC=dx[:, :, Rk + N:],
                dD=fx["dD"], ddelta_bias=fx["ddelta_bias"])
    for k, w in want.items():
        err = np.abs(r[k] - w).max()
        assert err <= 1e-4 * max(1.0, np.abs(w).max()), (k, err)   # golden grads are fp32 autograd
    assert np.abs(fx["dx_dbl"][:, :, :Rk]).max() == 0.0            # dt rows of x_dbl get no grad from the scan


def test_python_loop_matches_golden_bitwise_small():
    fx = load_golden("scan_small.npz")
    u, delta, A, B, C, D, bias = _scan_inputs(fx)
    out = selective_scan_ref(u, delta, A, B, C, D, z=None, delta_bias=bias, delta_softplus=True)
    assert np.array_equal(out.numpy(), fx["out"])


def test_chunk_states_consistent():
    fx = load_golden("scan_small.npz")
    u, delta, A, B, C, D, bias = _scan_inputs(fx)
    o, xs = c_scan_fwd(u, delta, A, B, C, D, bias, True, f64=True, chunk_states=16)
    assert xs.shape == (2, 32, 3, 16)
    o2, xs2 = c_scan_fwd(u[:, :, :32], delta[:, :, :32], A, B[..., :32], C[..., :32], D, bias, True, f64=True,
                         chunk_states=32)
    assert np.allclose(xs[:, :, 1], xs2[:, :, 0])


@pytest.mark.parametrize("name", ["ss2d_d8.npz", "ss2d_d48.npz"])
@pytest.mark.parametrize("scan", ["loop", "c"])
def test_ss2d_restatement_matches_reference(name, scan):
    fx = load_golden(name)
    p = split_sd(fx)
    x = torch.from_numpy(fx["x"])
    fn = selective_scan_ref if scan == "loop" else c_selective_scan_fn
    y = R.ss2d_forward(p, "", x, fn)
    assert np.abs(y.numpy() - fx["y"]).max() <= 2e-5 * max(1.0, np.abs(fx["y"]).max())
    xz = torch.nn.functional.linear(x, p["in_proj.weight"])
    xc = torch.nn.functional.silu(torch.nn.functional.conv2d(
        xz.chunk(2, -1)[0].permute(0, 3, 1, 2).contiguous(), p["conv2d.weight"], p["conv2d.bias"], padding=1,
        groups=p["conv2d.weight"].shape[0]))
    assert np.allclose(xc.numpy(), fx["conv_out"], atol=1e-6)
    core = R.ss2d_core(p, "", xc, fn)
    assert np.abs(core.numpy() - fx["core_out"]).max() <= 2e-5 * max(1.0, np.abs(fx["core_out"]).max())


def test_ss2d_restatement_grads_match_reference():
    fx = load_golden("ss2d_d8.npz")
    p = {k: v.requires_grad_() for k, v in split_sd(fx).items()}
    x = torch.from_numpy(fx["x"]).requires_grad_()
    R.ss2d_forward(p, "", x, c_selective_scan_fn).backward(torch.from_numpy(fx["dy"]))
    assert np.abs(x.grad.numpy() - fx["dx"]).max() <= 1e-4 * max(1.0, np.abs(fx["dx"]).max())
    for k, v in p.items():
        w = fx["grad/" + k]
        assert np.abs(v.grad.numpy() - w).max() <= 2e-4 * max(1.0, np.abs(w).max()), k


def test_block_restatement_matches_reference():
    fx = load_golden("block_c16.npz")
    p = split_sd(fx)
    x = torch.from_numpy(fx["x"])
    y = R.block_forward(p, "", x, c_selective_scan_fn, training=False)
    assert np.abs(y.numpy() - fx["y_eval"]).max() <= 2e-5
    upd = {}
    y = R.block_forward(p, "", x, c_selective_scan_fn, training=True, bn_updates=upd)
    assert np.abs(y.numpy() - fx["y_train"]).max() <= 5e-5
    for k, v in upd.items():
        assert np.allclose(v.numpy(), fx["sd_after/" + k], atol=1e-6), k


def test_channel_shuffle_is_interleave():
    x = torch.arange(2 * 3 * 4 * 8, dtype=torch.float32).view(2, 3, 4, 8)
    y = R.channel_shuffle(x, 2)
    # out channel 2i+j <- in channel j*(C/2)+i  
    for i in range(4):
        for j in range(2):
            assert torch.equal(y[..., 2 * i + j], x[..., j * 4 + i])


def test_patch_merging_restatement_even_and_odd():
    fx = load_golden("patchmerge_c8.npz")
    p = split_sd(fx)
    for tag in ("even", "odd"):
        y = R.patch_merging(p, "", torch.from_numpy(fx["x_" + tag]))
        assert y.shape == fx["y_" + tag].shape
        assert np.abs(y.numpy() - fx["y_" + tag]).max() <= 1e-5


def test_vssm_tiny_restatement_matches_reference():
    fx = load_golden("vssm_tiny.npz")
    p = split_sd(fx)
    x = torch.from_numpy(fx["x"])
    depths = [int(v) for v in fx["depths"]]
    le = R.vssm_forward(p, x, depths, c_selective_scan_fn, training=False)
    assert np.abs(le.numpy() - fx["logits_eval"]).max() <= 2e-5
    pg = {k: (v.requires_grad_() if v.dtype.is_floating_point and "running" not in k else v) for k, v in p.items()}
    lt = R.vssm_forward(pg, x, depths, c_selective_scan_fn, training=True)
    assert np.abs(lt.detach().numpy() - fx["logits_train"]).max() <= 5e-5
    loss = torch.nn.functional.cross_entropy(lt, torch.from_numpy(fx["labels"]))
    assert abs(float(loss) - float(fx["loss"])) <= 1e-5
    loss.backward()
    worst = 0.0
    for k, v in fx.items():
        if k.startswith("grad/"):
            g = pg[k[5:]].grad.numpy()
            worst = max(worst, np.abs(g - v).max() / max(1e-3, np.abs(v).max()))
    assert worst <= 2e-3, worst
