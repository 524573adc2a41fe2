"""ORACLE — test infrastructure only.  Never imported by the product (`medmamba_amd/`).

Functional (state-dict driven) CPU restatement of the reference's in-tree hot path, plain PyTorch
ops only, each function citing the reference lines it follows.  Unlike oracle/scan_ref.py this part
IS pinned by the reference itself: tools/gen_golden.py executes the real /root/reference/MedMamba.py
(with `timm.layers` / `mamba_ssm` stubbed, the scan being oracle.scan_ref.selective_scan_ref) and
tests/test_oracle_golden.py checks every function here against those fixtures.

`scan` is any callable with selective_scan_fn's signature (MedMamba.py:273-279).
"""
import math

import torch
import torch.nn.functional as F

from .scan_ref import selective_scan_ref

LN_EPS = 1e-5   # VSSM passes nn.LayerNorm (eps 1e-5) down to every block: MedMamba.py:426,461,391-393
BN_EPS = 1e-5


def cross_scan(x):
    """MedMamba.py:256-257. x (B,D,H,W) -> xs (B,4,D,L): row-major, col-major, and their flips."""
    B, D, H, W = x.shape
    L = H * W
    x_hwwh = torch.stack([x.reshape(B, -1, L), x.transpose(2, 3).contiguous().view(B, -1, L)], dim=1)
    return torch.cat([x_hwwh, torch.flip(x_hwwh, dims=[-1])], dim=1)


def cross_merge(out_y, H, W):
    """MedMamba.py:282-286 + the 4-way sum of :298. out_y (B,4,D,L) -> y (B,D,L)."""
    B, K, D, L = out_y.shape
    inv_y = torch.flip(out_y[:, 2:4], dims=[-1]).view(B, 2, -1, L)
    wh_y = out_y[:, 1].view(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
    invwh_y = inv_y[:, 1].view(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
    return out_y[:, 0] + inv_y[:, 0] + wh_y + invwh_y


def ss2d_core(p, pre, x, scan=selective_scan_ref):
    """SS2D.forward_corev0, MedMamba.py:249-286. x (B,D,H,W) -> y (B,D,L) (already 4-way summed)."""
    B, D, H, W = x.shape
    L, K = H * W, 4
    xw, dtw = p[pre + "x_proj_weight"], p[pre + "dt_projs_weight"]
    R, N = dtw.shape[2], p[pre + "A_logs"].shape[1]
    xs = cross_scan(x)
    x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, xw)                            # :259
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)                        # :261
    dts = torch.einsum("bkrl,kdr->bkdl", dts, dtw)                            # :262
    out_y = scan(xs.float().reshape(B, -1, L), dts.contiguous().float().view(B, -1, L),
                 -torch.exp(p[pre + "A_logs"].float()).view(-1, N),           # :270
                 Bs.float(), Cs.float(), p[pre + "Ds"].float().view(-1), z=None,
                 delta_bias=p[pre + "dt_projs_bias"].float().view(-1),
                 delta_softplus=True, return_last_state=False).view(B, K, -1, L)
    return cross_merge(out_y, H, W)


def ss2d_forward(p, pre, x, scan=selective_scan_ref):
    """SS2D.forward, MedMamba.py:288-305. x (B,H,W,d_model) -> (B,H,W,d_model)."""
    B, H, W, _ = x.shape
    xz = F.linear(x, p[pre + "in_proj.weight"], p.get(pre + "in_proj.bias"))  # :291
    x, z = xz.chunk(2, dim=-1)                                                # :292
    x = x.permute(0, 3, 1, 2).contiguous()                                    # :294
    D = x.shape[1]
    x = F.silu(F.conv2d(x, p[pre + "conv2d.weight"], p.get(pre + "conv2d.bias"), padding=1, groups=D))  # :295
    y = ss2d_core(p, pre, x, scan)                                            # :296-298
    y = y.transpose(1, 2).contiguous().view(B, H, W, -1)                      # :299
    y = F.layer_norm(y, (D,), p[pre + "out_norm.weight"], p[pre + "out_norm.bias"], LN_EPS)  # :300
    y = y * F.silu(z)                                                         # :301
    return F.linear(y, p[pre + "out_proj.weight"], p.get(pre + "out_proj.bias"))  # :302


def channel_shuffle(x, groups):
    """MedMamba.py:308-320."""
    b, h, w, c = x.shape
    return x.view(b, h, w, groups, c // groups).transpose(3, 4).contiguous().view(b, h, w, -1)


def _bn(p, pre, x, training, bn_updates):
    """nn.BatchNorm2d forward. training=True uses batch statistics (and records the running-stat update)."""
    w, b = p[pre + "weight"], p[pre + "bias"]
    if not training:
        return F.batch_norm(x, p[pre + "running_mean"], p[pre + "running_var"], w, b, False, 0.0, BN_EPS)
    rm, rv = p[pre + "running_mean"].clone(), p[pre + "running_var"].clone()
    y = F.batch_norm(x, rm, rv, w, b, True, 0.1, BN_EPS)
    if bn_updates is not None:
        bn_updates[pre + "running_mean"], bn_updates[pre + "running_var"] = rm, rv
    return y


def conv_branch(p, pre, x, training=False, bn_updates=None):
    """SS_Conv_SSM.conv33conv33conv11, MedMamba.py:337-347. x NCHW (B,C/2,H,W)."""
    q = pre + "conv33conv33conv11."
    x = _bn(p, q + "0.", x, training, bn_updates)
    x = F.conv2d(x, p[q + "1.weight"], p[q + "1.bias"], padding=1)
    x = F.relu(_bn(p, q + "2.", x, training, bn_updates))
    x = F.conv2d(x, p[q + "4.weight"], p[q + "4.bias"], padding=1)
    x = F.relu(_bn(p, q + "5.", x, training, bn_updates))
    return F.relu(F.conv2d(x, p[q + "7.weight"], p[q + "7.bias"]))


def block_forward(p, pre, inp, scan=selective_scan_ref, training=False, bn_updates=None):
    """SS_Conv_SSM.forward, MedMamba.py:349-357 (DropPath = identity: eval or drop_path 0)."""
    left, right = inp.chunk(2, dim=-1)                                                       # :350
    C2 = right.shape[-1]
    r = F.layer_norm(right, (C2,), p[pre + "ln_1.weight"], p[pre + "ln_1.bias"], LN_EPS)
    x = ss2d_forward(p, pre + "self_attention.", r, scan)                                    # :351
    left = conv_branch(p, pre, left.permute(0, 3, 1, 2).contiguous(), training, bn_updates)  # :352-353
    left = left.permute(0, 2, 3, 1).contiguous()                                             # :354
    return channel_shuffle(torch.cat((left, x), dim=-1), 2) + inp                            # :355-357


def patch_merging(p, pre, x):
    """PatchMerging2D.forward, MedMamba.py:93-119 (odd H/W are cropped, :97-111)."""
    B, H, W, C = x.shape
    h2, w2 = H // 2, W // 2
    parts = [x[:, 0::2, 0::2, :], x[:, 1::2, 0::2, :], x[:, 0::2, 1::2, :], x[:, 1::2, 1::2, :]]
    parts = [t[:, :h2, :w2, :] for t in parts]
    x = torch.cat(parts, -1).reshape(B, h2, w2, 4 * C)
    x = F.layer_norm(x, (4 * C,), p[pre + "norm.weight"], p[pre + "norm.bias"], LN_EPS)
    return F.linear(x, p[pre + "reduction.weight"])


def patch_embed(p, x):
    """PatchEmbed2D.forward, MedMamba.py:72-76 (patch 4, stride 4, LayerNorm)."""
    w = p["patch_embed.proj.weight"]
    x = F.conv2d(x, w, p["patch_embed.proj.bias"], stride=w.shape[-1]).permute(0, 2, 3, 1)
    return F.layer_norm(x, (w.shape[0],), p["patch_embed.norm.weight"], p["patch_embed.norm.bias"], LN_EPS)


def vssm_forward(p, x, depths, scan=selective_scan_ref, training=False, bn_updates=None):
    """VSSM.forward, MedMamba.py:499-515. p: state dict of the reference model; x (B,3,H,W)."""
    x = patch_embed(p, x)
    for i, depth in enumerate(depths):
        for j in range(depth):
            x = block_forward(p, f"layers.{i}.blocks.{j}.", x, scan, training, bn_updates)
        if i < len(depths) - 1:
            x = patch_merging(p, f"layers.{i}.downsample.", x)
    x = x.permute(0, 3, 1, 2).mean(dim=(2, 3))                       # :511-513 AdaptiveAvgPool2d(1)+flatten
    return F.linear(x, p["head.weight"], p["head.bias"])


CONFIGS = {  # train.py:179-182
    "T": dict(depths=[2, 2, 4, 2], dims=[96, 192, 384, 768]),
    "S": dict(depths=[2, 2, 8, 2], dims=[96, 192, 384, 768]),
    "B": dict(depths=[2, 2, 12, 2], dims=[128, 256, 512, 1024]),
    "Te": dict(depths=[2, 3, 3, 2], dims=[96, 192, 384, 768]),
}


def dt_rank(dim):
    """SS2D gets d_model = dim//2 (MedMamba.py:334) and dt_rank = ceil(d_model/16) (:150)."""
    return math.ceil((dim // 2) / 16)


def shuffle_residual_ref(left_nchw, ssm, inp_nhwc, channel_first=False, ssm_scale=None, left_relu=False, left_bias=None):
    """Test double for medmamba_amd.ops.shuffle_residual: the reference's own op chain, MedMamba.py:354-357 (with the
    trailing ReLU of the conv branch, :347, and the DropPath factor of :353 when they are handed over).
    channel_first: ssm is (B, C/2, H*W) instead of (B, H, W, C/2)."""
    if left_bias is not None:                     # the closing 1x1 conv's bias (MedMamba.py:345), handed over by the caller
        left_nchw = left_nchw + left_bias.reshape(1, -1, 1, 1)
    if left_relu:
        left_nchw = F.relu(left_nchw)
    left = left_nchw.permute(0, 2, 3, 1).contiguous()
    if channel_first:
        ssm = ssm.transpose(1, 2).reshape(left.shape)
    if ssm_scale is not None:
        ssm = ssm * ssm_scale.reshape(-1, 1, 1, 1)
    return channel_shuffle(torch.cat((left, ssm), dim=-1), 2) + inp_nhwc


def in_proj_cf_ref(x_rows, weight, bias):
    """Test double for medmamba_amd.ops.in_proj_cf: in_proj + chunk (MedMamba.py:291-292), returned channel-first."""
    xz = F.linear(x_rows, weight, bias).transpose(1, 2)
    D = weight.shape[0] // 2
    return xz[:, :D], xz[:, D:]


def dwconv_silu_cross_ref(x_cf, weight, bias, H, W):
    """Test double for medmamba_amd.ops.dwconv_silu_cross: MedMamba.py:295 on (B,D,L) planes + the two orders of :256."""
    B, D, L = x_cf.shape
    xc = F.silu(F.conv2d(x_cf.reshape(B, D, H, W), weight, bias, padding=1, groups=D))
    return torch.stack([xc.reshape(B, D, L), xc.transpose(2, 3).reshape(B, D, L)], 1).reshape(B, 2 * D, L)


def ss2d_core_ref(u2, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, z_cf, ln_w, ln_b, H, W, eps=1e-5,
                  prescan_event=None):
    """Test double for medmamba_amd.ops.ss2d_core: the x / dt projections (MedMamba.py:259-262) as einsums, A = -exp(A_logs)
    (:271), the oracle scan on explicitly flipped tensors, the reference's merge (:282-286, 298), out_norm (:300) and gate
    (:301), channel-first.  Parameters in the module's (reference) direction order k; the scan double wants kernel order g."""
    from .scan_ref import c_cross_scan_fn
    B, D2, L = u2.shape
    D, R, N = D2 // 2, dt_projs_weight.shape[2], A_logs.shape[1]
    pk = lambda t: torch.stack([t[k] for k in (0, 2, 1, 3)], 0)      # kernel direction g -> reference direction k
    Wx, Wdt = pk(x_proj_weight), pk(dt_projs_weight)
    A = -torch.exp(pk(A_logs.float().view(4, D, N))).reshape(4 * D, N)
    Dp, dbias = pk(Ds.float().view(4, D)).reshape(-1), pk(dt_projs_bias.float()).reshape(-1)
    u4 = u2.view(B, 2, 1, D, L).expand(B, 2, 2, D, L).reshape(B, 4, D, L)           # direction g reads block g // 2
    x_dbl = torch.einsum("bgdl,gcd->bgcl", u4, Wx)
    delta = torch.einsum("bgrl,gdr->bgdl", x_dbl[:, :, :R], Wdt).reshape(B, 4 * D, L)
    y2 = c_cross_scan_fn(u2, delta, A, x_dbl[:, :, R:R + N], x_dbl[:, :, R + N:], Dp, dbias).view(B, 2, D, L)
    m = y2[:, 0] + y2[:, 1].reshape(B, D, W, H).transpose(2, 3).reshape(B, D, L)
    n = F.layer_norm(m.transpose(1, 2), (D,), ln_w, ln_b, eps).transpose(1, 2)
    return n * F.silu(z_cf)


def ss2d_conv_core_ref(x_cf, conv_weight, conv_bias, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, z_cf, ln_w,
                       ln_b, H, W, eps=1e-5, prescan_event=None):
    """Test double for medmamba_amd.ops.ss2d_conv_core: dwconv_silu_cross_ref followed by ss2d_core_ref."""
    u2 = dwconv_silu_cross_ref(x_cf, conv_weight, conv_bias, H, W)
    return ss2d_core_ref(u2, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, z_cf, ln_w, ln_b, H, W, eps)


def block_split_ref(inp, gamma, beta, eps):
    """Test double for medmamba_amd.ops.block_split: chunk + permute + ln_1 exactly as MedMamba.py:350-352."""
    left, right = inp.chunk(2, dim=-1)
    return left.permute(0, 3, 1, 2).contiguous(), F.layer_norm(right, (right.shape[-1],), gamma, beta, eps), inp
