"""nn.Module surface of the MedMamba hot path, backed by the gfx950 HIP kernels.

Same constructor arguments, attribute names, forward signatures and state-dict keys/shapes as the
reference model file, so reference checkpoints load and reference-style callers
(`net.layers[-1].blocks[-1].conv33conv33conv11[-2]`, test.py:101) keep working:

    PatchEmbed2D      MedMamba.py:54-76        SS2D          MedMamba.py:123-305
    PatchMerging2D    MedMamba.py:79-119       SS_Conv_SSM   MedMamba.py:322-357
    channel_shuffle   MedMamba.py:308-320      VSSLayer      MedMamba.py:359-422
    DropPath / trunc_normal_ (timm.layers, MedMamba.py:11)   VSSM  MedMamba.py:423-515

Parameter initialisation consumes the torch RNG in the same order as the reference constructors
(incl. the "fake init" of VSSLayer, MedMamba.py:398-404), so `torch.manual_seed(s); VSSM(...)`
yields the same weights — pinned by tests/golden/kat_seed42.json.

There is no CPU path: the selective scan is `medmamba_amd.selective_scan_fn` (HIP, C ABI).
"""
import math
from functools import partial
from typing import Callable

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as checkpoint
import torch.nn.modules.module as _nn_module

from . import _lib, ops
from .ops import (PointwiseConvFn, block_split, deferred_bn_counters, block_split_infer, bn_relu_train, conv2d_bias, nchw_ln_rows,
                  nchw_ln_rows_supported, patch_merge_ln, patch_merge_ln_supported, dwconv_silu_cross, in_proj_cf, out_proj_cf, shuffle_residual,
                  ss2d_conv_core, ss2d_core)
from .selective_scan_interface import CROSS_SCAN_K_OF_G, cross_scan_fn, selective_scan_fn

trunc_normal_ = nn.init.trunc_normal_   # timm.layers.trunc_normal_ == torch.nn.init.trunc_normal_

_TWO_STREAMS = __import__("os").environ.get("MM_TWO_STREAMS", "1") == "1"   # MM_TWO_STREAMS=0: single-stream blocks
_FOLD_BN = __import__("os").environ.get("MM_FOLD_BN", "1") == "1"         # MM_FOLD_BN=0: eval() keeps the BatchNorm launches
# MM_GRAPH_CONV=1 (experimental): the conv branch of every block is replayed from hipGraphs in training (about 30 of a block's
# 80 launches become 2); the SS2D branch stays eager.  Off by default: measured in DESIGN.md §4.5.
_GRAPH_CONV = __import__("os").environ.get("MM_GRAPH_CONV", "0") == "1"
_OWN_BN = __import__("os").environ.get("MM_OWN_BN", "1") == "1"         # MM_OWN_BN=0: training-mode BatchNorm / ReLU through MIOpen / ATen
# MM_GRAPH_BLOCK_MAX_L=n (experimental, training): blocks whose images have at most n positions are replayed from hipGraphs
# (forward and backward one launch each) — the stages where the step is bound by the host's launch rate.  0 = off.
_GRAPH_BLOCK_MAX_L = int(__import__("os").environ.get("MM_GRAPH_BLOCK_MAX_L", "0"))
_SIDE_STREAMS = {}
_CONV_WARM = set()      # conv-branch input shapes whose MIOpen solver search has run (SS_Conv_SSM.forward)
_CONV_COLD_RUNS = {}
# images with at least this many positions start the side stream when the scan kernel is queued (see SS_Conv_SSM.forward).
# Default since the host stopped limiting the queues (round 3): never — the side stream starts at the split at every stage
# (29.02 / 29.12 vs 29.29 / 29.32 ms per MedMamba-S step with the 56x56 stage started late; MedMamba-B at 384^2: no difference)
_LATE_SIDE_MIN_L = int(__import__("os").environ.get("MM_LATE_SIDE_MIN_L", str(1 << 30)))


def _has_hooks(module):
    """True if `module` or anything below it carries a forward / backward hook (or a global module hook is installed).
    The fused paths below reach past sub-modules' __call__ (they read .weight / .bias directly); code that hangs hooks on
    sub-modules — e.g. Grad-CAM on `conv33conv33conv11[-2]` (test.py:101-108) — gets the module-by-module path instead."""
    _m = _nn_module
    if (_m._global_forward_hooks or _m._global_forward_pre_hooks or _m._global_backward_hooks
            or _m._global_backward_pre_hooks):
        return True
    # the sub-module list is walked once per module object (nn.Module.modules() is a recursive generator: 0.4 ms per step over
    # the 28 checks of a MedMamba-S forward); add_module / attribute assignment of a new sub-module drops the cached list
    subs = module.__dict__.get("_mm_submodules")
    if subs is None or subs[0] != len(module._modules):
        subs = module.__dict__["_mm_submodules"] = (len(module._modules), tuple(module.modules()))
    for sub in subs[1]:
        if sub._forward_hooks or sub._forward_pre_hooks or sub._backward_hooks or sub._backward_pre_hooks:
            return True
    return False


def _is_pointwise(m):
    return (isinstance(m, nn.Conv2d) and m.kernel_size == (1, 1) and m.stride == (1, 1) and m.padding == (0, 0)
            and m.dilation == (1, 1) and m.groups == 1 and m.padding_mode == "zeros")


def _conv_branch(mods, x, skip_last_bias=False):
    """Run the modules of the conv branch in order; dense convs with a bias go through ops.conv2d_bias (same MIOpen
    kernels, fast bias gradient), 1x1 convs through PointwiseConvFn (batched GEMM).
    skip_last_bias: the last module is a 1x1 conv whose bias the caller applies itself (shuffle_residual's left_bias)."""
    mods = list(mods)
    i = 0
    while i < len(mods):
        m = mods[i]
        own_bn = lambda bn, t: (_OWN_BN and type(bn) is nn.BatchNorm2d and t.is_cuda and bn.affine and t.dtype == torch.float32
                                and t.dim() == 4 and (bn.training or not bn.track_running_stats))
        if isinstance(m, nn.Conv2d) and x.is_cuda:
            if _is_pointwise(m):
                x = PointwiseConvFn.apply(x, m.weight, None if (skip_last_bias and i == len(mods) - 1) else m.bias)
            elif ops.own_conv3x3_ok(x, m):
                # our MFMA conv: bias in its epilogue, and — when one of our BatchNorms follows — that BatchNorm's statistics
                # pass too (the BatchNorm then only applies)
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                if nxt is not None and own_bn(nxt, x):
                    x, partials = ops.Conv3x3Fn.apply(x, m.weight, m.bias, True)
                    relu = i + 2 < len(mods) and type(mods[i + 2]) is nn.ReLU
                    x = bn_relu_train(x, nxt, relu, partials)
                    i += 2 if relu else 1
                else:
                    x = ops.Conv3x3Fn.apply(x, m.weight, m.bias, False)
            else:
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                if nxt is not None and own_bn(nxt, x) and nxt.track_running_stats and ops.conv2d_bias_ok(x, m):
                    # conv -> BatchNorm (MedMamba.py:339-340, 342-343): the conv's bias changes nothing but the BatchNorm's running
                    # mean — the conv runs without it and the BatchNorm kernel accounts for it (no bias-add pass)
                    x = conv2d_bias(x, m, add_bias=False)
                    relu = i + 2 < len(mods) and type(mods[i + 2]) is nn.ReLU
                    x = bn_relu_train(x, nxt, relu, pre_bias=m.bias)
                    i += 2 if relu else 1
                else:
                    x = conv2d_bias(x, m)
        elif own_bn(m, x):
            # training-mode BatchNorm through our kernels; a directly following nn.ReLU is folded into them
            relu = i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU
            x = bn_relu_train(x, m, relu)
            i += 1 if relu else 0
        else:
            x = m(x)
        i += 1
    return x


class _ConvBody(nn.Module):
    """The conv branch of a block (MedMamba.py:338-346, without the trailing ReLU that shuffle_residual applies) as a module
    of its own — the unit that MM_GRAPH_CONV=1 records into a pair of hipGraphs (forward / backward)."""

    def __init__(self, mods):
        super().__init__()
        self.mods = nn.ModuleList(mods)       # the SAME module objects as in SS_Conv_SSM.conv33conv33conv11

    def forward(self, x):
        return _conv_branch(self.mods, x)


def _side_stream(device):
    s = _SIDE_STREAMS.get(device)
    if s is None:
        s = _SIDE_STREAMS[device] = torch.cuda.Stream(device=device)
    return s


class DropPath(nn.Module):
    """Per-sample stochastic depth (timm.layers.DropPath semantics; MedMamba.py:335)."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def factor(self, x):
        """Per-sample factor mask / keep_prob of shape (B, 1, ..., 1) — the same random draw as forward() — or None."""
        if self.drop_prob == 0.0 or not self.training:
            return None
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return mask

    def forward(self, x):
        mask = self.factor(x)
        return x if mask is None else x * mask

    def __repr__(self):
        return f"timm.DropPath({self.drop_prob})"


class PatchEmbed2D(nn.Module):
    """Image -> (B, H/4, W/4, C) patch tokens: strided conv + LayerNorm (MedMamba.py:54-76)."""

    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None, **kwargs):
        super().__init__()
        if isinstance(patch_size, int):
            patch_size = (patch_size, patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        # the strided conv through ConvBiasFn: same MIOpen kernels, the bias gradient from mm_channel_sum_nchw (autograd's own is an
        # ATen reduction over the (B, C, H/4, W/4) gradient: 59 us at 64 x 96 x 56 x 56, at the very end of backward where nothing hides it)
        x = conv2d_bias(x, self.proj) if (not _has_hooks(self.proj) and ops.conv2d_bias_ok(x, self.proj)) else self.proj(x)
        n = self.norm
        if (x.is_cuda and x.dtype == torch.float32 and type(n) is nn.LayerNorm and n.elementwise_affine and n.bias is not None
                and not _has_hooks(self) and nchw_ln_rows_supported(x.shape[1], x.shape[2] * x.shape[3])):
            return nchw_ln_rows(x, n.weight, n.bias, n.eps)          # permute + LayerNorm, one pass over HBM each way
        x = x.permute(0, 2, 3, 1)
        return x if n is None else n(x)


class PatchMerging2D(nn.Module):
    """2x2 patch merge: gather -> LayerNorm(4C) -> Linear(4C, 2C) (MedMamba.py:79-119).
    Odd H/W are cropped to the even part, with the reference's printed warning (MedMamba.py:97-111)."""

    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x):
        B, H, W, C = x.shape
        if (W % 2 != 0) or (H % 2 != 0):
            print(f"Warning, x.shape {x.shape} is not match even ===========", flush=True)
        h2, w2 = H // 2, W // 2
        nm = self.norm
        if (x.is_cuda and h2 > 0 and w2 > 0 and type(nm) is nn.LayerNorm and nm.elementwise_affine and nm.bias is not None
                and tuple(nm.normalized_shape) == (4 * C,) and x.dtype == torch.float32 and not _has_hooks(self)
                and patch_merge_ln_supported(C)):
            # gather + LayerNorm(4C) in one HIP pass each way (the gathered tensor never exists); :117 is the GEMM below
            return self.reduction(patch_merge_ln(x, nm.weight, nm.bias, nm.eps))
        x = x[:, :2 * h2, :2 * w2, :].reshape(B, h2, 2, w2, 2, C)
        # reference channel order: (row even, col even), (row odd, col even), (row even, col odd), (row odd, col odd)
        x = x.permute(0, 1, 3, 4, 2, 5).reshape(B, h2, w2, 4 * C)
        return self.reduction(self.norm(x))


def channel_shuffle(x, groups: int):
    """(…, g, c/g) -> (…, c/g, g) on the last dim (MedMamba.py:308-320)."""
    b, h, w, c = x.shape
    return x.view(b, h, w, groups, c // groups).transpose(3, 4).reshape(b, h, w, c)


class SS2D(nn.Module):
    """2-D selective scan block (MedMamba.py:123-305): in_proj -> depthwise conv3x3 + SiLU ->
    4-direction scan (cross-scan, x/dt projections, selective scan, cross-merge) -> LayerNorm ->
    gate with SiLU(z) -> out_proj."""

    def __init__(self, d_model, d_state=16, d_conv=3, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, dropout=0., conv_bias=True, bias=False,
                 device=None, dtype=None, **kwargs):
        fk = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner = int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        K, C = 4, self.dt_rank + 2 * d_state

        self.in_proj = nn.Linear(d_model, 2 * self.d_inner, bias=bias, **fk)
        self.conv2d = nn.Conv2d(self.d_inner, self.d_inner, groups=self.d_inner, bias=conv_bias, kernel_size=d_conv,
                                padding=(d_conv - 1) // 2, **fk)
        self.act = nn.SiLU()
        # RNG order of the reference: 4 x_proj Linears, then 4 x (dt weight, dt bias rand) — MedMamba.py:164-181
        xp = [nn.Linear(self.d_inner, C, bias=False, **fk).weight for _ in range(K)]
        self.x_proj_weight = nn.Parameter(torch.stack(xp, dim=0))                      # (4, R+2N, d_inner)
        dts = [self.dt_init(self.dt_rank, self.d_inner, dt_scale, dt_init, dt_min, dt_max, dt_init_floor, **fk)
               for _ in range(K)]
        self.dt_projs_weight = nn.Parameter(torch.stack([t.weight for t in dts], dim=0))   # (4, d_inner, R)
        self.dt_projs_bias = nn.Parameter(torch.stack([t.bias for t in dts], dim=0))       # (4, d_inner)
        self.A_logs = self.A_log_init(d_state, self.d_inner, copies=K, merge=True)         # (4*d_inner, N)
        self.Ds = self.D_init(self.d_inner, copies=K, merge=True)                          # (4*d_inner,)
        self.forward_core = self.forward_corev0
        self.out_norm = nn.LayerNorm(self.d_inner)
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=bias, **fk)
        self.dropout = nn.Dropout(dropout) if dropout > 0. else None

    @staticmethod
    def dt_init(dt_rank, d_inner, dt_scale=1.0, dt_init="random", dt_min=0.001, dt_max=0.1, dt_init_floor=1e-4,
                **fk):
        """dt projection whose bias is softplus^-1 of a log-uniform dt in [dt_min, dt_max] (MedMamba.py:193-218)."""
        proj = nn.Linear(dt_rank, d_inner, bias=True, **fk)
        std = dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(proj.weight, std)
        elif dt_init == "random":
            nn.init.uniform_(proj.weight, -std, std)
        else:
            raise NotImplementedError
        dt = torch.exp(torch.rand(d_inner, **fk) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min))
        dt = dt.clamp(min=dt_init_floor)
        with torch.no_grad():
            proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))
        proj.bias._no_reinit = True
        return proj

    @staticmethod
    def A_log_init(d_state, d_inner, copies=1, device=None, merge=True):
        """A = -(1..N) per channel, stored as log (MedMamba.py:220-235)."""
        a = torch.log(torch.arange(1, d_state + 1, dtype=torch.float32, device=device)).repeat(d_inner, 1)
        if copies > 1:
            a = a.unsqueeze(0).repeat(copies, 1, 1)
            if merge:
                a = a.flatten(0, 1)
        p = nn.Parameter(a.contiguous())
        p._no_weight_decay = True
        return p

    @staticmethod
    def D_init(d_inner, copies=1, device=None, merge=True):
        """Skip parameter D = 1 (MedMamba.py:237-247)."""
        d = torch.ones(d_inner, device=device)
        if copies > 1:
            d = d.unsqueeze(0).repeat(copies, 1)
            if merge:
                d = d.flatten(0, 1)
        p = nn.Parameter(d)
        p._no_weight_decay = True
        return p

    def forward_corev0(self, x):
        """x (B, D, H, W) -> the four un-flipped / un-transposed direction outputs (MedMamba.py:249-286)."""
        self.selective_scan = selective_scan_fn
        B, _, H, W = x.shape
        L, K = H * W, 4
        x_hwwh = torch.stack([x.view(B, -1, L), x.transpose(2, 3).contiguous().view(B, -1, L)], dim=1)
        xs = torch.cat([x_hwwh, x_hwwh.flip(-1)], dim=1)                                   # (B, 4, D, L)
        x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, self.x_proj_weight)
        dts, Bs, Cs = torch.split(x_dbl, [self.dt_rank, self.d_state, self.d_state], dim=2)
        dts = torch.einsum("bkrl,kdr->bkdl", dts, self.dt_projs_weight)
        out_y = self.selective_scan(
            xs.float().view(B, -1, L), dts.contiguous().float().view(B, -1, L),
            -torch.exp(self.A_logs.float()).view(-1, self.d_state), Bs.float(), Cs.float(),
            self.Ds.float().view(-1), z=None, delta_bias=self.dt_projs_bias.float().view(-1),
            delta_softplus=True, return_last_state=False).view(B, K, -1, L)
        assert out_y.dtype == torch.float
        inv_y = out_y[:, 2:4].flip(-1)
        wh_y = out_y[:, 1].view(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
        invwh_y = inv_y[:, 1].view(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
        return out_y[:, 0], inv_y[:, 0], wh_y, invwh_y

    def forward_core_fused(self, x):
        """Same result as summing the four outputs of forward_corev0, without materialising the cross-scan:
        x (B, D, H, W) -> y (B, H, W, D).  The scan kernel reads the row-major and the column-major image
        (one transpose copy) and walks the two backward directions in reverse, so no flipped / 4x-stacked tensor
        and no un-flip exists (MedMamba.py:256-257, 282-286, 298); projections are plain batched GEMMs."""
        B, D, H, W = x.shape
        L, R, N = H * W, self.dt_rank, self.d_state
        pk = lambda t: torch.stack([t[k] for k in CROSS_SCAN_K_OF_G], dim=0)   # kernel direction g -> reference k
        u2 = x.new_empty(B, 2, D, L)
        u2[:, 0] = x.view(B, D, L)
        u2[:, 1].view(B, D, W, H).copy_(x.transpose(2, 3))
        Wx = pk(self.x_proj_weight).reshape(1, 2, 2 * (R + 2 * N), D)
        x_dbl = torch.matmul(Wx, u2).view(B, 4, R + 2 * N, L)                              # :259
        dts = torch.matmul(pk(self.dt_projs_weight).unsqueeze(0), x_dbl[:, :, :R])         # :262  (B,4,D,L)
        y2 = cross_scan_fn(
            u2.view(B, 2 * D, L), dts.view(B, 4 * D, L),
            -torch.exp(pk(self.A_logs.float().view(4, D, N))).view(4 * D, N),
            x_dbl[:, :, R:R + N], x_dbl[:, :, R + N:],
            pk(self.Ds.float().view(4, D)).reshape(-1), pk(self.dt_projs_bias.float()).reshape(-1)).view(B, 2, D, L)
        y = y2[:, 0].view(B, D, H, W).permute(0, 2, 3, 1) + y2[:, 1].view(B, D, W, H).permute(0, 3, 2, 1)
        return y.contiguous()

    def forward_cf(self, x, prescan_event=None):
        """(B, H, W, d_model) -> (B, d_model, H*W), channel-first.  MI355X layout: everything between in_proj and
        out_proj lives in channel-first planes (B, channel, H*W) — stored batch-major for long sequences (projections =
        batched GEMMs with a broadcast weight) and channel-major (channel, B, H*W) for short ones (projections = single
        GEMMs over B*H*W columns; ops.channel_major) — the depthwise conv, the scan, the cross-merge, out_norm and the gate
        are plane-wise HIP kernels taking (batch, channel) strides, so none of the reference's permute / stack / flip /
        transpose copies (MedMamba.py:294-299) exists."""
        B, H, W, _ = x.shape
        L, D, R, N = H * W, self.d_inner, self.dt_rank, self.d_state
        cv = self.conv2d
        conv_ok = (cv.kernel_size == (3, 3) and cv.padding == (1, 1) and cv.stride == (1, 1) and cv.dilation == (1, 1)
                   and cv.groups == D and cv.padding_mode == "zeros" and _lib.lib().mm_dwconv_silu_cross_supported(H, W))
        if conv_ok and ops.ss2d_branch_native_ok(x, self.in_proj, self.out_proj, cv, (
                self.x_proj_weight, self.dt_projs_weight, self.dt_projs_bias, self.A_logs, self.Ds, self.out_norm.weight,
                self.out_norm.bias)):
            # training: the whole branch as ONE autograd node, sequenced in C++ (csrc_host/ss2d_host.cpp) — same launches as below
            out = ops.ss2d_branch(x.reshape(B, L, -1), self.in_proj.weight, cv.weight, cv.bias, self.x_proj_weight,
                                  self.dt_projs_weight, self.dt_projs_bias, self.A_logs, self.Ds, self.out_norm.weight,
                                  self.out_norm.bias, self.out_proj.weight, H, W, self.out_norm.eps, prescan_event=prescan_event)
            return out if self.dropout is None else self.dropout(out)
        x_cf, z_cf = in_proj_cf(x.reshape(B, L, -1), self.in_proj.weight, self.in_proj.bias)  # :291-292, (B, D, L) each
        # depthwise conv + SiLU (:294-295) writing both image orders of :256, projections (:259-262), A = -exp(A_logs)
        # (:271), scan, merge, out_norm and gate (:273-301) — one autograd Function; the parameters go in as the module
        # holds them, the kernel-order packing is one launch inside
        if conv_ok:
            y_cf = ss2d_conv_core(x_cf, cv.weight, cv.bias, self.x_proj_weight, self.dt_projs_weight, self.dt_projs_bias,
                                  self.A_logs, self.Ds, z_cf, self.out_norm.weight, self.out_norm.bias, H, W,
                                  self.out_norm.eps, prescan_event=prescan_event)
        else:   # any other d_conv, or planes beyond the fused kernels' LDS budget (decided HERE, with the backward's need, so
                # that a pass never fails halfway): the conv through MIOpen, then the core on its two image orders
            xc = self.act(cv(x_cf.reshape(B, D, H, W)))
            u2 = torch.stack([xc.reshape(B, D, L), xc.transpose(2, 3).reshape(B, D, L)], 1).reshape(B, 2 * D, L)
            y_cf = ss2d_core(u2, self.x_proj_weight, self.dt_projs_weight, self.dt_projs_bias, self.A_logs, self.Ds,
                             z_cf, self.out_norm.weight, self.out_norm.bias, H, W, self.out_norm.eps,
                             prescan_event=prescan_event)
        out = out_proj_cf(y_cf, self.out_proj.weight, self.out_proj.bias)                    # :302, (B, d_model, L)
        return out if self.dropout is None else self.dropout(out)

    def forward_modules(self, x):
        """MedMamba.py:288-305 module by module (every sub-module through its __call__, so hooks fire); the scan itself
        is still the HIP operator (forward_corev0 -> selective_scan_fn)."""
        B, H, W, _ = x.shape
        x, z = self.in_proj(x).chunk(2, dim=-1)
        x = self.act(self.conv2d(x.permute(0, 3, 1, 2).contiguous()))
        y1, y2, y3, y4 = self.forward_core(x)
        y = (y1 + y2 + y3 + y4).transpose(1, 2).contiguous().view(B, H, W, -1)
        out = self.out_proj(self.out_norm(y) * F.silu(z))
        return out if self.dropout is None else self.dropout(out)

    def forward(self, x, **kwargs):
        """(B, H, W, d_model) -> (B, H, W, d_model) (MedMamba.py:288-305)."""
        if _has_hooks(self) or self.d_state != 16:
            return self.forward_modules(x)
        B, H, W, _ = x.shape
        return self.forward_cf(x).transpose(1, 2).reshape(B, H, W, -1)


class SS_Conv_SSM(nn.Module):
    """One MedMamba block (MedMamba.py:322-357): channel halves -> {conv branch | LayerNorm + SS2D} ->
    concat -> channel shuffle -> residual."""

    def __init__(self, hidden_dim: int = 0, drop_path: float = 0,
                 norm_layer: Callable[..., nn.Module] = partial(nn.LayerNorm, eps=1e-6),
                 attn_drop_rate: float = 0, d_state: int = 16, **kwargs):
        super().__init__()
        half = hidden_dim // 2
        self.ln_1 = norm_layer(half)
        self.self_attention = SS2D(d_model=half, dropout=attn_drop_rate, d_state=d_state, **kwargs)
        self.drop_path = DropPath(drop_path)
        self.conv33conv33conv11 = nn.Sequential(
            nn.BatchNorm2d(half),
            nn.Conv2d(half, half, kernel_size=3, stride=1, padding=1),
            nn.BatchNorm2d(half),
            nn.ReLU(),
            nn.Conv2d(half, half, kernel_size=3, stride=1, padding=1),
            nn.BatchNorm2d(half),
            nn.ReLU(),
            nn.Conv2d(half, half, kernel_size=1, stride=1),
            nn.ReLU(),
        )

    # ---- inference: BatchNorm folded into its neighbours (consumers: test.py:76-108, app_streamlit_demo.py) ----
    def _foldable(self):
        m = list(self.conv33conv33conv11)
        kinds = (nn.BatchNorm2d, nn.Conv2d, nn.BatchNorm2d, nn.ReLU, nn.Conv2d, nn.BatchNorm2d, nn.ReLU, nn.Conv2d, nn.ReLU)
        if len(m) != len(kinds) or not all(isinstance(a, k) for a, k in zip(m, kinds)):
            return False
        bns, convs = (m[0], m[2], m[5]), (m[1], m[4], m[7])
        ok_bn = all(b.track_running_stats and b.running_mean is not None and b.affine for b in bns)
        ok_cv = all(c.groups == 1 and c.padding_mode == "zeros" and c.stride == (1, 1) and c.dilation == (1, 1) and c.bias is not None
                    for c in convs) and _is_pointwise(m[7])
        return ok_bn and ok_cv

    def train(self, mode: bool = True):
        # MIOpen's training-mode BatchNorm updates running_mean / running_var without bumping their version counters, so the
        # fold below cannot see that change: every train() / eval() switch drops it (it is rebuilt on the next inference pass)
        self._fold_cache = None
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self._fold_cache = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _eval_fold(self):
        """Eval-mode constants of the conv branch (MedMamba.py:338-346) with every BatchNorm2d folded away, computed in
        fp64 and cast once:  BN1 -> per-channel affine applied while the block prologue transposes the left half (before the
        first conv's zero padding: exact at the border);  conv -> BN2 / conv -> BN3 -> into the conv's weights and bias.
        Cached per block; any in-place change of a parameter or running statistic (optimizer step, load_state_dict, a
        training pass) bumps that tensor's version counter and rebuilds it."""
        m = self.conv33conv33conv11
        tensors = [m[0].weight, m[0].bias, m[0].running_mean, m[0].running_var, m[1].weight, m[1].bias, m[2].weight, m[2].bias,
                   m[2].running_mean, m[2].running_var, m[4].weight, m[4].bias, m[5].weight, m[5].bias, m[5].running_mean,
                   m[5].running_var]
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        cache = getattr(self, "_fold_cache", None)
        if cache is not None and cache[0] == key:
            return cache[1]

        def affine(bn):
            s = bn.weight.detach().double() / torch.sqrt(bn.running_var.double() + bn.eps)
            return s, bn.bias.detach().double() - bn.running_mean.double() * s

        def fold_conv(conv, bn):
            s, t = affine(bn)
            w = (conv.weight.detach().double() * s[:, None, None, None]).float().contiguous()
            return w, (conv.bias.detach().double() * s + t).float().contiguous()

        s1, t1 = affine(m[0])
        fold = dict(left_affine=torch.cat([s1, t1]).float().contiguous(), a=fold_conv(m[1], m[2]), b=fold_conv(m[4], m[5]),
                    pad_a=m[1].padding, pad_b=m[4].padding)
        self._fold_cache = (key, fold)
        return fold

    @staticmethod
    def _conv_relu(x, w, b, padding):
        """conv + bias + ReLU as ONE MIOpen fusion plan where this PyTorch build exposes it (inference only)."""
        try:
            return torch.ops.aten.miopen_convolution_relu(x, w, b, [1, 1], list(padding), [1, 1], 1)
        except (RuntimeError, AttributeError):
            return F.relu_(F.conv2d(x, w, b, padding=padding))

    def _forward_infer(self, input):
        """eval() + no_grad: 3 BatchNorm, 2 bias and 2 ReLU launches of the conv branch disappear (folded as _eval_fold says;
        the trailing ReLU is applied by shuffle_residual as in training)."""
        fold = self._eval_fold()
        conv = self.conv33conv33conv11
        left, right_n = block_split_infer(input, self.ln_1.weight, self.ln_1.bias, self.ln_1.eps, fold["left_affine"])

        def conv_body(t):
            t = self._conv_relu(t, *fold["a"], fold["pad_a"])
            t = self._conv_relu(t, *fold["b"], fold["pad_b"])
            return PointwiseConvFn.apply(t, conv[7].weight, None)                 # pre-activation of the trailing ReLU, without
                                                                                  # the bias (shuffle_residual adds it)

        warm_key = (tuple(left.shape), left.device, "infer")      # see forward(): MIOpen's solver search runs on an idle GPU
        if _TWO_STREAMS and warm_key in _CONV_WARM:
            main = torch.cuda.current_stream()
            side = _side_stream(input.device)
            left.record_stream(side)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                left = conv_body(left)
            x_cf = self.self_attention.forward_cf(right_n)
            main.wait_stream(side)
            left.record_stream(main)
        else:
            x_cf = self.self_attention.forward_cf(right_n)
            left = conv_body(left)
            _CONV_WARM.add(warm_key)
        return shuffle_residual(left, x_cf, input, channel_first=True, ssm_scale=None, left_relu=True, left_bias=conv[7].bias)

    def _graphed_conv_body(self, mods, left):
        """torch.cuda.make_graphed_callables over the conv branch for this input shape (built once per shape).  The BatchNorm
        running statistics that its warm-up / capture passes touch are restored afterwards."""
        cache = self.__dict__.setdefault("_conv_graphs", {})
        key = (tuple(left.shape), left.device)
        g = cache.get(key)
        if g is None:
            body = _ConvBody(mods)
            bufs = [(b, b.detach().clone()) for b in body.buffers()]
            sample = torch.randn_like(left).requires_grad_()
            with ops.immediate_bn_counters():
                g = torch.cuda.make_graphed_callables(body, (sample,))
            with torch.no_grad():
                for b, saved in bufs:
                    b.copy_(saved)
            for prm in body.parameters():
                prm.grad = None
            cache[key] = g
        return g

    def forward_modules(self, input):
        """MedMamba.py:349-357 module by module (hooks on any sub-module fire; used only when hooks are present)."""
        left, right = input.chunk(2, dim=-1)
        right = self.drop_path(self.self_attention(self.ln_1(right)))
        left = self.conv33conv33conv11(left.permute(0, 3, 1, 2).contiguous()).permute(0, 2, 3, 1).contiguous()
        return channel_shuffle(torch.cat((left, right), dim=-1), groups=2) + input

    def forward(self, input):
        # the fused plane-wise core is built for the reference's d_state = 16 (MedMamba.py:329,457); any other d_state takes the
        # reference's own op chain around selective_scan_fn, which runs it as 16-state slices on the same kernels
        if _has_hooks(self) or not isinstance(self.drop_path, DropPath) or self.self_attention.d_state != 16:
            return self.forward_modules(input)
        ln_ok = isinstance(self.ln_1, nn.LayerNorm) and self.ln_1.elementwise_affine and self.ln_1.bias is not None \
            and input.shape[-1] % 2 == 0 and input.shape[-1] // 2 <= 512
        if (_FOLD_BN and not self.training and not torch.is_grad_enabled() and input.is_cuda and ln_ok and self._foldable()
                and not any(bn.training for bn in (self.conv33conv33conv11[0], self.conv33conv33conv11[2], self.conv33conv33conv11[5]))):
            return self._forward_infer(input)
        if isinstance(self.ln_1, nn.LayerNorm) and self.ln_1.elementwise_affine and self.ln_1.bias is not None \
                and input.shape[-1] % 2 == 0 and input.shape[-1] // 2 <= 512:     # block_split keeps a row in registers
            # chunk + permute(0,3,1,2).contiguous() + ln_1 (MedMamba.py:350-352) in one fused HIP prologue
            left, right_n, input = block_split(input, self.ln_1.weight, self.ln_1.bias, self.ln_1.eps)
        else:                                                  # any other norm_layer: the reference's own op chain
            left, right = input.chunk(2, dim=-1)
            left, right_n = left.permute(0, 3, 1, 2).contiguous(), self.ln_1(right)
        conv = self.conv33conv33conv11
        fold_relu = isinstance(conv[-1], nn.ReLU)              # the trailing ReLU (:347) is applied by shuffle_residual
        mods = list(conv)[:-1] if fold_relu else list(conv)
        graph_conv = _GRAPH_CONV and self.training and torch.is_grad_enabled() and input.is_cuda
        # the closing 1x1 conv's bias (MedMamba.py:345) is added by shuffle_residual in front of the ReLU it applies anyway: the
        # conv is a bias-free GEMM (a broadcast bias costs baddbmm a fill pass over the whole output first)
        last = mods[-1] if mods else None
        defer_bias = (fold_relu and not graph_conv and input.is_cuda and last is not None and _is_pointwise(last)
                      and last.bias is not None)
        def conv_body(t):
            if defer_bias and _OWN_BN:       # the reference's conv branch in training mode: one C++-sequenced autograd node
                y = ops.conv_branch_native(mods, t)
                if y is not None:
                    return y
            return _conv_branch(mods, t, skip_last_bias=defer_bias)

        if graph_conv:
            conv_body = self._graphed_conv_body(mods, left)
        # PyTorch asks MIOpen to FIND a solver at the first call of every conv configuration, and MIOpen answers by timing its
        # candidates on the spot.  With the scan running beside them the timings are noise (stage-3 convs landed on an implicit
        # GEMM + 2 layout transposes instead of Winograd on one box in three): the first pass over a shape — forward and, in
        # training, the backward that follows it — therefore runs the block on ONE stream, the overlap starts with the next.
        warm_key = (tuple(left.shape), left.device, torch.is_grad_enabled())
        cold = input.is_cuda and warm_key not in _CONV_WARM
        if _TWO_STREAMS and input.is_cuda and not cold:
            # the conv branch and the SS2D branch are independent (MedMamba.py:351-353): run the conv branch on a side
            # HIP stream so that its MIOpen kernels overlap the scan (autograd replays the backward on the same streams)
            main = torch.cuda.current_stream()
            side = _side_stream(input.device)
            left.record_stream(side)       # allocated on `main`, read by the side stream in forward and backward
            if left.shape[2] * left.shape[3] >= _LATE_SIDE_MIN_L:
                # (off by default, see _LATE_SIDE_MIN_L) long sequences: queue the SS2D branch first and let the side stream start
                # when the scan kernel does — the conv kernels then share the GPU with the latency-bound scan instead of with
                # the bandwidth-bound projections before it.  For short sequences the late start only delays the join below.
                ev = torch.cuda.Event()
                x_cf = self.self_attention.forward_cf(right_n, prescan_event=ev)             # (B, C/2, H*W)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    left = conv_body(left)                                                   # stays NCHW
            else:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    left = conv_body(left)                                                   # stays NCHW
                x_cf = self.self_attention.forward_cf(right_n)                               # (B, C/2, H*W)
            main.wait_stream(side)
            left.record_stream(main)
        else:
            x_cf = self.self_attention.forward_cf(right_n)                                   # (B, C/2, H*W)
            left = conv_body(left)                                                           # stays NCHW
            if cold:
                n = _CONV_COLD_RUNS[warm_key] = _CONV_COLD_RUNS.get(warm_key, 0) + 1
                if left.requires_grad and n <= 64:     # every forward of the first step precedes its first backward: warm from
                    left.register_hook(lambda g, k=warm_key: _CONV_WARM.add(k))        # then on (64: a caller that never
                else:                                                                  # differentiates is not kept cold for ever)
                    _CONV_WARM.add(warm_key)
        # trailing ReLU + drop_path + permute back + cat + channel_shuffle(groups=2) + residual (MedMamba.py:347, 353-357)
        # fused in one HIP kernel
        scale = getattr(self, "_dp_factor", None)      # drawn for all blocks at once by VSSM.forward_backbone
        if scale is None:
            scale = self.drop_path.factor(x_cf)
        return shuffle_residual(left, x_cf, input, channel_first=True, ssm_scale=scale, left_relu=fold_relu,
                                left_bias=last.bias if defer_bias else None)


class _GraphableBlock(nn.Module):
    """SS_Conv_SSM with its per-step DropPath factor as an explicit tensor argument (what a hipGraph capture needs)."""

    def __init__(self, blk):
        super().__init__()
        self.blk = blk

    def forward(self, x, factor):
        prev = getattr(self.blk, "_dp_factor", None)
        self.blk._dp_factor = factor
        try:
            return self.blk(x)
        finally:
            self.blk._dp_factor = prev


class VSSLayer(nn.Module):
    """One stage: `depth` blocks then an optional downsample (MedMamba.py:359-422)."""

    def __init__(self, dim, depth, attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, downsample=None,
                 use_checkpoint=False, d_state=16, **kwargs):
        super().__init__()
        self.dim = dim
        self.use_checkpoint = use_checkpoint
        self.blocks = nn.ModuleList([
            SS_Conv_SSM(hidden_dim=dim, drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                        norm_layer=norm_layer, attn_drop_rate=attn_drop, d_state=d_state)
            for i in range(depth)])
        # The reference draws (and discards) a kaiming-uniform sample per out_proj.weight here, once per
        # module that owns one under that name (MedMamba.py:398-404).  Only the RNG stream matters.
        for m in self.modules():
            for name, p in m.named_parameters():
                if name == "out_proj.weight":
                    nn.init.kaiming_uniform_(p.clone().detach_(), a=math.sqrt(5))
        self.downsample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def _graphed(self, blk, x):
        """torch.cuda.make_graphed_callables over one whole block (both branches, both streams) for this input shape: the
        block's ~80 launches per step become 2 graph launches.  The DropPath factor of the step goes in as a tensor argument
        (a graph would otherwise replay the factor tensor of its capture).  BatchNorm statistics touched by the warm-up /
        capture passes are restored."""
        cache = blk.__dict__.setdefault("_block_graphs", {})
        key = (tuple(x.shape), x.device)
        g = cache.get(key)
        if g is None:
            wrap = _GraphableBlock(blk)
            bufs = [(b, b.detach().clone()) for b in blk.buffers()]
            sx = torch.randn_like(x).requires_grad_()
            sf = torch.ones(x.shape[0], device=x.device, dtype=torch.float32)
            with ops.immediate_bn_counters():
                g = torch.cuda.make_graphed_callables(wrap, (sx, sf))
            with torch.no_grad():
                for b, saved in bufs:
                    b.copy_(saved)
            for prm in blk.parameters():
                prm.grad = None
            cache[key] = g
        f = getattr(blk, "_dp_factor", None)
        if f is None:
            f = blk.drop_path.factor(x)
            f = torch.ones(x.shape[0], device=x.device, dtype=torch.float32) if f is None else f.reshape(-1)
        return g(x, f)

    def forward(self, x):
        graph = (_GRAPH_BLOCK_MAX_L > 0 and self.training and torch.is_grad_enabled() and x.is_cuda and not self.use_checkpoint
                 and x.shape[1] * x.shape[2] <= _GRAPH_BLOCK_MAX_L and x.requires_grad)
        for blk in self.blocks:
            if graph and not _has_hooks(blk):
                x = self._graphed(blk, x)
            else:
                x = checkpoint.checkpoint(blk, x) if self.use_checkpoint else blk(x)
        return x if self.downsample is None else self.downsample(x)


class VSSM(nn.Module):
    """MedMamba classifier (MedMamba.py:423-515). T/S/B/Te are just depths/dims (train.py:179-182)."""

    def __init__(self, patch_size=4, in_chans=3, num_classes=1000, depths=[2, 2, 4, 2], dims=[96, 192, 384, 768],
                 d_state=16, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1, norm_layer=nn.LayerNorm,
                 patch_norm=True, use_checkpoint=False, **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.num_layers = len(depths)
        if isinstance(dims, int):
            dims = [int(dims * 2 ** i) for i in range(self.num_layers)]
        self.embed_dim, self.num_features, self.dims = dims[0], dims[-1], dims
        self.patch_embed = PatchEmbed2D(patch_size=patch_size, in_chans=in_chans, embed_dim=self.embed_dim,
                                        norm_layer=norm_layer if patch_norm else None)
        self.ape = False
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(VSSLayer(
                dim=dims[i], depth=depths[i], d_state=math.ceil(dims[0] / 6) if d_state is None else d_state,
                drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])],
                norm_layer=norm_layer, downsample=PatchMerging2D if i < self.num_layers - 1 else None,
                use_checkpoint=use_checkpoint))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.apply(self._init_weights)
        for m in self.modules():                      # MedMamba.py:471-473 (every Conv2d, incl. depthwise)
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'absolute_pos_embed'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table'}

    def _draw_drop_path(self, batch, device):
        """All blocks' DropPath factors (mask / keep_prob, MedMamba.py:335) of this step in two launches instead of two per
        block; the blocks pick theirs up in SS_Conv_SSM.forward.  Only on HIP tensors in training mode and without
        activation checkpointing (a recomputed block must see the mask of its first run)."""
        # only blocks that would draw themselves: in training mode (a frozen block put into eval() inside a training model
        # drops nothing, like the reference's per-block DropPath) and with 0 < drop_prob < 1 (keep_prob 0 is not divided
        # by — DropPath.factor / timm return zeros there)
        blocks = [b for layer in self.layers if not layer.use_checkpoint for b in layer.blocks
                  if isinstance(b.drop_path, DropPath) and b.training and b.drop_path.training
                  and 0.0 < b.drop_path.drop_prob < 1.0 and b.drop_path.scale_by_keep]
        if not blocks:
            return []
        key = (batch, str(device), tuple(b.drop_path.drop_prob for b in blocks))
        if getattr(self, "_dp_key", None) != key:
            keep = torch.tensor([1.0 - b.drop_path.drop_prob for b in blocks], device=device, dtype=torch.float32)
            self._dp_key, self._dp_keep = key, keep[:, None].expand(-1, batch).contiguous()
        factors = torch.bernoulli(self._dp_keep).div_(self._dp_keep)
        for i, b in enumerate(blocks):
            b._dp_factor = factors[i]
        return blocks

    def forward_backbone(self, x):
        drawn = self._draw_drop_path(x.shape[0], x.device) if (self.training and x.is_cuda) else []
        try:
            with deferred_bn_counters():          # one multi-tensor add for all BatchNorm step counters of this forward
                x = self.pos_drop(self.patch_embed(x))
                for layer in self.layers:
                    x = layer(x)
        finally:
            for b in drawn:
                b._dp_factor = None
        return x

    def forward(self, x):
        x = self.forward_backbone(x).permute(0, 3, 1, 2)
        return self.head(torch.flatten(self.avgpool(x), start_dim=1))


MEDMAMBA_CONFIGS = {   # train.py:179-182
    "T": dict(depths=[2, 2, 4, 2], dims=[96, 192, 384, 768]),
    "S": dict(depths=[2, 2, 8, 2], dims=[96, 192, 384, 768]),
    "B": dict(depths=[2, 2, 12, 2], dims=[128, 256, 512, 1024]),
    "Te": dict(depths=[2, 3, 3, 2], dims=[96, 192, 384, 768]),
}


def medmamba(size="T", num_classes=6, **kwargs):
    return VSSM(num_classes=num_classes, **MEDMAMBA_CONFIGS[size], **kwargs)
