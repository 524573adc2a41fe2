"""Autograd wrappers of the glue kernels in libmedmamba_hip.so (csrc/glue.hip) — HIP tensors only."""
import os

import torch

from . import _lib, blas


def _stream():
    return _lib.raw_stream()


def _gemm(a, b, out=None, batch=None):
    """a @ b as ONE fp32 GEMM launch: a (m, k) or (B, m, k), b (k, n) or (B, k, n); a 2-D operand of a 3-D product is shared by
    the batch.  rocBLAS directly when a solution is recorded for the shape (blas.gemm: 7.6 us of host time instead of 17-31 us
    through the dispatcher), torch.mm / torch.bmm otherwise.  `out` must be fully overwritable (uninitialised is fine)."""
    three = a.dim() == 3 or b.dim() == 3
    if out is None:
        nb = a.shape[0] if a.dim() == 3 else (b.shape[0] if b.dim() == 3 else 0)
        out = torch.empty(((nb,) if three else ()) + (a.shape[-2], b.shape[-1]), device=a.device, dtype=torch.float32)
    if not blas.gemm(out, a, b):
        if three:
            nb = out.shape[0]
            torch.bmm(a if a.dim() == 3 else a.unsqueeze(0).expand(nb, -1, -1), b if b.dim() == 3 else b.unsqueeze(0).expand(nb, -1, -1),
                      out=out)
        else:
            torch.mm(a, b, out=out)
    return out


class _BranchTimer:
    """Diagnostic (tools/branch_balance.py; off unless enabled): device events around the forward / backward of a block's two
    branches, each on the stream the branch runs on — which branch of which block the other one waits for, WITHOUT a profiler
    slowing the host down (under rocprofv3 the host falls behind the GPU and the side stream starts late, which the unprofiled
    step does not do)."""

    def __init__(self):
        self.enabled = False
        self.records = []          # (tag, start_event, end_event)

    def start(self):
        if not self.enabled:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def stop(self, tag, start):
        if start is None:
            return
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        self.records.append((tag, start, end))


BRANCH_TIMER = _BranchTimer()


def sum_lead(t, out=None):
    """t.sum(0) — the sum over the batch behind a batched weight-gradient GEMM, over per-workgroup partial rows, ... — through
    mm_sum_lead for dense fp32 HIP tensors (one small launch with a fixed summation order instead of ATen's generic 10-us
    reduction; csrc_host sum_lead is the same call, so both launch routes give the same bits); anything else through torch."""
    n = t.shape[0] if t.dim() >= 1 else 0
    if (t.is_cuda and t.dtype == torch.float32 and t.dim() >= 2 and 2 <= n <= 4096 and t.is_contiguous() and t.numel() > 0
            and (out is None or (out.is_contiguous() and out.dtype == torch.float32 and out.numel() * n == t.numel()))):
        if out is None:
            out = torch.empty(t.shape[1:], device=t.device, dtype=torch.float32)
        ninner = t.numel() // n
        with _lib.device_guard(t.device):
            rc = _lib.lib().mm_sum_lead(t.data_ptr(), out.data_ptr(), n, ninner, ninner, _stream())
        _lib.check(rc, "mm_sum_lead")
        return out
    return t.sum(0) if out is None else torch.sum(t, 0, out=out)


def _wants_grad(*ts):
    """Does anything downstream differentiate this call?  Inside Function.forward grad mode is always off and
    ctx.needs_input_grad ignores torch.no_grad(), so the wrappers decide here and pass a plain flag: under no_grad (the usual
    eval loop with trainable parameters) the inference form — no state checkpoints, dt projection inside the scan — engages."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def _need_hip(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("medmamba_amd.ops: tensors must live on a HIP device (there is no CPU path)")


# ---- storage layout of the (B, channel, L) plane tensors between in_proj and out_proj -------------------------------
# batch-major  : storage (B, D, L) — the projections are batched GEMMs over B (each D x L) plus a sum over B for weights;
# channel-major: storage (D, B, L), handed around as the permuted view (B, D, L) with strides (L, B*L, 1) — the
#                projections are single GEMMs with B*L columns.  Measured on MI355X (tools/bench_gemm_layouts.py, S, B=64):
#                per block 714 -> 458 us at L=196 and 582 -> 342 us at L=49, but 1086 -> 1880 us at L=3136 (hipBLASLt has no
#                good split-K for K = B*L ~ 2e5), hence channel-major only for short sequences.
# Every plane kernel takes (batch stride, channel stride), so both layouts run the same code.
_CM_MAX_L = int(os.environ.get("MM_CM_MAX_L", "256"))   # longest sequence stored channel-major (tuning knob)
_LAYOUT = os.environ.get("MM_LAYOUT", "auto")          # "auto" | "bm" | "cm" (tests force both)
_PACK_FOLD = os.environ.get("MM_PACK_FOLD", "1") == "1"  # the SS2D backward's small reductions ride on mm_ss2d_pack_bwd (0: ATen sums)
_FUSE_DT = os.environ.get("MM_FUSE_DT", "1") == "1"     # inference: dt projection inside the scan kernel (MM_FUSE_DT=0: always a GEMM)


def channel_major(B, L):
    if _LAYOUT == "cm":
        return True
    if _LAYOUT == "bm":
        return False
    return B > 1 and L <= _CM_MAX_L and B * L <= 65536  # beyond that K = B*L makes the weight-gradient GEMMs split-K bound


def _planes(B, D, L, device, cm):
    """Uninitialised fp32 (B, D, L) planes in the requested storage layout."""
    if cm:
        return torch.empty((D, B, L), device=device, dtype=torch.float32).permute(1, 0, 2)
    return torch.empty((B, D, L), device=device, dtype=torch.float32)


def _rows(t):
    """fp32 with unit stride along L (any batch / channel strides)."""
    t = t.float()
    return t if t.stride(-1) == 1 else t.contiguous()


def _is_cm(t):
    B, D, L = t.shape
    return t.stride(2) == 1 and t.stride(1) == B * L and (B == 1 or t.stride(0) == L)


def _cm2d(t):
    """(D, B*L) matrix over the storage of a channel-major (B, D, L) tensor (a copy is made for any other layout)."""
    B, D, L = t.shape
    if not _is_cm(t):
        t = t.permute(1, 0, 2).contiguous().permute(1, 0, 2)
    return t.permute(1, 0, 2).reshape(D, B * L)


def _joined_halves(a, b):
    """(B, 2D, L) view over `a` followed by `b` when the two (B, D, L) tensors are the channel halves of one buffer, else None."""
    if (a.shape != b.shape or a.stride() != b.stride() or a.dtype != torch.float32 or b.dtype != torch.float32
            or a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr()
            or b.storage_offset() != a.storage_offset() + a.shape[1] * a.stride(1)):
        return None
    B, D, L = a.shape
    return a.as_strided((B, 2 * D, L), a.stride(), a.storage_offset())


def _pl(t):
    """(pointer, batch stride, channel stride) of a (B, D, L) plane tensor with unit stride along L."""
    assert t.stride(2) == 1 or t.shape[2] == 1
    return t.data_ptr(), t.stride(0), t.stride(1)


class ShuffleResidualFn(torch.autograd.Function):
    """out = channel_shuffle(cat(left_nhwc, ssm), 2) + inp   (MedMamba.py:354-357) in one kernel.
    left: (B, C/2, H, W) NCHW conv-branch output; ssm: SS2D-branch output, (B, H, W, C/2) or — channel_first —
    (B, C/2, H*W) planes in either storage layout; inp: (B, H, W, C).  Optionally folds in the two neighbours of the
    chain: left_relu — `left` is the pre-activation of the conv branch's trailing ReLU (MedMamba.py:347); ssm_scale (B,) —
    the DropPath factor mask / keep_prob of every sample (MedMamba.py:335, 353)."""

    @staticmethod
    def forward(ctx, left, ssm, inp, channel_first, ssm_scale, left_relu, left_bias=None):
        left, inp = left.float().contiguous(), inp.float().contiguous()
        left_bias = None if left_bias is None else left_bias.float().contiguous()
        ssm = _rows(ssm) if channel_first else ssm.float().contiguous()
        B, C2, H, W = left.shape
        if ssm_scale is not None:
            ssm_scale = ssm_scale.float().contiguous()
            assert ssm_scale.numel() == B
        out = torch.empty_like(inp)
        sb, sd = (ssm.stride(0), ssm.stride(1)) if channel_first else (0, 0)
        with _lib.device_guard(inp.device):
            rc = _lib.lib().mm_shuffle_residual_fwd(left.data_ptr(), ssm.data_ptr(), sb, sd, inp.data_ptr(), out.data_ptr(),
                                                    None if ssm_scale is None else ssm_scale.data_ptr(), int(bool(left_relu)),
                                                    None if left_bias is None else left_bias.data_ptr(),
                                                    B, H * W, C2, int(channel_first), _stream())
        _lib.check(rc, "mm_shuffle_residual_fwd")
        ctx.shape = (B, C2, H, W)
        ctx.cf = bool(channel_first)
        ctx.cm = bool(channel_first) and _is_cm(ssm) and B > 1
        ctx.save_for_backward(ssm_scale, left if left_relu else None, left_bias)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, C2, H, W = ctx.shape
        ssm_scale, left_pre, left_bias = ctx.saved_tensors
        dout = dout.float().contiguous()
        dleft = torch.empty((B, C2, H, W), device=dout.device, dtype=torch.float32)
        if ctx.cf:
            dssm = _planes(B, C2, H * W, dout.device, ctx.cm)      # same storage layout as ssm: out_proj's backward GEMM
            sb, sd = dssm.stride(0), dssm.stride(1)
        else:
            dssm = torch.empty((B, H, W, C2), device=dout.device, dtype=torch.float32)
            sb, sd = 0, 0
        with _lib.device_guard(dout.device):
            rc = _lib.lib().mm_shuffle_residual_bwd(dout.data_ptr(), dleft.data_ptr(), dssm.data_ptr(), sb, sd,
                                                    None if ssm_scale is None else ssm_scale.data_ptr(),
                                                    None if left_pre is None else left_pre.data_ptr(),
                                                    None if (left_bias is None or left_pre is None) else left_bias.data_ptr(),
                                                    B, H * W, C2, int(ctx.cf), _stream())
        _lib.check(rc, "mm_shuffle_residual_bwd")
        # d(left_bias) = channel sums of dleft (the bias sits in front of the ReLU whose mask dleft already carries)
        dlb = _bias_grad(dleft) if (left_bias is not None and ctx.needs_input_grad[6]) else None
        return dleft, dssm, dout, None, None, None, dlb


def shuffle_residual(left_nchw, ssm, inp_nhwc, channel_first=False, ssm_scale=None, left_relu=False, left_bias=None):
    _need_hip(left_nchw, ssm, inp_nhwc)
    return ShuffleResidualFn.apply(left_nchw, ssm, inp_nhwc, channel_first, ssm_scale, left_relu, left_bias)


class InProjFn(torch.autograd.Function):
    """SS2D.in_proj (MedMamba.py:291-292) producing the two channel-first halves without any slicing in autograd:
    x (B, L, d_model) NHWC rows, weight (2D, d_model)[, bias (2D)] -> (x_cf, z_cf) = two (B, D, L) views of one buffer,
    stored batch-major (a batched GEMM) or channel-major (one GEMM with B*L columns) — see channel_major().
    The backward consumes both gradients directly (no zero-fill / copy / add of sliced activations or weights)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        Bsz, L, dm = x.shape
        D = weight.shape[0] // 2
        cm = channel_major(Bsz, L)
        if cm:
            x = x.contiguous()
            xz = _gemm(weight, x.view(Bsz * L, dm).t())                                       # (2D, B*L)
            if bias is not None:
                xz += bias[:, None]
            xz = xz.view(2 * D, Bsz, L).permute(1, 0, 2)
        else:
            xz = _gemm(weight, x.transpose(1, 2))                                             # (B, 2D, L)
            if bias is not None:
                xz += bias[:, None]
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.cm = cm
        return xz[:, :D], xz[:, D:]

    @staticmethod
    def backward(ctx, dx_cf, dz_cf):
        x, weight = ctx.saved_tensors
        Bsz, L, dm = x.shape
        D = weight.shape[0] // 2
        w0, w1 = weight[:D], weight[D:]
        dw = torch.empty_like(weight)
        g = _joined_halves(dx_cf, dz_cf)
        if g is not None:              # the SS2D core hands both gradients back as halves of one (B, 2D, L) buffer
            if ctx.cm and _is_cm(g):
                g2, x2 = _cm2d(g), x.view(Bsz * L, dm)
                dx = _gemm(g2.t(), weight).view(Bsz, L, dm)
                _gemm(g2, x2, out=dw)
            else:
                dx = _gemm(g.transpose(1, 2), weight)
                sum_lead(_gemm(g, x), out=dw)
        elif ctx.cm:
            gx, gz, x2 = _cm2d(dx_cf), _cm2d(dz_cf), x.view(Bsz * L, dm)                         # (D, B*L) each
            dx = torch.mm(gx.t(), w0)
            dx.addmm_(gz.t(), w1)
            dx = dx.view(Bsz, L, dm)
            torch.mm(gx, x2, out=dw[:D])
            torch.mm(gz, x2, out=dw[D:])
        else:
            # d x (B, L, dm) = dx_cf^T @ W[:D] + dz_cf^T @ W[D:]
            dx = torch.bmm(dx_cf.transpose(1, 2), w0.unsqueeze(0).expand(Bsz, -1, -1))
            dx.baddbmm_(dz_cf.transpose(1, 2), w1.unsqueeze(0).expand(Bsz, -1, -1))
            sum_lead(torch.bmm(dx_cf, x), out=dw[:D])
            sum_lead(torch.bmm(dz_cf, x), out=dw[D:])
        db = None
        if ctx.has_bias:
            db = torch.cat([dx_cf.sum(dim=(0, 2)), dz_cf.sum(dim=(0, 2))])
        return dx, dw, db


def in_proj_cf(x_rows, weight, bias):
    _need_hip(x_rows, weight)
    return InProjFn.apply(x_rows, weight, bias)


class OutProjFn(torch.autograd.Function):
    """SS2D.out_proj without bias (MedMamba.py:302) on channel-first planes, one GEMM launch per product: over B*L columns for
    channel-major planes, batched over B otherwise (the weight gradient then takes a sum over B)."""

    @staticmethod
    def forward(ctx, y_cf, weight):
        B, D, L = y_cf.shape
        cm = B > 1 and _is_cm(y_cf)
        ctx.save_for_backward(y_cf, weight)
        ctx.cm = cm
        if cm:
            return _gemm(weight, _cm2d(y_cf)).view(-1, B, L).permute(1, 0, 2)
        y_cf = _rows(y_cf)
        return _gemm(weight, y_cf)

    @staticmethod
    def backward(ctx, dout):
        y_cf, weight = ctx.saved_tensors
        B, D, L = y_cf.shape
        dy = dw = None
        if ctx.cm and B > 1 and _is_cm(dout):
            g2 = _cm2d(dout)                                                                   # (d_model, B*L)
            if ctx.needs_input_grad[0]:
                dy = _gemm(weight.t(), g2).view(D, B, L).permute(1, 0, 2)
            if ctx.needs_input_grad[1]:
                dw = _gemm(g2, _cm2d(y_cf).t())
        else:
            g = _rows(dout)
            if ctx.needs_input_grad[0]:
                dy = _gemm(weight.t(), g)
            if ctx.needs_input_grad[1]:
                dw = sum_lead(_gemm(g, _rows(y_cf).transpose(1, 2)))
        return dy, dw


def out_proj_cf(y_cf, weight, bias=None):
    """SS2D.out_proj (MedMamba.py:302) on channel-first planes: (B, D, L) -> (B, d_model, L) in the storage layout of y_cf."""
    out = OutProjFn.apply(y_cf, weight)
    return out if bias is None else out + bias[:, None]


class DwConvSiluCrossFn(torch.autograd.Function):
    """Depthwise conv3x3 + bias + SiLU on channel-first planes, written in the scan's two image orders
    (MedMamba.py:153-162, 295 + the stack/transpose of :256).  x_cf: (B, D, L) planes (any batch / channel stride),
    weight (D,1,3,3), bias (D) or None  ->  u2 (B, 2*D, L) in the storage layout channel_major(B, L) picks."""

    @staticmethod
    def forward(ctx, x_cf, weight, bias, H, W):
        B, D, L = x_cf.shape
        assert L == H * W
        x_cf = _rows(x_cf)
        weight = weight.float().contiguous()
        bias = None if bias is None else bias.float().contiguous()
        cm = channel_major(B, L)
        u2 = _planes(B, 2 * D, L, x_cf.device, cm)
        with _lib.device_guard(x_cf.device):
            rc = _lib.lib().mm_dwconv_silu_cross_fwd(*_pl(x_cf), weight.data_ptr(), None if bias is None else bias.data_ptr(),
                                                     *_pl(u2), B, D, H, W, _stream())
        _lib.check(rc, "mm_dwconv_silu_cross_fwd")
        ctx.save_for_backward(x_cf, weight, bias)
        ctx.hw = (H, W)
        ctx.cm = cm
        return u2

    @staticmethod
    def backward(ctx, du2):
        x_cf, weight, bias = ctx.saved_tensors
        H, W = ctx.hw
        B, D, L = x_cf.shape
        du2 = _rows(du2)
        dx = _planes(B, D, L, du2.device, ctx.cm)
        ws = torch.empty((B, D * _lib.lib().mm_dwconv_silu_cross_strips(H, W), 10), device=du2.device, dtype=torch.float32)
        with _lib.device_guard(du2.device):
            rc = _lib.lib().mm_dwconv_silu_cross_bwd(*_pl(du2), None, 0, 0, *_pl(x_cf), weight.data_ptr(),
                                                     None if bias is None else bias.data_ptr(), *_pl(dx), ws.data_ptr(),
                                                     B, D, H, W, _stream())
        _lib.check(rc, "mm_dwconv_silu_cross_bwd")
        s = ws.view(B, D, -1, 10).sum((0, 2))
        return dx, s[:, :9].reshape(D, 1, 3, 3), (None if bias is None else s[:, 9]), None, None


def dwconv_silu_cross(x_cf, weight, bias, H, W):
    _need_hip(x_cf, weight)
    return DwConvSiluCrossFn.apply(x_cf, weight, bias, H, W)


class SS2DCoreFn(torch.autograd.Function):
    """x/dt projections + 4-direction selective scan + cross-merge + out_norm LayerNorm + SiLU(z) gate in channel-first
    planes (MedMamba.py:259-262, 271-286, 298-301).

    u2 (B, 2D, L): row-major | column-major image (dwconv_silu_cross);  the five SS2D parameters exactly as the module
    holds them (reference direction order): x_proj_weight (4, R+2N, D), dt_projs_weight (4, D, R), dt_projs_bias (4, D),
    A_logs (4D, N), Ds (4D);  z_cf (B, D, L);  LayerNorm weight / bias / eps; image size.
    Returns y_cf (B, D, L) = LN_D(merge(scan(...))) * silu(z).

    One kernel (mm_ss2d_pack_fwd) brings the parameters into kernel direction order with A = -exp(A_logs); its mirror
    returns all five parameter gradients from one packed buffer that the GEMMs and the backward scan write into — no
    per-parameter permute / exp / zero-fill launches.  The projections live inside the Function so that autograd
    never slices x_dbl: the backward kernel writes dB / dC straight into the row blocks of d(x_dbl) (strided outputs of
    mm_scan_bwd), the dt rows are filled by one GEMM, and d(u2) collects its three contributions (two direction pairs
    + the x projection) with one add and one GEMM (beta = 1).  Everything the Function allocates follows
    channel_major(B, L): in channel-major storage x_dbl is (4, C, B*L), delta (4D, B*L), and each projection (and each
    weight gradient) is ONE GEMM; batch-major keeps the batched GEMMs + sums over B.  Saved: u2, x_dbl, delta, the packed
    parameters, the state checkpoints, the merged pre-norm tensor m and the LN statistics; the (B, 4D, L) scan output is
    freed after the merge."""

    @staticmethod
    def _segments(P, D, C, R, N):
        al = lambda n: (n + 63) & ~63                 # segments start on 256-B boundaries (mm_ss2d_pack_fwd)
        o1 = al(4 * C * D)
        o2 = al(o1 + 4 * D * R)
        o3 = al(o2 + 4 * D * N)
        o4 = al(o3 + 4 * D)
        return (P[:4 * C * D].view(4, C, D), P[o1:o1 + 4 * D * R].view(4, D, R), P[o2:o2 + 4 * D * N].view(4 * D, N),
                P[o3:o3 + 4 * D], P[o4:o4 + 4 * D])

    @staticmethod
    def forward(ctx, u2, conv_w, conv_b, x_proj_w, dt_w, dt_b, A_logs, Ds, z_cf, ln_w, ln_b, H, W, eps, prescan_event=None,
                need_grad=True):
        from .selective_scan_interface import _CROSS_SHARED, _launch_fwd, dt_fusable
        lib = _lib.lib()
        dev = u2.device
        L = u2.shape[2]
        Bsz = u2.shape[0]
        cm = channel_major(Bsz, L)
        x_cf = None
        if conv_w is not None:
            # depthwise conv3x3 + SiLU (MedMamba.py:294-295) inside the Function: u2 is (B, D, L) = x here, and the backward
            # hands the three contributions to d(u2) — two direction pairs of the scan + the x projection — to ONE kernel
            x_cf = _rows(u2)
            conv_w = conv_w.float().contiguous()
            conv_b = None if conv_b is None else conv_b.float().contiguous()
            Dc = x_cf.shape[1]
            u2 = _planes(Bsz, 2 * Dc, L, dev, cm)
            with _lib.device_guard(dev):
                rc = lib.mm_dwconv_silu_cross_fwd(*_pl(x_cf), conv_w.data_ptr(), None if conv_b is None else conv_b.data_ptr(),
                                                  *_pl(u2), Bsz, Dc, H, W, _stream())
            _lib.check(rc, "mm_dwconv_silu_cross_fwd")
        D2 = u2.shape[1]
        D, R, N = D2 // 2, dt_w.shape[2], A_logs.shape[1]
        C, Q = R + 2 * N, Bsz * L
        srcs = [t.float().contiguous() for t in (x_proj_w, dt_w, dt_b, A_logs, Ds)]
        P = torch.empty((lib.mm_ss2d_pack_size(D, C, R, N),), device=dev, dtype=torch.float32)
        with _lib.device_guard(dev):
            rc = lib.mm_ss2d_pack_fwd(*[t.data_ptr() for t in srcs], P.data_ptr(), D, C, R, N, _stream())
        _lib.check(rc, "mm_ss2d_pack_fwd")
        Wx, Wdt, A, Dp, dbias = SS2DCoreFn._segments(P, D, C, R, N)
        need_grad = bool(need_grad) and any(ctx.needs_input_grad)
        if cm:
            u2m = _cm2d(_rows(u2))                                                             # (2D, Q)
            u2 = u2m.view(2 * D, Bsz, L).permute(1, 0, 2)
            x_dbl = _gemm(Wx.view(2, 2 * C, D), u2m.view(2, D, Q)).view(4, C, Q)                  # :259
            xb = x_dbl.view(4, C, Bsz, L).permute(2, 0, 1, 3)                                   # (B, 4, C, L) view
        else:
            u2 = u2.float().contiguous()
            x_dbl = torch.matmul(Wx.view(1, 2, 2 * C, D), u2.view(Bsz, 2, D, L)).view(Bsz, 4, C, L)
            xb = x_dbl
        # inference (nothing to differentiate) with a small dt rank: the dt projection (:262) runs inside the scan's staging
        # phase — no (B, 4D, L) delta tensor is written or read, one GEMM less; training keeps the GEMM (the backward kernel
        # consumes delta as a tensor)
        fuse_dt = _FUSE_DT and not need_grad and dt_fusable(R, L, xb[:, :, :R])
        if fuse_dt:
            delta = None
        elif cm:
            delta = _gemm(Wdt, x_dbl[:, :R]).view(4 * D, Bsz, L).permute(1, 0, 2)                # :262  (B, 4D, L) view
        else:
            delta = torch.matmul(Wdt.unsqueeze(0), x_dbl[:, :, :R]).view(Bsz, 4 * D, L)
        if prescan_event is not None:
            prescan_event.record()          # the projections are queued; what follows is the (latency-bound) scan
        out4, x_chk = _launch_fwd(u2, delta, A, xb[:, :, R:R + N], xb[:, :, R + N:], Dp, dbias, True, need_grad, 0,
                                  _CROSS_SHARED, dt=(xb[:, :, :R], Wdt.reshape(4 * D, R)) if fuse_dt else None)
        z_cf = _rows(z_cf)
        ln_w, ln_b = ln_w.float().contiguous(), ln_b.float().contiguous()
        m = _planes(Bsz, D, L, dev, cm)
        y = _planes(Bsz, D, L, dev, cm)
        mu = torch.empty((Bsz, L), device=dev, dtype=torch.float32)
        rstd = torch.empty((Bsz, L), device=dev, dtype=torch.float32)
        with _lib.device_guard(dev):
            rc = lib.mm_cross_merge_fwd(out4.data_ptr(), *_pl(m), Bsz, D, H, W, _stream())
            _lib.check(rc, "mm_cross_merge_fwd")
            rc = lib.mm_ln_gate_fwd(*_pl(m), *_pl(z_cf), ln_w.data_ptr(), ln_b.data_ptr(), float(eps), *_pl(y),
                                    mu.data_ptr(), rstd.data_ptr(), Bsz, D, L, _stream())
            _lib.check(rc, "mm_ln_gate_fwd")
        if need_grad:
            ctx.save_for_backward(u2, x_dbl, delta, P, x_chk, m, mu, rstd, z_cf, ln_w, ln_b, x_cf, conv_w, conv_b)
            ctx.dims = (H, W, D, C, R, N, cm)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .selective_scan_interface import _CROSS_SHARED, _launch_bwd
        u2, x_dbl, delta, P, x_chk, m, mu, rstd, z_cf, ln_w, ln_b, x_cf, conv_w, conv_b = ctx.saved_tensors
        fused_conv = conv_w is not None
        H, W, D, C, R, N, cm = ctx.dims
        Wx, Wdt, A, Dp, dbias = SS2DCoreFn._segments(P, D, C, R, N)
        Bsz, _, L = m.shape
        Q = Bsz * L
        dev = m.device
        dy = _rows(dy)
        dout2 = _planes(Bsz, 2 * D, L, dev, cm)         # channel block 0: dm, block 1: its plane transpose
        # d(x_cf) | d(z_cf) as the two channel halves of ONE buffer, the layout in_proj produced them in: its backward then
        # runs one GEMM per product over all 2D rows (InProjFn.backward)
        dxz = _planes(Bsz, 2 * D, L, dev, cm)
        dz = dxz[:, D:]
        lib = _lib.lib()
        ws = torch.empty((lib.mm_ln_gate_rows(Bsz, D, L), 2 * D), device=dev, dtype=torch.float32)
        with _lib.device_guard(dev):
            rc = lib.mm_ln_gate_bwd(*_pl(dy), *_pl(m), *_pl(z_cf), ln_w.data_ptr(), ln_b.data_ptr(), mu.data_ptr(),
                                    rstd.data_ptr(), *_pl(dout2), *_pl(dz), ws.data_ptr(), Bsz, D, L, _stream())
            _lib.check(rc, "mm_ln_gate_bwd")
            d1 = dout2[:, D:]
            rc = lib.mm_plane_transpose(*_pl(dout2), *_pl(d1), Bsz, D, H, W, _stream())
            _lib.check(rc, "mm_plane_transpose")
        # (the rows of ws are summed by mm_ss2d_pack_bwd below, together with the depthwise conv's partial sums)
        # gradient of the packed parameters: the GEMMs below write the two weight segments of dP; the A / D / bias segments come
        # from per-batch-item partial buffers that the scan kernel fills with plain stores and mm_ss2d_pack_bwd sums in a
        # fixed order (no atomics, no zero-fill: bitwise reproducible)
        dP = torch.empty_like(P)
        dWx, dWdt, dA, dD, ddb = SS2DCoreFn._segments(dP, D, C, R, N)
        S = lib.mm_ss2d_pack_parts_size(D, C, R, N)
        oA = dA.storage_offset()
        parts = torch.empty((Bsz, S), device=dev, dtype=torch.float32)
        # d(x_dbl): the dB / dC rows are fully written by the kernel (in place, or summed from per-workgroup planes), the dt rows
        # by the GEMM below
        if cm:
            dx_dbl = torch.empty((4, C, Q), device=dev, dtype=torch.float32)
            xb, dxb = x_dbl.view(4, C, Bsz, L).permute(2, 0, 1, 3), dx_dbl.view(4, C, Bsz, L).permute(2, 0, 1, 3)
        else:
            dx_dbl = torch.empty((Bsz, 4, C, L), device=dev, dtype=torch.float32)
            xb, dxb = x_dbl, dx_dbl
        du4, ddelta = _launch_bwd(u2, delta, A, xb[:, :, R:R + N], xb[:, :, R + N:], Dp, dbias, x_chk, dout2, True,
                                  _CROSS_SHARED, dBC_dst=dxb[:, :, R:],
                                  parts=(parts, 0, dD.storage_offset() - oA, ddb.storage_offset() - oA), channel_major=cm)[:2]
        if cm:
            dd = ddelta.permute(1, 0, 2).reshape(4, D, Q)                                      # views of (4D, B, L) storage
            _gemm(dd, x_dbl[:, :R].transpose(1, 2), out=dWdt)                                   # (4, D, R)
            _gemm(Wdt.transpose(1, 2), dd, out=dx_dbl[:, :R])                                   # dt rows of d(x_dbl), in place
            dx2 = dx_dbl.view(2, 2 * C, Q)
            if fused_conv:
                du2m = _gemm(Wx.view(2, 2 * C, D).transpose(1, 2), dx2)                          # Wx^T d(x_dbl); pairs added later
            else:
                d4 = du4.permute(1, 0, 2).reshape(2, 2, D, Q)
                du2m = d4[:, 0] + d4[:, 1]                                                     # the two directions of a pair
                du2m.baddbmm_(Wx.view(2, 2 * C, D).transpose(1, 2), dx2)                         # + Wx^T d(x_dbl)
            _gemm(dx2, _cm2d(u2).view(2, D, Q).transpose(1, 2), out=dWx.view(2, 2 * C, D))
            du2 = du2m.view(2 * D, Bsz, L).permute(1, 0, 2)
        else:
            dd = ddelta.view(Bsz, 4, D, L)
            xr = x_dbl[:, :, :R]
            sum_lead(torch.matmul(dd, xr.transpose(-1, -2)), out=dWdt)                           # (4, D, R)
            dx_dbl[:, :, :R] = torch.matmul(Wdt.transpose(-1, -2).unsqueeze(0), dd)               # dt rows of d(x_dbl)
            Wx2 = Wx.view(2, 2 * C, D)
            dxd2 = dx_dbl.view(Bsz, 2, 2 * C, L)
            WxT = Wx2.transpose(1, 2).unsqueeze(0).expand(Bsz, -1, -1, -1).reshape(Bsz * 2, D, 2 * C)
            if fused_conv:
                du2 = torch.bmm(WxT, dxd2.reshape(Bsz * 2, 2 * C, L))                            # Wx^T d(x_dbl); pairs added later
            else:
                d4 = du4.view(Bsz, 2, 2, D, L)
                du2 = (d4[:, :, 0] + d4[:, :, 1]).view(Bsz * 2, D, L)                             # the two directions of a pair
                du2.baddbmm_(WxT, dxd2.reshape(Bsz * 2, 2 * C, L))                               # + Wx^T d(x_dbl)
            sum_lead(torch.matmul(dxd2, u2.view(Bsz, 2, D, L).transpose(-1, -2)), out=dWx.view(2, 2 * C, D))
            du2 = du2.view(Bsz, 2 * D, L)
        dcw = dcb = wsc = None
        strips = 0
        if fused_conv:
            # d(u2) = projection part (du2) + the scan's two direction pairs (du4), summed inside the conv's backward kernel
            dxc = dxz[:, :D]
            strips = lib.mm_dwconv_silu_cross_strips(H, W)
            wsc = torch.empty((Bsz, D * strips, 10), device=dev, dtype=torch.float32)
            with _lib.device_guard(dev):
                rc = lib.mm_dwconv_silu_cross_bwd(*_pl(du2), *_pl(du4), *_pl(x_cf), conv_w.data_ptr(),
                                                  None if conv_b is None else conv_b.data_ptr(), *_pl(dxc), wsc.data_ptr(),
                                                  Bsz, D, H, W, _stream())
            _lib.check(rc, "mm_dwconv_silu_cross_bwd")
            du2 = dxc
        # one launch: packed gradients back to the module's parameter layouts (A / D / bias summed over the batch partials), the
        # rows of ln_gate's ws -> d(out_norm weight | bias), the depthwise conv's partial sums -> d(conv weight | bias)
        npk = P.numel()
        G = torch.empty((npk + 2 * D + 10 * D,), device=dev, dtype=torch.float32)
        ln_out, dw_out = G[npk:npk + 2 * D], G[npk + 2 * D:]
        fold = _PACK_FOLD
        with _lib.device_guard(dev):
            rc = lib.mm_ss2d_pack_bwd(dP.data_ptr(), P.data_ptr(), parts.data_ptr(), G.data_ptr(), D, C, R, N, Bsz,
                                      ws.data_ptr() if fold else None, ws.shape[0], ln_out.data_ptr(),
                                      None if (wsc is None or not fold) else wsc.data_ptr(), Bsz, strips,
                                      None if wsc is None else dw_out.data_ptr(), _stream())
        _lib.check(rc, "mm_ss2d_pack_bwd")
        gWx, gWdt, gA, gD, gb = SS2DCoreFn._segments(G[:npk], D, C, R, N)
        if not fold:                                  # A/B switch (MM_PACK_FOLD=0): the reductions as separate ATen launches
            ln_out = sum_lead(ws)
            if wsc is not None:
                sc = wsc.view(Bsz, D, -1, 10).sum((0, 2))
                dw_out = torch.cat([sc[:, :9].reshape(-1), sc[:, 9]])
        if fused_conv:
            dcw, dcb = dw_out[:9 * D].view(D, 1, 3, 3), (None if conv_b is None else dw_out[9 * D:])
        return (du2, dcw, dcb, gWx, gWdt, gb.view(4, D), gA, gD, dz, ln_out[:D], ln_out[D:], None, None, None, None, None)


# ---- off by default: the SS2D backward's parameter half on a third stream -----------------------------------------
# Nothing on the way to d(input) waits for the four weight-gradient GEMMs and the un-packing launch of a block.  MM_PARAM_STREAM=1
# issues them (ss2d_bwd_params) on a third stream behind an event recorded after the data half (ss2d_bwd_data), for the channel-major
# blocks (the 14x14 and 7x7 stages); the stream of the backward pass waits for that stream ONCE, in an end-of-backward callback of
# the autograd engine.  Measured on one box, alternating: 27.64 instead of 28.18 ms per step of S / 64 (-1.9 %).
# The GEMMs on that stream must be rocBLAS kernels: with hipBLASLt kernels there (the overall winners of three of the four shapes)
# the GPU STOPPED in 9 of 20 runs of bench.py — whatever GPU_MAX_HW_QUEUES, whatever route the GEMM took — while the same steps
# around element-wise work or square GEMMs, or with rocBLAS kernels only (tuning/gemm_gfx950_rocblas.csv, set_rocblas_only), never
# did: 0 of 107 runs.  Off by default until it has been soaked through the test suite and with RCCL beside it.
_PARAM_STREAM_MODE = os.environ.get("MM_PARAM_STREAM", "0")
_PARAM_STREAMS = {}            # device -> [stream, join scheduled?, used since the last join?]
_PARAM_COVERED = {}            # (batch, L, d_model, d_inner, rows of x_dbl) -> the block's four weight-gradient GEMMs have rocBLAS records


def _param_state(device):
    st = _PARAM_STREAMS.get(device)
    if st is None:
        st = _PARAM_STREAMS[device] = [torch.cuda.Stream(device=device), False, False]
    return st


def join_param_stream(device=None):
    """Make the current stream wait for the parameter-gradient work queued so far (no-op when there is none)."""
    for dev, st in _PARAM_STREAMS.items():
        if (device is None or dev == torch.device(device)) and st[2]:
            torch.cuda.current_stream(dev).wait_stream(st[0])
            st[2] = False


def _join_at_end_of_backward(device):
    st = _param_state(device)

    def cb():
        st[1] = False
        join_param_stream(device)
    if not st[1]:
        st[1] = True
        torch.autograd.Variable._execution_engine.queue_callback(cb)


class SS2DBranchFn(torch.autograd.Function):
    """The whole SS2D branch (MedMamba.py:288-305 without dropout) as one autograd node whose forward and backward are ONE call
    each into the C++ sequencing layer (csrc_host/ss2d_host.cpp): x (B, L, d_model) rows -> (B, d_model, L) planes.  Same
    kernels, GEMMs, allocations and layouts as InProjFn + SS2DCoreFn (fused depthwise conv) + OutProjFn, issued without the
    interpreter between the launches (0.2 + 0.4 ms of host time per block and step)."""

    @staticmethod
    def forward(ctx, x, in_w, conv_w, conv_b, x_proj_w, dt_w, dt_b, A_logs, Ds, ln_w, ln_b, out_w, H, W, eps, prescan_event,
                need_grad=True):
        from . import _host
        from .selective_scan_interface import KERNEL_TIMER, _FWD_VARIANT, scan_bytes_fwd
        Bsz, L, _ = x.shape
        D, N = in_w.shape[0] // 2, A_logs.shape[1]
        cm = channel_major(Bsz, L)
        need_grad = bool(need_grad) and any(ctx.needs_input_grad)
        x = x.contiguous()
        with _lib.device_guard(x.device):
            bt = BRANCH_TIMER.start()
            ev0, ev1 = KERNEL_TIMER.pair("scan_fwd", scan_bytes_fwd(Bsz, 4 * D, L, N, 4), Bsz * 4 * D * L * N)
            res = _host.module().ss2d_fwd(x, in_w, conv_w, conv_b, x_proj_w, dt_w, dt_b, A_logs, Ds, ln_w, ln_b, out_w, H, W, float(eps),
                                          cm, need_grad, _FWD_VARIANT, _stream(), ev0, ev1, prescan_event, _FUSE_DT)
            BRANCH_TIMER.stop("ss2d_fwd", bt)
        if need_grad:
            ctx.save_for_backward(x, in_w, conv_w, conv_b, ln_w, ln_b, out_w, *res[1:])
            ctx.dims = (H, W, cm)
        return res[0]

    @staticmethod
    def backward(ctx, dout):
        from . import _host
        from .selective_scan_interface import KERNEL_TIMER, _BWD_VARIANT, scan_bytes_bwd
        x, in_w, conv_w, conv_b, ln_w, ln_b, out_w, xz, u2, x_dbl, delta, P, x_chk, m, mu, rstd, y = ctx.saved_tensors
        H, W, cm = ctx.dims
        Bsz, L, _ = x.shape
        D = in_w.shape[0] // 2
        with _lib.device_guard(x.device):
            bt = BRANCH_TIMER.start()
            ev0, ev1 = KERNEL_TIMER.pair("scan_bwd", scan_bytes_bwd(Bsz, 4 * D, L, 16, 4), Bsz * 4 * D * L * 16)
            ps_ok = _PARAM_STREAM_MODE != "0" and cm     # (batch-major blocks: their weight gradients go through ATen, hipBLASLt included)
            if ps_ok:           # ... and only with an explicit rocBLAS solution on record for each of the four GEMMs
                key = (Bsz, L, x.shape[2], D, x_dbl.shape[1])
                ps_ok = _PARAM_COVERED.get(key)
                if ps_ok is None:
                    ps_ok = _PARAM_COVERED[key] = bool(_host.module().ss2d_params_covered(Bsz, L, x.shape[2], D, x_dbl.shape[1], x_dbl.shape[1] - 32))
            if not ps_ok:
                g = _host.module().ss2d_bwd(dout, x, in_w, conv_w, conv_b, ln_w, ln_b, out_w, xz, u2, x_dbl, delta, P, x_chk, m, mu, rstd, y,
                                            H, W, cm, _PACK_FOLD, _BWD_VARIANT, _stream(), ev0, ev1)
            else:
                mod = _host.module()
                d = mod.ss2d_bwd_data(dout, x, in_w, conv_w, conv_b, ln_w, ln_b, out_w, xz, u2, x_dbl, delta, P, x_chk, m, mu, rstd,
                                      H, W, cm, _BWD_VARIANT, _stream(), ev0, ev1)
                st = _param_state(x.device)
                ps = st[0]
                main = torch.cuda.current_stream(x.device)
                ev = torch.cuda.Event()     # a new event per call (re-recording one with a wait still queued is asking for trouble)
                ev.record(main)
                with torch.cuda.stream(ps):
                    ps.wait_event(ev)
                    mod.set_rocblas_only(True)      # no hipBLASLt kernel on the third stream (see above)
                    try:
                        gp = mod.ss2d_bwd_params(dout, x, in_w, conv_b, out_w, u2, x_dbl, P, y, *d[1:], H, W, cm, _PACK_FOLD, _stream())
                    finally:
                        mod.set_rocblas_only(False)
                for t in (dout, x, u2, x_dbl, P, y, *d[1:]):   # freed by the engine / this frame while `ps` may still read them
                    t.record_stream(ps)
                for t in gp:                # allocated under `ps`, consumed on the main stream (optimizer, all-reduce)
                    if t is not None:
                        t.record_stream(main)
                st[2] = True
                _join_at_end_of_backward(x.device)
                g = [d[0], *gp]
            BRANCH_TIMER.stop("ss2d_bwd", bt)
        dx, d_in, dcw, dcb, gWx, gWdt, gb, gA, gD, dlw, dlb, d_out = g
        return dx, d_in, dcw, dcb, gWx, gWdt, gb, gA, gD, dlw, dlb, d_out, None, None, None, None, None


def ss2d_branch_native_ok(x, mod_in_proj, mod_out_proj, conv, params):
    """True when SS2DBranchFn can take the whole branch: the extension is built, fp32 contiguous parameters, no biases on the two
    projections (MedMamba.py:139, 181: bias=False by default).  Without gradients the same call runs the inference form (no
    checkpoints, the dt projection inside the scan kernel)."""
    from . import _host
    if _host.module() is None or not x.is_cuda or x.dtype != torch.float32:
        return False
    if mod_in_proj.bias is not None or mod_out_proj.bias is not None:
        return False
    ts = [mod_in_proj.weight, mod_out_proj.weight, conv.weight, *params] + ([] if conv.bias is None else [conv.bias])
    return all(t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda for t in ts)


def ss2d_branch(x_rows, in_w, conv_w, conv_b, x_proj_w, dt_w, dt_b, A_logs, Ds, ln_w, ln_b, out_w, H, W, eps=1e-5, prescan_event=None):
    D = in_w.shape[0] // 2
    if (A_logs.shape != (4 * D, 16) or x_proj_w.shape[0] != 4 or x_proj_w.shape[2] != D or dt_w.shape[:2] != (4, D)
            or dt_b.shape != (4, D) or Ds.shape != (4 * D,) or x_proj_w.shape[1] != dt_w.shape[2] + 32
            or tuple(conv_w.shape) != (D, 1, 3, 3) or out_w.shape[1] != D):
        raise NotImplementedError("ss2d_branch: expects 4 directions, d_state 16, a depthwise 3x3 conv over d_inner channels")
    return SS2DBranchFn.apply(x_rows, in_w, conv_w, conv_b, x_proj_w, dt_w, dt_b, A_logs, Ds, ln_w, ln_b, out_w, H, W, eps,
                              prescan_event, _wants_grad(x_rows, in_w, conv_w, conv_b, x_proj_w, dt_w, dt_b, A_logs, Ds, ln_w, ln_b, out_w))


def ss2d_core(u2, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, z_cf, ln_w, ln_b, H, W, eps=1e-5,
              prescan_event=None):
    """Parameters in the module's own (reference) layout and direction order; see SS2DCoreFn.
    prescan_event: optional torch.cuda.Event recorded on the current stream right before the scan kernel is queued (lets a
    caller start independent work on another stream exactly when the latency-bound scan begins)."""
    _need_hip(u2, z_cf)
    D = u2.shape[1] // 2
    if (A_logs.shape != (4 * D, 16) or x_proj_weight.shape[0] != 4 or x_proj_weight.shape[2] != D
            or dt_projs_weight.shape[:2] != (4, D) or dt_projs_bias.shape != (4, D) or Ds.shape != (4 * D,)
            or x_proj_weight.shape[1] != dt_projs_weight.shape[2] + 32):
        raise NotImplementedError("ss2d_core: expects 4 directions, d_state 16, u2 with 2*D channels")
    return SS2DCoreFn.apply(u2, None, None, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, z_cf, ln_w, ln_b, H, W,
                            eps, prescan_event, _wants_grad(u2, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, z_cf, ln_w, ln_b))


def ss2d_conv_core(x_cf, conv_weight, conv_bias, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, z_cf, ln_w, ln_b,
                   H, W, eps=1e-5, prescan_event=None):
    """dwconv_silu_cross + ss2d_core in ONE autograd Function (MedMamba.py:294-301 on channel-first planes): x_cf (B, D, L) ->
    y_cf (B, D, L).  Same values; the backward adds the scan's per-direction input gradients and the projection's inside the
    depthwise conv's backward kernel instead of a pair-sum kernel plus an accumulating GEMM."""
    _need_hip(x_cf, z_cf)
    D = x_cf.shape[1]
    if (A_logs.shape != (4 * D, 16) or x_proj_weight.shape[0] != 4 or x_proj_weight.shape[2] != D
            or dt_projs_weight.shape[:2] != (4, D) or dt_projs_bias.shape != (4, D) or Ds.shape != (4 * D,)
            or x_proj_weight.shape[1] != dt_projs_weight.shape[2] + 32 or tuple(conv_weight.shape) != (D, 1, 3, 3)):
        raise NotImplementedError("ss2d_conv_core: expects 4 directions, d_state 16, a (D,1,3,3) depthwise kernel")
    return SS2DCoreFn.apply(x_cf, conv_weight, conv_bias, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, z_cf,
                            ln_w, ln_b, H, W, eps, prescan_event,
                            _wants_grad(x_cf, conv_weight, conv_bias, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, z_cf, ln_w, ln_b))


def channel_sum_nchw(x):
    """(B, C, H, W) or (B, C, L) contiguous fp32 -> (C,) sums over batch and positions (a conv's bias gradient)."""
    _need_hip(x)
    x = x.float().contiguous()
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    lib = _lib.lib()
    split = lib.mm_channel_sum_nchw_split(B, C)
    out = torch.empty((split, C), device=x.device, dtype=torch.float32)
    with _lib.device_guard(x.device):
        rc = lib.mm_channel_sum_nchw(x.data_ptr(), out.data_ptr(), B, C, HW, _stream())
    _lib.check(rc, "mm_channel_sum_nchw")
    return out[0] if split == 1 else sum_lead(out)     # rows of per-batch-part sums, added in a fixed order (no atomics)


def _bias_grad(dy):
    """Sum over batch and positions of an NC(HW) gradient: our kernel for large planes (tools/bench_channel_sum.py, B = 64:
    10.7 vs 58 us at 48x56x56, 10.5 vs 16.6 us at 96x28x28), ATen's reduction for small ones (7-8 us vs our 11-12 us)."""
    hw = dy.numel() // (dy.shape[0] * dy.shape[1])
    return channel_sum_nchw(dy) if hw >= 512 else dy.sum(dim=tuple(d for d in range(dy.dim()) if d != 1))


class ConvBiasFn(torch.autograd.Function):
    """F.conv2d(x, w, b) (groups = 1) through MIOpen exactly as autograd would run it, except for the bias gradient: ATen's
    generic reduction takes 59 us for a 64x48x56x56 gradient (0.65 TB/s), mm_channel_sum_nchw a fraction of that."""

    @staticmethod
    def forward(ctx, x, w, b, stride, padding, dilation, add_bias=True):
        # add_bias=False: the caller folds the bias into the BatchNorm that follows (ops.bn_relu_train(pre_bias=...)): y is the
        # convolution WITHOUT bias (MIOpen adds a bias in a separate elementwise pass: 47 us at 64 x 48 x 56 x 56); the bias gets
        # its gradient (the channel sums of this conv's dy = the BatchNorm's dx) from BNReluFn.backward
        y = torch.nn.functional.conv2d(x, w, b if add_bias else None, stride, padding, dilation)
        ctx.save_for_backward(x, w)
        ctx.cfg = (list(stride), list(padding), list(dilation), b is not None and add_bias)     # deferred bias: its gradient comes
        return y                                                                                 # from BNReluFn (pre_bias)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, dilation, has_bias = ctx.cfg
        dy = dy.contiguous()
        if (torch.backends.cudnn.deterministic and ctx.needs_input_grad[1] and x.is_cuda and tuple(w.shape[2:]) == (3, 3)
                and stride == [1, 1] and padding == [1, 1] and dilation == [1, 1]):
            # deterministic mode (the reference's set_seed, train.py:28-29): MIOpen's reproducible weight gradient is a per-image
            # im2col + GEMM (640 launches, +17.5 ms per MedMamba-S step); one im2col + one batched GEMM + an ordered sum over the
            # batch is the same arithmetic, reproducible, at a third of the cost (csrc_host conv3x3_bwd is the same code)
            dx = torch.ops.aten.convolution_backward(dy, x, w, None, stride, padding, dilation, False, [0, 0], 1,
                                                     [True, False, False])[0] if ctx.needs_input_grad[0] else None
            B, K, C, HW = dy.shape[0], dy.shape[1], x.shape[1], x.shape[2] * x.shape[3]
            tiles, gs = ((K + 63) // 64) * ((9 * C + 63) // 64), 1                     # images per GEMM (csrc_host conv3x3_bwd)
            while gs * 2 <= B and B % (gs * 2) == 0 and (B // (gs * 2)) * tiles >= 512:
                gs *= 2
            cols = torch.empty((B // gs, 9 * C, gs * HW), device=x.device, dtype=torch.float32)   # gs = 1: F.unfold(x, 3, padding=1),
            with _lib.device_guard(x.device):                                          # all images in one launch
                _lib.check(_lib.lib().mm_im2col3x3(x.data_ptr(), cols.data_ptr(), B, C, x.shape[2], x.shape[3], gs, _stream()),
                           "mm_im2col3x3")
            dyg = dy.reshape(B, K, HW) if gs == 1 else dy.reshape(B // gs, gs, K, HW).transpose(1, 2).reshape(B // gs, K, gs * HW)
            dw = sum_lead(torch.bmm(dyg, cols.transpose(1, 2))).view(w.shape)
        elif (torch.backends.cudnn.deterministic and ctx.needs_input_grad[1] and not ctx.needs_input_grad[0] and x.is_cuda
              and list(w.shape[2:]) == stride and padding == [0, 0] and dilation == [1, 1]
              and x.shape[2] % stride[0] == 0 and x.shape[3] % stride[1] == 0):
            # deterministic mode, a patch convolution (kernel = stride: PatchEmbed2D's 4x4 / 4, MedMamba.py:62) whose input needs no
            # gradient: MIOpen's reproducible weight gradient is a per-image im2col + GEMM (128 launches, 2 ms per MedMamba-S step at
            # 64 images); the patches do not overlap, so im2col is one permuted copy and the gradient one batched GEMM + an ordered sum
            B, C, H, W = x.shape
            kh, kw = stride
            cols = x.view(B, C, H // kh, kh, W // kw, kw).permute(0, 1, 3, 5, 2, 4).reshape(B, C * kh * kw, (H // kh) * (W // kw))
            dw = sum_lead(torch.bmm(dy.reshape(B, dy.shape[1], -1), cols.transpose(1, 2))).view(w.shape)
            dx = None
        else:
            dx, dw, _ = torch.ops.aten.convolution_backward(dy, x, w, None, stride, padding, dilation, False, [0, 0], 1,
                                                            [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False])
        db = _bias_grad(dy) if (has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None, None, None, None


def conv2d_bias_ok(x, conv):
    return (x.is_cuda and conv.bias is not None and conv.groups == 1 and conv.padding_mode == "zeros"
            and not isinstance(conv.padding, str) and x.dtype == torch.float32 and x.is_contiguous())


def conv2d_bias(x, conv, add_bias=True):
    """nn.Conv2d `conv` applied to x with ConvBiasFn when it is a plain dense conv with a bias on a HIP tensor.
    add_bias=False (only when conv2d_bias_ok): the output lacks the bias, see ConvBiasFn."""
    if conv2d_bias_ok(x, conv):
        return ConvBiasFn.apply(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, add_bias)
    assert add_bias
    return conv(x)


class PointwiseConvFn(torch.autograd.Function):
    """nn.Conv2d(kernel_size=1) on NCHW input as out[b] = W @ x[b] + bias: batched GEMMs with a broadcast weight (MIOpen's
    weight-gradient kernel for it ran one 64x64 tile per image: 0.39 ms per call at 56x56) and mm_channel_sum_nchw for the
    bias gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        B, C, H, W = x.shape
        w = weight.view(weight.shape[0], C)
        x3 = x.reshape(B, C, H * W)
        out = _gemm(w, x3) if bias is None else torch.baddbmm(bias.view(1, -1, 1), w.unsqueeze(0).expand(B, -1, -1), x3)
        ctx.save_for_backward(x3, w)
        ctx.shape = (B, C, H, W, bias is not None, weight.shape)
        return out.view(B, -1, H, W)

    @staticmethod
    def backward(ctx, dy):
        x3, w = ctx.saved_tensors
        B, C, H, W, has_bias, wshape = ctx.shape
        dy3 = dy.contiguous().view(B, -1, H * W)
        dx = _gemm(w.t(), dy3).view(B, C, H, W) if ctx.needs_input_grad[0] else None
        dw = sum_lead(_gemm(dy3, x3.transpose(1, 2))).view(wshape) if ctx.needs_input_grad[1] else None
        db = _bias_grad(dy3) if (has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db


class BlockSplitFn(torch.autograd.Function):
    """SS_Conv_SSM prologue (MedMamba.py:350-352): inp (B,H,W,C) -> (left NCHW (B,C/2,H,W), LayerNorm(right) (B,H,W,C/2),
    inp itself for the residual add of :357).  The backward writes both halves of d(inp) in place — no chunk/cat copies,
    no strided-LayerNorm copy — and adds the gradient that arrives through the residual output in the same pass, so
    autograd never has to sum two gradients of the block input."""

    @staticmethod
    def forward(ctx, inp, gamma, beta, eps):
        inp = inp.float().contiguous()
        gamma, beta = gamma.float().contiguous(), beta.float().contiguous()
        B, H, W, C = inp.shape
        C2, P = C // 2, H * W
        dev = inp.device
        left = torch.empty((B, C2, H, W), device=dev, dtype=torch.float32)
        rn = torch.empty((B, H, W, C2), device=dev, dtype=torch.float32)
        mu = torch.empty((B * P,), device=dev, dtype=torch.float32)
        rstd = torch.empty((B * P,), device=dev, dtype=torch.float32)
        with _lib.device_guard(dev):
            rc = _lib.lib().mm_block_split_fwd(inp.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), None, left.data_ptr(),
                                               rn.data_ptr(), mu.data_ptr(), rstd.data_ptr(), B, P, C2, _stream())
        _lib.check(rc, "mm_block_split_fwd")
        ctx.save_for_backward(inp, gamma, mu, rstd)
        return left, rn, inp.view_as(inp)

    @staticmethod
    def backward(ctx, dleft, drn, dres):
        inp, gamma, mu, rstd = ctx.saved_tensors
        B, H, W, C = inp.shape
        C2, P = C // 2, H * W
        dev = inp.device
        dleft, drn = dleft.float().contiguous(), drn.float().contiguous()
        dres = None if dres is None else dres.float().contiguous()
        dinp = torch.empty_like(inp)
        lib = _lib.lib()
        ws = torch.empty((lib.mm_block_split_rows(B, P, C2), 2 * C2), device=dev, dtype=torch.float32)
        with _lib.device_guard(dev):
            rc = lib.mm_block_split_bwd(dleft.data_ptr(), drn.data_ptr(), None if dres is None else dres.data_ptr(),
                                        inp.data_ptr(), gamma.data_ptr(), mu.data_ptr(), rstd.data_ptr(), dinp.data_ptr(),
                                        ws.data_ptr(), B, P, C2, _stream())
        _lib.check(rc, "mm_block_split_bwd")
        s = sum_lead(ws)
        return dinp, s[:C2], s[C2:], None


def block_split_infer(inp, gamma, beta, eps, left_affine):
    """Inference form of block_split (no autograd): the left half additionally gets the per-channel affine
    left_affine = [scale C/2 | shift C/2] — the eval-mode BatchNorm2d that opens the conv branch (MedMamba.py:338) — while it
    is transposed to NCHW.  Returns (left NCHW, LayerNorm(right) NHWC)."""
    _need_hip(inp)
    inp = inp.float().contiguous()
    B, H, W, C = inp.shape
    C2, P = C // 2, H * W
    dev = inp.device
    left = torch.empty((B, C2, H, W), device=dev, dtype=torch.float32)
    rn = torch.empty((B, H, W, C2), device=dev, dtype=torch.float32)
    stats = torch.empty((2, B * P), device=dev, dtype=torch.float32)
    with _lib.device_guard(dev):
        rc = _lib.lib().mm_block_split_fwd(inp.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), left_affine.data_ptr(),
                                           left.data_ptr(), rn.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), B, P, C2,
                                           _stream())
    _lib.check(rc, "mm_block_split_fwd")
    return left, rn


def block_split(inp, gamma, beta, eps):
    """-> (left NCHW, LayerNorm(right) NHWC, inp for the residual add — use THIS alias as the residual input so that its
    gradient is folded into the backward kernel)."""
    _need_hip(inp)
    if inp.shape[-1] % 2 or inp.shape[-1] // 2 > 512:
        raise NotImplementedError("block_split: even channel count <= 1024 expected")
    return BlockSplitFn.apply(inp, gamma, beta, eps)


class PatchMergeLNFn(torch.autograd.Function):
    """PatchMerging2D's gather + LayerNorm(4C) (MedMamba.py:93-116) in one kernel each way: x (B, H, W, C) NHWC ->
    (B, H//2, W//2, 4C) normalised rows (the Linear of :117 follows as a plain GEMM).  The backward writes d(x) straight into
    the gathered positions; for odd H / W the cropped row / column gets zeros (MedMamba.py:97-111)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x, gamma, beta = x.float().contiguous(), gamma.float().contiguous(), beta.float().contiguous()
        B, H, W, C = x.shape
        h2, w2 = H // 2, W // 2
        dev = x.device
        out = torch.empty((B, h2, w2, 4 * C), device=dev, dtype=torch.float32)
        stats = torch.empty((2, B * h2 * w2), device=dev, dtype=torch.float32)
        with _lib.device_guard(dev):
            rc = _lib.lib().mm_patch_merge_ln_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), out.data_ptr(),
                                                  stats[0].data_ptr(), stats[1].data_ptr(), B, H, W, C, _stream())
        _lib.check(rc, "mm_patch_merge_ln_fwd")
        ctx.save_for_backward(x, gamma, stats)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, gamma, stats = ctx.saved_tensors
        B, H, W, C = x.shape
        dev = x.device
        dy = dy.float().contiguous()
        lib = _lib.lib()
        dinp = (torch.zeros_like(x) if (H % 2 or W % 2) else torch.empty_like(x))
        ws = torch.empty((lib.mm_patch_merge_ln_rows(B, H, W), 2, 4 * C), device=dev, dtype=torch.float32)
        with _lib.device_guard(dev):
            rc = lib.mm_patch_merge_ln_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                                           dinp.data_ptr(), ws.data_ptr(), B, H, W, C, _stream())
        _lib.check(rc, "mm_patch_merge_ln_bwd")
        s = sum_lead(ws)
        return dinp, s[0], s[1], None


def patch_merge_ln_supported(C):
    return bool(_lib.lib().mm_patch_merge_ln_supported(int(C)))


def patch_merge_ln(x, gamma, beta, eps):
    _need_hip(x)
    return PatchMergeLNFn.apply(x, gamma, beta, eps)


class NchwLNRowsFn(torch.autograd.Function):
    """PatchEmbed2D's permute + LayerNorm (MedMamba.py:70-76) in one kernel each way: x (B, C, H, W) contiguous NCHW (the strided
    conv's output) -> (B, H, W, C) normalised NHWC rows; the backward returns d(x) in NCHW for the conv's own backward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x, gamma, beta = x.float().contiguous(), gamma.float().contiguous(), beta.float().contiguous()
        B, C, H, W = x.shape
        dev = x.device
        out = torch.empty((B, H, W, C), device=dev, dtype=torch.float32)
        stats = torch.empty((2, B * H * W), device=dev, dtype=torch.float32)
        with _lib.device_guard(dev):
            rc = _lib.lib().mm_nchw_ln_rows_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), out.data_ptr(),
                                                stats[0].data_ptr(), stats[1].data_ptr(), B, C, H * W, _stream())
        _lib.check(rc, "mm_nchw_ln_rows_fwd")
        ctx.save_for_backward(x, gamma, stats)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, gamma, stats = ctx.saved_tensors
        B, C, H, W = x.shape
        dev = x.device
        dy = dy.float().contiguous()
        lib = _lib.lib()
        dx = torch.empty_like(x)
        ws = torch.empty((lib.mm_nchw_ln_rows_ws_rows(B, H * W), 2, C), device=dev, dtype=torch.float32)
        with _lib.device_guard(dev):
            rc = lib.mm_nchw_ln_rows_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                                         dx.data_ptr(), ws.data_ptr(), B, C, H * W, _stream())
        _lib.check(rc, "mm_nchw_ln_rows_bwd")
        s = sum_lead(ws)
        return dx, s[0], s[1], None


def nchw_ln_rows_supported(C, HW=0):
    """C <= 512 channels; one image's C planes of HW positions within 2 GB (the kernels address them with 32-bit byte offsets)."""
    return bool(_lib.lib().mm_nchw_ln_rows_supported(int(C))) and int(C) * int(HW) * 4 < 0x7fffffff


def nchw_ln_rows(x, gamma, beta, eps):
    _need_hip(x)
    return NchwLNRowsFn.apply(x, gamma, beta, eps)


_OWN_CONV = os.environ.get("MM_OWN_CONV", "0") == "1"     # experiment: dense 3x3 convs (forward) through csrc/conv.hip — needs the
                                                          # experiments build (python -m medmamba_amd.build --experiments)


class Conv3x3Fn(torch.autograd.Function):
    """nn.Conv2d(C, K, 3, padding=1) forward through mm_conv3x3_fwd (fp32 MFMA implicit GEMM on NCHW, bias in the epilogue);
    with want_stats the kernel also emits the per-channel (count, mean, M2) partials of its output — the statistics pass of the
    training-mode BatchNorm that follows (BNReluFn(partials=...)).  The backward is MIOpen's (data and weight gradient) plus
    the channel-sum bias gradient, exactly as ConvBiasFn."""

    @staticmethod
    def forward(ctx, x, w, b, want_stats):
        x, w = x.float().contiguous(), w.float().contiguous()
        b = None if b is None else b.float().contiguous()
        B, C, H, W = x.shape
        K = w.shape[0]
        lib = _lib.exp_lib()          # an experiment: lives in the experiments build of the library only
        y = torch.empty((B, K, H, W), device=x.device, dtype=torch.float32)
        stats = torch.empty((lib.mm_conv3x3_fwd_tiles(B, H, W), K, 3), device=x.device, dtype=torch.float32) if want_stats else None
        with _lib.device_guard(x.device):
            rc = lib.mm_conv3x3_fwd(x.data_ptr(), w.data_ptr(), None if b is None else b.data_ptr(), None, 0, y.data_ptr(),
                                    None if stats is None else stats.data_ptr(), B, C, K, H, W, _stream())
        _lib.check(rc, "mm_conv3x3_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        if stats is None:
            return y
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, dstats=None):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx, dw, _ = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                        [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False])
        db = _bias_grad(dy) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None


def own_conv3x3_ok(x, conv):
    return (_OWN_CONV and _lib.exp_lib() is not None and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.kernel_size == (3, 3)
            and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.padding_mode == "zeros")


class BNReluFn(torch.autograd.Function):
    """Training-mode nn.BatchNorm2d (+ the nn.ReLU behind it when relu=True) on contiguous NCHW tensors — the conv branch's
    BN2+ReLU / BN3+ReLU / BN1 (MedMamba.py:338-344).  Updates running_mean / running_var in place like the module does; the
    caller advances num_batches_tracked."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu, partials=None, pre_bias=None):
        x = x.float().contiguous()
        gamma, beta = gamma.float().contiguous(), beta.float().contiguous()
        pre_bias = None if pre_bias is None else pre_bias.float().contiguous()
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // (B * C)
        dev = x.device
        lib = _lib.lib()
        y = torch.empty_like(x)
        stats = torch.empty((2, C), device=dev, dtype=torch.float32)
        rm = None if running_mean is None else running_mean.data_ptr()
        rv = None if running_var is None else running_var.data_ptr()
        if partials is not None:       # batch statistics already produced by the kernel that wrote x (mm_conv3x3_fwd): apply only
            assert partials.is_contiguous() and partials.shape[1:] == (C, 3)
            with _lib.device_guard(dev):
                rc = lib.mm_bn_relu_fwd_stats(x.data_ptr(), partials.data_ptr(), partials.shape[0], gamma.data_ptr(), beta.data_ptr(),
                                              float(eps), float(momentum), rm, rv, y.data_ptr(), stats[0].data_ptr(),
                                              stats[1].data_ptr(), int(bool(relu)), B, C, HW, _stream())
            _lib.check(rc, "mm_bn_relu_fwd_stats")
        else:
            ws = torch.empty((3 * C * lib.mm_bn_splits(B, C, HW),), device=dev, dtype=torch.float32)
            with _lib.device_guard(dev):
                rc = lib.mm_bn_relu_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), float(momentum), rm, rv,
                                        y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), ws.data_ptr(),
                                        None if pre_bias is None else pre_bias.data_ptr(), int(bool(relu)), B, C, HW, _stream())
            _lib.check(rc, "mm_bn_relu_fwd")
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.relu = bool(relu)
        ctx.has_pre_bias = pre_bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // (B * C)
        dev = x.device
        dy = dy.float().contiguous()
        lib = _lib.lib()
        dx = torch.empty_like(x)
        # pre_bias is the bias of the convolution that produced x: its gradient is the per-channel sum of dx, which the
        # one-kernel form (small planes) returns for free; larger planes take the channel-sum kernel
        want_db = ctx.has_pre_bias and ctx.needs_input_grad[9]
        fused = bool(lib.mm_bn_fused(B, C, HW))
        dgb = torch.empty((3, C), device=dev, dtype=torch.float32)
        ws = dgb if fused else torch.empty((2 * C * lib.mm_bn_splits(B, C, HW),), device=dev, dtype=torch.float32)
        with _lib.device_guard(dev):
            rc = lib.mm_bn_relu_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), stats[0].data_ptr(),
                                    stats[1].data_ptr(), dx.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(), ws.data_ptr(),
                                    dgb[2].data_ptr() if (want_db and fused) else None, int(ctx.relu), B, C, HW, _stream())
        _lib.check(rc, "mm_bn_relu_bwd")
        db = None
        if want_db:
            db = dgb[2] if fused else _bias_grad(dx)
        return dx, dgb[0], dgb[1], None, None, None, None, None, None, db


class ConvBranchFn(torch.autograd.Function):
    """The conv branch of SS_Conv_SSM (MedMamba.py:338-345; the trailing ReLU and the 1x1 conv's bias are shuffle_residual's) as one
    autograd node, its forward and backward ONE call each into csrc_host/ss2d_host.cpp (conv_branch_fwd / _bwd): the launches of
    BNReluFn x 3, ConvBiasFn(add_bias=False) x 2 and PointwiseConvFn, without the interpreter and five autograd nodes in between.
    Running statistics are updated by the kernels; the caller (conv_branch_native) advances the counters and version numbers."""

    @staticmethod
    def forward(ctx, x, g1, b1, w1, cb1, g2, b2, w2, cb2, g3, b3, w3, buffers, eps_mom):
        from . import _host
        with _lib.device_guard(x.device):
            bt = BRANCH_TIMER.start()
            res = _host.module().conv_branch_fwd(x, [g1, b1, buffers[0], buffers[1]], w1, cb1, [g2, b2, buffers[2], buffers[3]], w2, cb2,
                                                 [g3, b3, buffers[4], buffers[5]], w3, eps_mom, _stream())
            BRANCH_TIMER.stop("conv_fwd", bt)
        ctx.save_for_backward(x, g1, b1, w1, g2, b2, w2, g3, b3, w3, *res[1:])
        return res[0]

    @staticmethod
    def backward(ctx, dout):
        from . import _host
        t = ctx.saved_tensors
        with _lib.device_guard(dout.device):
            bt = BRANCH_TIMER.start()
            g = _host.module().conv_branch_bwd(dout, *t, _stream())
            BRANCH_TIMER.stop("conv_bwd", bt)
        return (*g, None, None)


def conv_branch_native(mods, x):
    """`mods` = the conv branch without its trailing ReLU (8 modules).  Returns the pre-activation of that ReLU WITHOUT the 1x1 conv's
    bias through ConvBranchFn, or None when the branch is not the reference's BN-conv3x3-BN-ReLU-conv3x3-BN-ReLU-conv1x1 in
    training mode on fp32 HIP tensors (the caller then walks the modules one by one)."""
    from . import _host
    nn = torch.nn
    if (_host.module() is None or len(mods) != 8 or not torch.is_grad_enabled() or not x.is_cuda or x.dtype != torch.float32
            or x.dim() != 4 or _OWN_CONV or torch.cuda.is_current_stream_capturing()):
        return None
    bn1, c1, bn2, r2, c2, bn3, r3, c3 = mods
    if type(r2) is not nn.ReLU or type(r3) is not nn.ReLU:
        return None
    for bn in (bn1, bn2, bn3):
        if not (type(bn) is nn.BatchNorm2d and bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None
                and bn.num_batches_tracked is not None and bn.weight.dtype == torch.float32):
            return None
    for c in (c1, c2):
        if not (type(c) is nn.Conv2d and c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.dilation == (1, 1)
                and c.groups == 1 and c.padding_mode == "zeros" and c.bias is not None and c.weight.dtype == torch.float32):
            return None
    if not (type(c3) is nn.Conv2d and c3.kernel_size == (1, 1) and c3.stride == (1, 1) and c3.padding == (0, 0) and c3.groups == 1
            and c3.padding_mode == "zeros" and c3.weight.dtype == torch.float32):
        return None
    bns = (bn1, bn2, bn3)
    if _DEFERRED_COUNTERS is not None:
        _DEFERRED_COUNTERS.extend(bn.num_batches_tracked for bn in bns)
    else:
        torch._foreach_add_([bn.num_batches_tracked for bn in bns], 1)
    buffers = [t for bn in bns for t in (bn.running_mean, bn.running_var)]
    y = ConvBranchFn.apply(x.contiguous(), bn1.weight, bn1.bias, c1.weight, c1.bias, bn2.weight, bn2.bias, c2.weight, c2.bias,
                           bn3.weight, bn3.bias, c3.weight, buffers, [bn1.eps, bn1.momentum, bn2.eps, bn2.momentum, bn3.eps, bn3.momentum])
    torch.autograd.graph.increment_version(buffers)      # updated through raw pointers (see bn_relu_train)
    return y


_DEFERRED_COUNTERS = None      # list while a caller batches the BatchNorm step counters of a whole forward (VSSM.forward_backbone)


class deferred_bn_counters:
    """Inside this context the `num_batches_tracked += 1` of every BatchNorm that bn_relu_train runs is collected and applied
    by ONE multi-tensor add on exit (42 one-element launches per MedMamba-S step otherwise).  Only for the default
    momentum semantics; a BatchNorm with momentum=None needs its counter at once and keeps the immediate add, and so does a
    forward that is being captured into a hipGraph (the add must be part of what is replayed)."""

    def __enter__(self):
        global _DEFERRED_COUNTERS
        self.prev, _DEFERRED_COUNTERS = _DEFERRED_COUNTERS, []
        return self

    def __exit__(self, *exc):
        global _DEFERRED_COUNTERS
        pending, _DEFERRED_COUNTERS = _DEFERRED_COUNTERS, self.prev
        if pending:
            torch._foreach_add_(pending, 1)
        return False


class immediate_bn_counters:
    """Suspends deferred_bn_counters: for code that snapshots / restores BatchNorm buffers around forward passes (the warm-up and
    capture passes of a hipGraph build) and needs every counter update to have happened when the pass returns."""

    def __enter__(self):
        global _DEFERRED_COUNTERS
        self.prev, _DEFERRED_COUNTERS = _DEFERRED_COUNTERS, None
        return self

    def __exit__(self, *exc):
        global _DEFERRED_COUNTERS
        _DEFERRED_COUNTERS = self.prev
        return False


def bn_relu_train(x, bn, relu, partials=None, pre_bias=None):
    """`bn` (an nn.BatchNorm2d in training mode, affine, default momentum semantics) applied to x, optionally followed by
    ReLU, through BNReluFn; num_batches_tracked advances as in the module's own forward."""
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        if _DEFERRED_COUNTERS is not None and bn.momentum is not None and not torch.cuda.is_current_stream_capturing():
            _DEFERRED_COUNTERS.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
        momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
    else:
        momentum = 0.0 if bn.momentum is None else bn.momentum
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    y = BNReluFn.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, momentum, relu, partials, pre_bias)
    if rm is not None:
        # the kernel updated the running statistics through raw pointers: bump their version counters like an in-place torch
        # op would, so that anything keyed on them (SS_Conv_SSM._eval_fold) sees the change even if only the BatchNorm modules
        # were switched to train() and back (e.g. a BN recalibration pass)
        torch.autograd.graph.increment_version((rm, rv))
    return y
