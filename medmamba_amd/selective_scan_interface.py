"""`selective_scan_fn` — same name, signature and error behaviour as
`mamba_ssm.ops.selective_scan_interface.selective_scan_fn`, the operator MedMamba calls at
MedMamba.py:273-279 (imported at MedMamba.py:12), backed by the gfx950 HIP kernels in
libmedmamba_hip.so through the C ABI (include/medmamba_hip.h).

Only the variant that exists on the MedMamba path is implemented natively:
real A, B/C of shape (batch, G, N, L), z=None, return_last_state=False, N = 16, fp32.
Anything else raises NotImplementedError; CPU tensors raise RuntimeError (no CPU fallback).
"""
import ctypes

import torch

from . import _lib


def _ptr(t):
    return None if t is None else t.data_ptr()


class _KernelTimer:
    """Optional hipEvent bracket around every kernel launch (used by bench.py for the roofline figures).
    Events are recorded on torch's current stream, which is the stream the kernels are launched on;
    recording is asynchronous, elapsed times are read after the caller has synchronised."""

    def __init__(self):
        self.enabled = False
        self.records = []          # (tag, start_event, end_event, algorithmic_bytes, state_steps)

    def start(self):
        if not self.enabled:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def stop(self, tag, start, nbytes, nsteps=0):
        if start is None:
            return
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        self.records.append((tag, start, end, nbytes, nsteps))

    def pair(self, tag, nbytes, nsteps=0):
        """For a launch that happens inside a native call: two timing events, as raw hipEvent_t handles, which the callee
        records around the kernel on the launch stream (mm_event_record).  (0, 0) when the timer is off."""
        if not self.enabled:
            return 0, 0
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        e.record()             # torch creates the hipEvent at its first record; the callee records both again
        self.records.append((tag, s, e, nbytes, nsteps))
        return s.cuda_event, e.cuda_event

    def summary(self):
        """{tag: dict(calls, ms, bytes)} — call only after torch.cuda.synchronize()."""
        out = {}
        for tag, s, e, nb, ns in self.records:
            d = out.setdefault(tag, dict(calls=0, ms=0.0, bytes=0, state_steps=0))
            d["calls"] += 1
            d["ms"] += s.elapsed_time(e)
            d["bytes"] += nb
            d["state_steps"] += ns
        return out


KERNEL_TIMER = _KernelTimer()
_BWD_VARIANT = int(__import__("os").environ.get("MM_BWD_VARIANT", "0"), 0)   # tuning knob: waves per workgroup << 16
_FWD_VARIANT = int(__import__("os").environ.get("MM_FWD_VARIANT", "0"), 0)   # tuning knob: forward kernel / states per lane when a call passes 0


def scan_bytes_fwd(batch, dim, L, N, G):
    """Algorithmic bytes of one forward call at the operator boundary: read u, delta; write out;
    read B, C; A, D, delta_bias once."""
    return 4 * batch * L * (3 * dim + 2 * G * N) + 4 * (dim * N + 2 * dim)


def scan_bytes_bwd(batch, dim, L, N, G):
    """Algorithmic bytes of one backward call: read u, delta, dout; write du, ddelta; read B, C;
    write dB, dC; parameters + parameter gradients."""
    return 4 * batch * L * (5 * dim + 4 * G * N) + 8 * (dim * N + 2 * dim)


def _check_inputs(u, delta, A, B, C, D, z, delta_bias, return_last_state):
    if z is not None:
        raise NotImplementedError("z gating is not on the MedMamba path (MedMamba.py:275 passes z=None)")
    if return_last_state:
        raise NotImplementedError("return_last_state=True is not on the MedMamba path (MedMamba.py:278)")
    if A.is_complex():
        raise NotImplementedError("complex A is not on the MedMamba path (MedMamba.py:28)")
    if B.dim() != 4 or C.dim() != 4:
        raise NotImplementedError("only variable, grouped B/C of shape (batch, G, N, L) are supported")
    if not u.is_cuda:
        raise RuntimeError("selective_scan_fn: tensors must live on a HIP device "
                           "(medmamba_amd has no CPU path; the CPU oracle lives under oracle/ for tests only)")
    batch, dim, L = u.shape
    N, G = A.shape[1], B.shape[1]
    if delta.shape != u.shape or A.shape[0] != dim or B.shape != (batch, G, N, L) or C.shape != B.shape:
        raise RuntimeError(f"selective_scan_fn: inconsistent shapes u{tuple(u.shape)} delta{tuple(delta.shape)} "
                           f"A{tuple(A.shape)} B{tuple(B.shape)} C{tuple(C.shape)}")
    if dim % G != 0:
        raise RuntimeError("selective_scan_fn: dim must be divisible by the number of B/C groups")
    if N < 1:
        raise RuntimeError("selective_scan_fn: d_state must be >= 1")
    for name, t in (("D", D), ("delta_bias", delta_bias)):
        if t is not None and t.shape != (dim,):
            raise RuntimeError(f"selective_scan_fn: {name} must have shape ({dim},)")


_KERNEL_STATES = 16     # mm::kNState: states per channel inside the kernels


def _f32_rows(t):
    """fp32 with unit stride in the last dim (mamba_ssm copies in exactly this case too)."""
    t = t.float()
    return t if t.stride(-1) == 1 else t.contiguous()


def _fill_common(a, u, delta, A, B, C, D, delta_bias, delta_softplus):
    batch, L = u.shape[0], u.shape[2]
    dim = A.shape[0]
    a.batch, a.dim, a.L, a.N, a.G = batch, dim, L, A.shape[1], B.shape[1]
    a.delta_softplus = int(bool(delta_softplus))
    a.u, a.delta, a.A, a.B, a.C = u.data_ptr(), _ptr(delta), A.data_ptr(), B.data_ptr(), C.data_ptr()
    a.D, a.delta_bias = _ptr(D), _ptr(delta_bias)
    a.u_sb, a.u_sd = u.stride(0), u.stride(1)
    if delta is not None:
        a.delta_sb, a.delta_sd = delta.stride(0), delta.stride(1)
    a.B_sb, a.B_sg, a.B_sn = B.stride(0), B.stride(1), B.stride(2)
    a.C_sb, a.C_sg, a.C_sn = C.stride(0), C.stride(1), C.stride(2)


def dt_fusable(R, L, dts):
    """True if mm_scan_fwd can compute delta = dt_w @ dts itself (mm_scan_args.dt_w): rank within the kernel's limit,
    vector path (L % 4 == 0, 16-B aligned rows with strides that are multiples of 4 elements)."""
    return (0 < R <= _lib.lib().mm_scan_dt_max() and L % 4 == 0 and dts.data_ptr() % 16 == 0 and dts.stride(3) == 1
            and all(st % 4 == 0 for st in dts.stride()[:3]))


def _launch_fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, want_chk, variant=0, shared=(0, 0, 0), dt=None):
    """mm_scan_fwd on torch's current stream. Returns (out (batch, G*H, L), x_chk or None).
    dt = (dts (batch, G, R, L) view, dt_w (dim, R) contiguous): the dt projection is fused into the kernel and `delta` is None."""
    batch, L = u.shape[0], u.shape[2]
    dim = A.shape[0]
    out = torch.empty((batch, dim, L), device=u.device, dtype=torch.float32)
    x_chk = None
    if want_chk:
        chunk = _lib.scan_chunk()
        x_chk = torch.empty((batch, (L + chunk - 1) // chunk, dim, A.shape[1]), device=u.device, dtype=torch.float32)
    a = _lib.ScanArgs()
    _fill_common(a, u, delta, A, B, C, D, delta_bias, delta_softplus)
    a.out, a.x_chk, a.variant = out.data_ptr(), _ptr(x_chk), int(variant) or _FWD_VARIANT
    a.u_groups, a.u_map, a.rev_mask = shared
    if dt is not None:
        dts, dt_w = dt
        assert delta is None and dt_w.is_contiguous() and dt_w.shape[0] == dim and dts.shape == (batch, B.shape[1], dt_w.shape[1], L)
        a.dt_w, a.dts, a.dt_rank = dt_w.data_ptr(), dts.data_ptr(), dt_w.shape[1]
        a.dts_sb, a.dts_sg, a.dts_sn = dts.stride(0), dts.stride(1), dts.stride(2)
    with _lib.device_guard(u.device):
        t0 = KERNEL_TIMER.start()
        rc = _lib.lib().mm_scan_fwd(a, _lib.raw_stream())
        KERNEL_TIMER.stop("scan_fwd", t0, scan_bytes_fwd(batch, dim, L, A.shape[1], B.shape[1]), batch * dim * L * A.shape[1])
    _lib.check(rc, "mm_scan_fwd")
    return out, x_chk


def _chunks_of(t):
    """(nchunks, chunk, chunk_stride) if the view `t` is a sequence of dense chunks a constant stride apart (the B / C rows of
    d(x_dbl): 32 rows of every direction, the dt rows in between), else None."""
    dims = sorted(((st, sz) for st, sz in zip(t.stride(), t.shape) if sz > 1), reverse=True)
    chunk, i = 1, len(dims)
    while i > 0 and dims[i - 1][0] == chunk:
        chunk *= dims[i - 1][1]; i -= 1
    if i == 0:
        return (1, chunk, chunk)
    stride, n = dims[i - 1][0], 1
    for j in range(i - 1, -1, -1):                  # the outer dimensions must flatten to one index with that stride
        if dims[j][0] != stride * n:
            return None
        n *= dims[j][1]
    return (n, chunk, stride) if stride >= chunk else None


def _like_strided(dst, lead):
    """Uninitialised fp32 (lead, *dst.shape) whose planes have the same dimension order in memory as the view `dst` (dense)."""
    order = sorted(range(dst.dim()), key=lambda d: (-dst.stride(d), d))
    shape = [dst.shape[d] for d in order]
    t = torch.empty([lead] + shape, device=dst.device, dtype=torch.float32)
    inv = [0] * dst.dim()
    for pos, d in enumerate(order):
        inv[d] = pos + 1
    return t.permute([0] + inv)


def _launch_bwd(u, delta, A, B, C, D, delta_bias, x_chk, dout, delta_softplus, shared=(0, 0, 0), dBC_dst=None, parts=None,
                channel_major=False):
    """mm_scan_bwd on torch's current stream. Returns du (per group), ddelta, dA, dB, dC, dD, dbias.

    No atomics and nothing to zero-fill (ABI 19): dA / dD / dbias are written per batch item into a partial buffer
    (mm_scan_args.dpar_sb) and dB / dC either in place (one workgroup per direction) or into per-workgroup partial planes
    (mm_scan_args.dBC_sc) that are summed here — every sum runs in a fixed order, so two runs give the same bits.

    dBC_dst: optional (batch, G, 2N, L) fp32 view with unit stride along L — rows [0, N) receive dB, rows [N, 2N) dC (e.g.
      the B | C row block of the gradient of x_dbl); fully overwritten.  None: allocated here.
    parts: optional (buffer (batch, S), offset of dA, offset of dD, offset of dbias) — the caller sums over the batch itself
      (mm_ss2d_pack_bwd); dA / dD / dbias are then returned as None.  None: allocated and summed here.
    channel_major: du / ddelta are returned as (batch, dim, L) views of (dim, batch, L) storage; dout must then have the
      same channel stride (batch * L)."""
    batch, dim, L = delta.shape
    G, N = B.shape[1], A.shape[1]
    dev = u.device
    if channel_major:
        du = torch.empty((dim, batch, L), device=dev, dtype=torch.float32).permute(1, 0, 2)
        ddelta = torch.empty((dim, batch, L), device=dev, dtype=torch.float32).permute(1, 0, 2)
    else:
        du = torch.empty((batch, dim, L), device=dev, dtype=torch.float32)
        ddelta = torch.empty((batch, dim, L), device=dev, dtype=torch.float32)
    assert dout.stride(2) == 1 and dout.stride(1) == du.stride(1), "dout and du/ddelta must share the channel stride"
    own_parts = parts is None
    if own_parts:
        S = dim * N + 2 * dim
        parts = (torch.empty((batch, S), device=dev, dtype=torch.float32), 0, dim * N, dim * N + dim)
    pbuf, oA, oD, ob = parts
    assert pbuf.is_contiguous() and pbuf.shape[0] == batch
    if dBC_dst is None:
        dBC_dst = torch.empty((batch, G, 2 * N, L), device=dev, dtype=torch.float32)
    assert dBC_dst.shape == (batch, G, 2 * N, L) and dBC_dst.stride(3) == 1
    a = _lib.ScanArgs()
    _fill_common(a, u, delta, A, B, C, D, delta_bias, delta_softplus)
    a.x_chk, a.dout = x_chk.data_ptr(), dout.data_ptr()
    a.du, a.ddelta = du.data_ptr(), ddelta.data_ptr()
    a.dA = pbuf.data_ptr() + 4 * oA
    a.dD = None if D is None else pbuf.data_ptr() + 4 * oD
    a.ddelta_bias = None if delta_bias is None else pbuf.data_ptr() + 4 * ob
    a.dpar_sb = pbuf.stride(0)
    a.u_groups, a.u_map, a.rev_mask = shared
    a.dout_sb, a.dud_sb, a.o_sd = dout.stride(0), du.stride(0), du.stride(1)
    a.variant = _BWD_VARIANT

    def point_dBC(t, plane_stride):
        a.dB, a.dC = t.data_ptr(), t.data_ptr() + 4 * N * t.stride(2)
        a.dB_sb, a.dB_sg, a.dB_sn = t.stride(0), t.stride(1), t.stride(2)
        a.dC_sb, a.dC_sg, a.dC_sn = t.stride(0), t.stride(1), t.stride(2)
        a.dBC_sc = plane_stride

    point_dBC(dBC_dst, 0)
    plan = (ctypes.c_int32 * 8)()
    lib = _lib.lib()
    _lib.check(lib.mm_scan_plan(a, 1, plan), "mm_scan_plan")
    W = plan[6]
    planes = None
    if W > 1:      # a direction is shared by W workgroups: one partial plane each, summed below
        planes = _like_strided(dBC_dst, W)
        point_dBC(planes[0], planes.stride(0))
    with _lib.device_guard(dev):
        t0 = KERNEL_TIMER.start()
        rc = lib.mm_scan_bwd(a, _lib.raw_stream())
        KERNEL_TIMER.stop("scan_bwd", t0, scan_bytes_bwd(batch, dim, L, N, G), batch * dim * L * N)
    _lib.check(rc, "mm_scan_bwd")
    if planes is not None:
        ch = _chunks_of(dBC_dst)
        if ch is not None and 2 <= W <= 4096:      # mm_sum_lead_chunks: the planes are dense in dBC_dst's own dimension order
            with _lib.device_guard(dev):
                _lib.check(lib.mm_sum_lead_chunks(planes.data_ptr(), dBC_dst.data_ptr(), W, ch[0], ch[1], ch[2], _lib.raw_stream()),
                           "mm_sum_lead_chunks")
        else:
            torch.sum(planes, dim=0, out=dBC_dst)
    dA = dD = dbias = None
    if own_parts:
        sums = pbuf.sum(0)
        dA = sums[:dim * N].view(dim, N)
        dD = None if D is None else sums[dim * N:dim * N + dim]
        dbias = None if delta_bias is None else sums[dim * N + dim:]
    return du, ddelta, dA, dBC_dst[:, :, :N], dBC_dst[:, :, N:], dD, dbias


def _prep(u, delta, A, B, C, D, delta_bias):
    u, delta, B, C = _f32_rows(u), _f32_rows(delta), _f32_rows(B), _f32_rows(C)
    A = A.float().contiguous()
    D = None if D is None else D.float().contiguous()
    delta_bias = None if delta_bias is None else delta_bias.float().contiguous()
    return u, delta, A, B, C, D, delta_bias


class SelectiveScanFn(torch.autograd.Function):
    """Autograd wrapper around mm_scan_fwd / mm_scan_bwd (the selective_scan_fn operator).

    Saved for backward: u, delta, A, B, C, D, delta_bias and the state checkpoints x_chk
    (batch, ceil(L/16), dim, 16) that the forward kernel writes — the backward kernel recomputes the
    states of each 16-step chunk from them instead of storing all L x 16 states."""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, variant=0):
        u, delta, A, B, C, D, delta_bias = _prep(u, delta, A, B, C, D, delta_bias)
        need_grad = any(ctx.needs_input_grad[:7])
        out, x_chk = _launch_fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, need_grad, variant)
        if need_grad:
            ctx.save_for_backward(u, delta, A, B, C, D, delta_bias, x_chk)
            ctx.delta_softplus = delta_softplus
        return out

    @staticmethod
    def backward(ctx, dout):
        u, delta, A, B, C, D, delta_bias, x_chk = ctx.saved_tensors
        r = _launch_bwd(u, delta, A, B, C, D, delta_bias, x_chk, dout.float().contiguous(), ctx.delta_softplus)
        return (*r, None, None)


# SS2D direction order used by cross_scan_fn (kernel group g): g = 0 row-major forward, 1 row-major backward,
# 2 column-major forward, 3 column-major backward  (= reference directions k = 0, 2, 1, 3; MedMamba.py:256-257).
CROSS_SCAN_K_OF_G = (0, 2, 1, 3)
_CROSS_SHARED = (2, 0x1100, 0b1010)      # u_groups, u_map (block of g: 0,0,1,1), rev_mask (g = 1, 3 reversed)


class CrossScanFn(torch.autograd.Function):
    """4-direction selective scan of SS2D without materialising the cross-scan (MedMamba.py:256-257, 273-286).

    u2:    (batch, 2*D, L)  channel block 0 = image in row-major order, block 1 = column-major order
    delta: (batch, 4*D, L), B, C: (batch, 4, N, L) — per direction g (order CROSS_SCAN_K_OF_G), all stored in
           POSITION order of their image order; the backward directions (g = 1, 3) are walked in reverse by the kernel
    returns y2 (batch, 2*D, L): block 0 = sum of the two row-major directions, block 1 = sum of the two
           column-major directions, both in position order (the un-flip of MedMamba.py:282 never happens).
    The sum over a direction pair is linear, so the backward feeds the same gradient block to both directions
    (the kernel reads dout with the u block mapping) and sums the two du blocks."""

    @staticmethod
    def forward(ctx, u2, delta, A, B, C, D, delta_bias):
        u2, delta, A, B, C, D, delta_bias = _prep(u2, delta, A, B, C, D, delta_bias)
        need_grad = any(ctx.needs_input_grad)
        out, x_chk = _launch_fwd(u2, delta, A, B, C, D, delta_bias, True, need_grad, 0, _CROSS_SHARED)
        bsz, dim, L = out.shape
        o = out.view(bsz, 2, 2, dim // 4, L)
        y2 = (o[:, :, 0] + o[:, :, 1]).view(bsz, dim // 2, L)
        if need_grad:
            ctx.save_for_backward(u2, delta, A, B, C, D, delta_bias, x_chk)
        return y2

    @staticmethod
    def backward(ctx, dy2):
        u2, delta, A, B, C, D, delta_bias, x_chk = ctx.saved_tensors
        du, ddelta, dA, dB, dC, dD, dbias = _launch_bwd(u2, delta, A, B, C, D, delta_bias, x_chk,
                                                        dy2.float().contiguous(), True, _CROSS_SHARED)
        bsz, dim, L = du.shape
        d = du.view(bsz, 2, 2, dim // 4, L)
        du2 = (d[:, :, 0] + d[:, :, 1]).view(bsz, dim // 2, L)
        return du2, ddelta, dA, dB, dC, dD, dbias


def cross_scan_fn(u2, delta, A, B, C, D, delta_bias):
    """See CrossScanFn. HIP tensors only."""
    if not u2.is_cuda:
        raise RuntimeError("cross_scan_fn: tensors must live on a HIP device (medmamba_amd has no CPU path)")
    if A.shape[1] != 16 or B.shape[1] != 4 or delta.shape[1] != 2 * u2.shape[1]:
        raise NotImplementedError("cross_scan_fn: expects 4 directions, d_state 16, delta with 4*D channels")
    return CrossScanFn.apply(u2, delta, A, B, C, D, delta_bias)


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """Drop-in for mamba_ssm's selective_scan_fn on the variant MedMamba uses (MedMamba.py:273-279).

    u, delta: (batch, dim, L); A: (dim, N); B, C: (batch, G, N, L) (may be non-contiguous views with
    unit stride along L); D, delta_bias: (dim,).  Returns (batch, dim, L) float32, contiguous.

    The kernels hold N = 16 states per channel (one per lane of a 16-lane group; every MedMamba block uses d_state = 16,
    MedMamba.py:329,457).  The states of one channel never interact — y_t = sum_n C_n,t x_n,t + D u_t — so any other
    d_state runs as ceil(N / 16) launches over 16-state slices whose outputs add up: a short slice is filled with states
    that have B = C = 0 (they stay zero and contribute nothing to y or to any gradient), D rides on the first slice only.
    Autograd sums du / ddelta / dbias over the slices and cuts dA / dB / dC back to the real states."""
    _check_inputs(u, delta, A, B, C, D, z, delta_bias, return_last_state)
    N = A.shape[1]
    if N == _KERNEL_STATES:
        return SelectiveScanFn.apply(u, delta, A, B, C, D, delta_bias, delta_softplus, 0)
    y = None
    for n0 in range(0, N, _KERNEL_STATES):
        n1 = min(N, n0 + _KERNEL_STATES)
        Ak, Bk, Ck = A[:, n0:n1], B[:, :, n0:n1], C[:, :, n0:n1]
        fill = _KERNEL_STATES - (n1 - n0)
        if fill:
            Ak = torch.cat([Ak.float(), Ak.new_full((A.shape[0], fill), -1.0, dtype=torch.float32)], dim=1)
            zeros = B.new_zeros((B.shape[0], B.shape[1], fill, B.shape[3]), dtype=torch.float32)
            Bk, Ck = torch.cat([Bk.float(), zeros], dim=2), torch.cat([Ck.float(), zeros], dim=2)
        yk = SelectiveScanFn.apply(u, delta, Ak, Bk, Ck, D if n0 == 0 else None, delta_bias, delta_softplus, 0)
        y = yk if y is None else y + yk
    return y
