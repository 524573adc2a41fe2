"""ORACLE — test infrastructure only.  Never imported by the product (`medmamba_amd/`).

CPU restatement of the selective-scan arithmetic that MedMamba calls through
`mamba_ssm.ops.selective_scan_interface.selective_scan_fn` (`MedMamba.py:12, 273-279`).
mamba_ssm (pinned ==1.0.1 in the reference's `README.md:19`) is a CUDA-only third-party
package that is absent from /root/reference and from this image, so the arithmetic is
restated from the only in-tree statement of it: the `selective_scan_ref` body quoted inside
`temp.py:57-139` (docstrings under `if False:` in `flops_selective_scan_ref`).

Parity status of THIS function: "parity unpinned" by the reference's own tests (it has none,
SURVEY.md §4/§8c).  It is pinned (a) line-by-line to the quoted text, (b) against the C
restatement in `oracle/selective_scan_ref.c` (fp32 and fp64), and (c) everything *around* it
(SS2D, SS_Conv_SSM, PatchMerging2D, VSSM) is pinned by fixtures produced by executing the real
`/root/reference/MedMamba.py` with this function plugged in (`tools/gen_golden.py`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))


def selective_scan_ref(u, delta, A, B, C, D=None, z=None, delta_bias=None,
                       delta_softplus=False, return_last_state=False):
    """Pure-PyTorch time loop; follows the quote at temp.py:57-139 statement by statement.

    u, delta: (batch, dim, L); A: (dim, N) real; B, C: (batch, G, N, L) [or (batch, N, L) /
    (dim, N)]; D, delta_bias: (dim,).  Returns (batch, dim, L) in u's dtype.
    """
    dtype_in = u.dtype                                   # temp.py:58
    u = u.float()                                        # :59
    delta = delta.float()                                # :60
    if delta_bias is not None:                           # :61-62
        delta = delta + delta_bias[..., None].float()
    if delta_softplus:                                   # :63-64  (F.softplus: beta=1, threshold=20)
        delta = F.softplus(delta)
    batch, dim, dstate = u.shape[0], A.shape[0], A.shape[1]   # :65
    is_variable_B = B.dim() >= 3                         # :66
    is_variable_C = C.dim() >= 3                         # :67
    if A.is_complex():                                   # :68-72 — never used by MedMamba (MedMamba.py:28)
        raise NotImplementedError("complex A is not on the MedMamba path")
    B = B.float()                                        # :74
    C = C.float()                                        # :75
    x = A.new_zeros((batch, dim, dstate))                # :76
    ys = []                                              # :77
    deltaA = torch.exp(torch.einsum('bdl,dn->bdln', delta, A))          # :88
    if not is_variable_B:                                               # :89-90
        deltaB_u = torch.einsum('bdl,dn,bdl->bdln', delta, B, u)
    else:
        if B.dim() == 3:                                                # :92-93
            deltaB_u = torch.einsum('bdl,bnl,bdl->bdln', delta, B, u)
        else:                                                           # :95-96  "B G N L -> B (G H) N L"
            B = B.repeat_interleave(dim // B.shape[1], dim=1)
            deltaB_u = torch.einsum('bdl,bdnl,bdl->bdln', delta, B, u)
    if is_variable_C and C.dim() == 4:                                  # :97-98
        C = C.repeat_interleave(dim // C.shape[1], dim=1)
    last_state = None                                                   # :99
    for i in range(u.shape[2]):                                         # :111
        x = deltaA[:, :, i] * x + deltaB_u[:, :, i]                     # :112
        if not is_variable_C:                                           # :113-114
            y = torch.einsum('bdn,dn->bd', x, C)
        else:
            if C.dim() == 3:                                            # :116-117
                y = torch.einsum('bdn,bn->bd', x, C[:, :, i])
            else:                                                       # :119
                y = torch.einsum('bdn,bdn->bd', x, C[:, :, :, i])
        if i == u.shape[2] - 1:                                         # :120-121
            last_state = x
        ys.append(y)                                                    # :124
    y = torch.stack(ys, dim=2)                                          # :125  (batch dim L)
    out = y if D is None else y + u * D[:, None]                        # :135  rearrange(D, "d -> d 1")
    if z is not None:                                                   # :136-137
        out = out * F.silu(z)
    out = out.to(dtype=dtype_in)                                        # :138
    return out if not return_last_state else (out, last_state)


# --------------------------------------------------------------------------------------
# ctypes binding of the C restatement (oracle/selective_scan_ref.c -> oracle/_build/…so)
# --------------------------------------------------------------------------------------
_LIB = None
_SO = os.path.join(_HERE, "_build", "liboracle_scan.so")


def build_c_oracle(force=False):
    """gcc the C restatement (called by __graft_entry__.build(); building != using)."""
    src = os.path.join(_HERE, "selective_scan_ref.c")
    if (not force) and os.path.exists(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(src):
        return _SO
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-std=c11",
                           "-ffp-contract=off", "-o", _SO, src, "-lm"])
    return _SO


def _lib():
    global _LIB
    if _LIB is None:
        build_c_oracle()
        _LIB = ctypes.CDLL(_SO)
        i, l, p = ctypes.c_int, ctypes.c_long, ctypes.c_void_p
        _LIB.oracle_scan_fwd.argtypes = [p, p, p, p, p, p, p, p, p, i, i, i, i, i,
                                         l, l, l, l, l, l, i, i, i]
        _LIB.oracle_scan_fwd.restype = i
        _LIB.oracle_scan_bwd.argtypes = [p, p, p, p, p, p, p, p,      # u delta A B C D bias dout
                                         p, p, p, p, p, p, p,         # du ddelta dA dB dC dD dbias
                                         i, i, i, i, i, l, l, l, l, l, l, i, i]
        _LIB.oracle_scan_bwd.restype = i
        _LIB.oracle_set_threads.argtypes = [i]
    return _LIB


def _np32(t):
    return np.ascontiguousarray(t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else t,
                                dtype=np.float32)


def _strided(t):
    """(array, strides-in-elements) for a 4-D B/C operand whose last dim is unit-stride."""
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    assert a.dtype == np.float32 and a.ndim == 4
    if a.strides[-1] != 4:
        a = np.ascontiguousarray(a)
    return a, [s // 4 for s in a.strides[:3]]


def c_scan_fwd(u, delta, A, B, C, D, delta_bias, delta_softplus=True, f64=False, threads=0,
               chunk_states=None):
    """C oracle forward. f64=False: fp32 arithmetic in the reference's operation order;
    f64=True: same algorithm in double (the arbiter).  Returns np.ndarray (batch, dim, L)
    of float32 (f64=False) or float64.  chunk_states (optional int T): also return the state
    after every T steps, shape (batch, dim, ceil(L/T), N)."""
    lib = _lib()
    u_, d_, A_ = _np32(u), _np32(delta), _np32(A)
    B_, sB = _strided(B)
    C_, sC = _strided(C)
    Dp = _np32(D) if D is not None else None
    bp = _np32(delta_bias) if delta_bias is not None else None
    batch, dim, L = u_.shape
    N, G = A_.shape[1], B_.shape[1]
    out = np.empty((batch, dim, L), dtype=np.float64 if f64 else np.float32)
    T = int(chunk_states or 0)
    xs = np.zeros((batch, dim, (L + T - 1) // T, N), dtype=out.dtype) if T else None
    lib.oracle_set_threads(int(threads))
    rc = lib.oracle_scan_fwd(u_.ctypes.data, d_.ctypes.data, A_.ctypes.data, B_.ctypes.data,
                             C_.ctypes.data, Dp.ctypes.data if Dp is not None else None,
                             bp.ctypes.data if bp is not None else None, out.ctypes.data,
                             xs.ctypes.data if T else None,
                             batch, dim, L, N, G, sB[0], sB[1], sB[2], sC[0], sC[1], sC[2],
                             int(bool(delta_softplus)), int(bool(f64)), T)
    if rc != 0:
        raise RuntimeError(f"oracle_scan_fwd rc={rc}")
    return (out, xs) if T else out


def c_scan_bwd(u, delta, A, B, C, D, delta_bias, dout, delta_softplus=True, threads=0):
    """C oracle backward (analytic adjoint of the loop above, accumulated in double).
    Returns dict of float64 arrays: du, ddelta, dA, dB, dC, dD, ddelta_bias."""
    lib = _lib()
    u_, d_, A_, g_ = _np32(u), _np32(delta), _np32(A), _np32(dout)
    B_, sB = _strided(B)
    C_, sC = _strided(C)
    Dp = _np32(D) if D is not None else None
    bp = _np32(delta_bias) if delta_bias is not None else None
    batch, dim, L = u_.shape
    N, G = A_.shape[1], B_.shape[1]
    f8 = np.float64
    r = dict(du=np.zeros((batch, dim, L), f8), ddelta=np.zeros((batch, dim, L), f8),
             dA=np.zeros((dim, N), f8), dB=np.zeros((batch, G, N, L), f8),
             dC=np.zeros((batch, G, N, L), f8), dD=np.zeros((dim,), f8),
             ddelta_bias=np.zeros((dim,), f8))
    lib.oracle_set_threads(int(threads))
    rc = lib.oracle_scan_bwd(u_.ctypes.data, d_.ctypes.data, A_.ctypes.data, B_.ctypes.data,
                             C_.ctypes.data, Dp.ctypes.data if Dp is not None else None,
                             bp.ctypes.data if bp is not None else None, g_.ctypes.data,
                             r["du"].ctypes.data, r["ddelta"].ctypes.data, r["dA"].ctypes.data,
                             r["dB"].ctypes.data, r["dC"].ctypes.data, r["dD"].ctypes.data,
                             r["ddelta_bias"].ctypes.data,
                             batch, dim, L, N, G, sB[0], sB[1], sB[2], sC[0], sC[1], sC[2],
                             int(bool(delta_softplus)), 0)
    if rc != 0:
        raise RuntimeError(f"oracle_scan_bwd rc={rc}")
    return r


class CScanFn(torch.autograd.Function):
    """torch.autograd wrapper over the C oracle — used ONLY by bench.py's cpu_baseline leg and by
    tests that need a fast CPU scan inside the torch restatement of the model."""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D, delta_bias, delta_softplus):
        ctx.save_for_backward(u, delta, A, B, C, D, delta_bias)
        ctx.sp = delta_softplus
        return torch.from_numpy(c_scan_fwd(u, delta, A, B, C, D, delta_bias, delta_softplus))

    @staticmethod
    def backward(ctx, dout):
        u, delta, A, B, C, D, bias = ctx.saved_tensors
        r = c_scan_bwd(u, delta, A, B, C, D, bias, dout.contiguous(), ctx.sp)
        f = lambda k: torch.from_numpy(r[k].astype(np.float32))
        return (f("du"), f("ddelta"), f("dA"), f("dB"), f("dC"), f("dD"), f("ddelta_bias"), None)


def c_selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None,
                        delta_softplus=False, return_last_state=False):
    assert z is None and not return_last_state
    return CScanFn.apply(u, delta, A, B, C, D, delta_bias, delta_softplus)


def c_cross_scan_fn(u2, delta, A, B, C, D, delta_bias):
    """Test double for medmamba_amd.cross_scan_fn built on the oracle: materialises what the kernel avoids.
    Direction g reads image-order block g // 2 and is time-reversed for odd g (the flips of MedMamba.py:257),
    runs the oracle scan on explicitly time-ordered tensors, un-flips (MedMamba.py:282) and sums each pair."""
    bsz, d2, L = u2.shape
    Dn = d2 // 2
    tm = lambda t, g: t.flip(-1) if g % 2 else t
    u2v, dv = u2.view(bsz, 2, Dn, L), delta.view(bsz, 4, Dn, L)
    u4 = torch.stack([tm(u2v[:, g // 2], g) for g in range(4)], 1).reshape(bsz, 4 * Dn, L)
    d4 = torch.stack([tm(dv[:, g], g) for g in range(4)], 1).reshape(bsz, 4 * Dn, L)
    B4 = torch.stack([tm(B[:, g], g) for g in range(4)], 1)
    C4 = torch.stack([tm(C[:, g], g) for g in range(4)], 1)
    o4 = CScanFn.apply(u4, d4, A, B4, C4, D, delta_bias, True).view(bsz, 4, Dn, L)
    return torch.stack([o4[:, 0] + o4[:, 1].flip(-1), o4[:, 2] + o4[:, 3].flip(-1)], 1).view(bsz, 2 * Dn, L)
