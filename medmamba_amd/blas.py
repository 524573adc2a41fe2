"""fp32 GEMMs of the projections, issued through the library's mm_gemm_f32 (rocBLAS of this process, called directly).

Why: a GEMM through `torch.bmm` / `torch.mm` costs 17-31 us of host time per call on this image (dispatcher + TunableOp's
string-keyed lookup + hipBLASLt / rocBLAS front end), the same rocBLAS kernel issued through `rocblas_gemm_strided_batched_ex`
costs 7.6 us (`tools/gemm_host_cost.py`).  A MedMamba-S training step has ~220 GEMMs and is within 10 % of being bound by the
host's launch rate, so the projections inside this package's autograd Functions go the short way.

Which kernel runs is unchanged: `medmamba_amd/tuning/gemm_gfx950.csv` (PyTorch TunableOp's record of the fastest solution per
GEMM shape) is read here too, and a GEMM takes the direct route ONLY when the table names a rocBLAS solution for exactly its
(transposes, m, n, k, batch, leading dimensions) and the table was recorded with the rocBLAS build this process runs.  Everything
else — shapes that are not in the table, shapes whose winner is a hipBLASLt solution, a library mismatch, `MM_DIRECT_GEMM=0` —
returns False and the caller issues the same product through torch as before.
"""
import csv
import ctypes
import os

import torch

from . import _lib

_ENABLED = os.environ.get("MM_DIRECT_GEMM", "1") != "0"
# MM_DIRECT_GEMM=all (tests: any_shape(True)): shapes without a record go direct too, with rocBLAS's own choice of kernel
_ANY_SHAPE = os.environ.get("MM_DIRECT_GEMM", "1") == "all"
_TABLE = {}             # (batched?, "nn_m_n_k[_B_b]_ld_lda_ldb_ldc") -> rocBLAS solution index
_STATE = {"attached": None, "file": None}
STATS = {"direct": 0, "torch": 0}


def _rocblas_path():
    return os.path.join(os.path.dirname(torch.__file__), "lib", "librocblas.so")


def _rocblas_version(path):
    try:
        rb = ctypes.CDLL(path)
        n = ctypes.c_size_t()
        if rb.rocblas_get_version_string_size(ctypes.byref(n)) != 0:
            return None
        buf = ctypes.create_string_buffer(n.value)
        if rb.rocblas_get_version_string(buf, n) != 0:
            return None
        return buf.value.decode()
    except (OSError, AttributeError):
        return None


def attach():
    """Hand the process's rocBLAS to the library (once).  False when there is none to hand over."""
    if _STATE["attached"] is None:
        rb = _rocblas_path()
        _STATE["attached"] = bool(_ENABLED and torch.cuda.is_available() and os.path.exists(rb)
                                  and _lib.lib().mm_blas_attach(rb.encode()) == 0)
        if _STATE["attached"] and torch.are_deterministic_algorithms_enabled():
            _lib.lib().mm_blas_set_atomics(0)
    return _STATE["attached"]


def any_shape(on):
    """Tests: route every GEMM of the package's call sites through mm_gemm_f32 (unrecorded shapes with solution 0)."""
    global _ANY_SHAPE
    _ANY_SHAPE = bool(on) and attach()
    return _ANY_SHAPE


def load_table(path):
    """Read TunableOp's result file; keep the rocBLAS winners.  Returns the number of usable entries (0: nothing goes direct)."""
    clear()
    if not _ENABLED or not torch.cuda.is_available() or not os.path.exists(path):
        return 0
    rb = _rocblas_path()
    if not os.path.exists(rb):
        return 0
    rows = list(csv.reader(open(path)))
    recorded = {r[1]: r[2] for r in rows if len(r) >= 3 and r[0] == "Validator"}
    have = _rocblas_version(rb)
    # solution indices belong to one rocBLAS build: TunableOp records rocblas_get_version_string(), compare with the loaded library
    if have is None or recorded.get("ROCBLAS_VERSION") != have:
        return 0
    # ... and to one GPU architecture within that build (torch's own TunableOp validator compares GCN_ARCH_NAME the same way)
    try:
        arch = torch.cuda.get_device_properties(torch.cuda.current_device()).gcnArchName
    except (AttributeError, RuntimeError):
        arch = None
    if recorded.get("GCN_ARCH_NAME") is not None and arch is not None and recorded["GCN_ARCH_NAME"] != arch:
        return 0
    if not attach():
        return 0
    for r in rows:
        if len(r) < 3 or not r[2].startswith("Gemm_Rocblas_"):
            continue
        if r[0].startswith("GemmStridedBatchedTunableOp_float_"):
            _TABLE[(True, r[1])] = int(r[2][len("Gemm_Rocblas_"):])
        elif r[0].startswith("GemmTunableOp_float_"):
            _TABLE[(False, r[1])] = int(r[2][len("Gemm_Rocblas_"):])
    _STATE["file"] = path
    from . import _host
    if _host.module() is not None:          # the C++ sequencing layer issues its GEMMs the same way (csrc_host gemm_out)
        _host.module().set_gemm_table([(("B" if batched else "N") + key, sol) for (batched, key), sol in _TABLE.items()])
        # rocBLAS-only winners (tools/tune_param_gemms.py) for the parameter half on its own stream (ops.py, MM_PARAM_STREAM): same
        # validators, read only next to the default table
        rb_path = os.path.join(os.path.dirname(path), "gemm_gfx950_rocblas.csv")
        rb = []
        if os.path.exists(rb_path):
            rrows = list(csv.reader(open(rb_path)))
            rrec = {r[1]: r[2] for r in rrows if len(r) >= 3 and r[0] == "Validator"}
            if rrec.get("ROCBLAS_VERSION") == have and (arch is None or rrec.get("GCN_ARCH_NAME") in (None, arch)):
                for r in rrows:
                    if len(r) >= 3 and r[2].startswith("Gemm_Rocblas_") and r[0].startswith(("GemmStridedBatchedTunableOp_float_", "GemmTunableOp_float_")):
                        rb.append((("B" if r[0].startswith("GemmStridedBatched") else "N") + r[1], int(r[2][len("Gemm_Rocblas_"):])))
        _host.module().set_gemm_table_rb(rb)
    return len(_TABLE)


def clear():
    """Forget the table here and in the C++ sequencing layer: every GEMM goes through torch again."""
    _TABLE.clear()
    _STATE["file"] = None
    from . import _host
    if _host.module() is not None:
        _host.module().set_gemm_table([])
        _host.module().set_gemm_table_rb([])
    import sys
    ops = sys.modules.get(__package__ + ".ops")
    if ops is not None:
        ops._PARAM_COVERED.clear()        # (MM_PARAM_STREAM: which blocks' weight-gradient GEMMs have rocBLAS records)


def _operand(t):
    """Row-major matrix view (last two dims) -> (op letter of the column-major operand it is in memory, leading dimension)."""
    s0, s1 = t.stride(-2), t.stride(-1)
    if s1 == 1 and s0 >= t.shape[-1]:
        return "n", s0
    if s0 == 1 and s1 >= t.shape[-2]:
        return "t", s1
    return None, 0


def gemm(out, a, b, beta=0.0):
    """out = a @ b (+ beta * out) on the direct route; a (.., m, k), b (.., k, n), out (.., m, n), each 2-D or 3-D (a 2-D operand of a
    3-D product is shared by the whole batch).  Returns False — and does nothing — when this product has no recorded rocBLAS
    solution; the caller then runs it through torch."""
    if not _TABLE and not _ANY_SHAPE:
        return False
    m, k = a.shape[-2], a.shape[-1]
    n = b.shape[-1]
    if out.stride(-1) != 1:
        return False
    opa, lda = _operand(b)          # column-major: out^T = b^T a^T, so rocBLAS's A is b and its B is a
    opb, ldb = _operand(a)
    if opa is None or opb is None:
        return False
    ldc = out.stride(-2)
    if out.dim() == 3:
        batch = out.shape[0]
        key = (True, f"{opa}{opb}_{n}_{m}_{k}_B_{batch}_ld_{lda}_{ldb}_{ldc}")
        sa = b.stride(0) if b.dim() == 3 else 0
        sb = a.stride(0) if a.dim() == 3 else 0
        sc = out.stride(0)
    else:
        batch, sa, sb, sc = 1, 0, 0, 0
        key = (False, f"{opa}{opb}_{n}_{m}_{k}_ld_{lda}_{ldb}_{ldc}")
    sol = _TABLE.get(key)
    if sol is None:
        if not _ANY_SHAPE:
            STATS["torch"] += 1
            return False
        sol = 0
    # handle and stream are those of the CURRENT device inside mm_gemm_f32: make the tensors' device current (like every other launch)
    with _lib.device_guard(out.device):
        rc = _lib.lib().mm_gemm_f32(opa.upper().encode(), opb.upper().encode(), n, m, k, 1.0, b.data_ptr(), lda, sa, a.data_ptr(), ldb,
                                    sb, beta, out.data_ptr(), ldc, sc, batch, sol, _lib.raw_stream())
    if rc == -3:          # MM_ERR_UNSUPPORTED: first call on a stream that is being captured into a hipGraph — the caller uses torch
        return False
    if rc == -6 and sol != 0 and _lib.lib().mm_blas_attached():
        # MM_ERR_BLAS for a RECORDED solution (e.g. an index this GPU / build does not know: rocblas_status_invalid_value):
        # forget the record (the C++ layer does the same for its own copy) and let torch run this product from now on
        _TABLE.pop(key, None)
        STATS["torch"] += 1
        return False
    if rc != 0:
        raise _lib.MedMambaHipError(f"mm_gemm_f32 failed: status {rc}, rocBLAS status {_lib.lib().mm_blas_last_status()} for {key[1]}")
    STATS["direct"] += 1
    return True


def mm_out(out, a, b):
    """out = a @ b: direct route if recorded, else torch.mm / torch.bmm / torch.matmul with out=."""
    if not gemm(out, a, b):
        if a.dim() == b.dim() == 2:
            torch.mm(a, b, out=out)
        elif a.dim() == b.dim() == 3:
            torch.bmm(a, b, out=out)
        else:
            torch.matmul(a, b, out=out)
    return out
