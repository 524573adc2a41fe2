"""ctypes binding of libmedmamba_hip.so (C ABI: include/medmamba_hip.h).

There is deliberately NO fallback: if the shared library is missing or a kernel launch fails the
caller gets an exception.  PyTorch is only the owner of device memory and streams here.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "lib", "libmedmamba_hip.so")
# A/B measurements of two builds of the SAME ABI on one box (tools/): MM_HIP_LIB names another .so; never a fallback
if os.environ.get("MM_HIP_LIB"):
    SO_PATH = os.path.abspath(os.environ["MM_HIP_LIB"])

ABI_VERSION = 22        # == MM_ABI_VERSION of include/medmamba_hip.h (checked when the library is loaded)
_f32p = ctypes.c_void_p
_i64 = ctypes.c_int64


class ScanArgs(ctypes.Structure):
    """Mirror of `struct mm_scan_args` (include/medmamba_hip.h); struct_size is filled in on construction."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("batch", ctypes.c_int32), ("dim", ctypes.c_int32), ("L", ctypes.c_int32), ("N", ctypes.c_int32),
        ("G", ctypes.c_int32), ("delta_softplus", ctypes.c_int32),
        ("u", _f32p), ("delta", _f32p), ("A", _f32p), ("B", _f32p), ("C", _f32p), ("D", _f32p),
        ("delta_bias", _f32p), ("out", _f32p), ("x_chk", _f32p),
        ("u_sb", ctypes.c_int64), ("u_sd", ctypes.c_int64),
        ("delta_sb", ctypes.c_int64), ("delta_sd", ctypes.c_int64),
        ("B_sb", ctypes.c_int64), ("B_sg", ctypes.c_int64), ("B_sn", ctypes.c_int64),
        ("C_sb", ctypes.c_int64), ("C_sg", ctypes.c_int64), ("C_sn", ctypes.c_int64),
        ("dout", _f32p), ("du", _f32p), ("ddelta", _f32p), ("dA", _f32p), ("dB", _f32p), ("dC", _f32p),
        ("dD", _f32p), ("ddelta_bias", _f32p),
        ("variant", ctypes.c_int32), ("u_groups", ctypes.c_int32), ("u_map", ctypes.c_uint32),
        ("rev_mask", ctypes.c_uint32),
        ("dB_sb", ctypes.c_int64), ("dB_sg", ctypes.c_int64), ("dB_sn", ctypes.c_int64),
        ("dC_sb", ctypes.c_int64), ("dC_sg", ctypes.c_int64), ("dC_sn", ctypes.c_int64),
        ("dout_sb", ctypes.c_int64), ("dud_sb", ctypes.c_int64), ("o_sd", ctypes.c_int64),
        ("dt_w", _f32p), ("dts", _f32p), ("dts_sb", ctypes.c_int64), ("dts_sg", ctypes.c_int64), ("dts_sn", ctypes.c_int64),
        ("dt_rank", ctypes.c_int32),
        ("dpar_sb", ctypes.c_int64), ("dBC_sc", ctypes.c_int64),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = ctypes.sizeof(ScanArgs)


# every symbol include/medmamba_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "mm_abi_version": (ctypes.c_int, []),
    "mm_scan_chunk": (ctypes.c_int, []),
    "mm_scan_dt_max": (ctypes.c_int, []),
    "mm_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "mm_scan_fwd": (ctypes.c_int, [ctypes.POINTER(ScanArgs), ctypes.c_void_p]),
    "mm_scan_bwd": (ctypes.c_int, [ctypes.POINTER(ScanArgs), ctypes.c_void_p]),
    "mm_scan_plan": (ctypes.c_int, [ctypes.POINTER(ScanArgs), ctypes.c_int, ctypes.POINTER(ctypes.c_int32)]),
    "mm_shuffle_residual_fwd": (ctypes.c_int, [_f32p, _f32p, _i64, _i64, _f32p, _f32p, _f32p, ctypes.c_int, _f32p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "mm_dwconv_silu_cross_fwd": (ctypes.c_int, [_f32p, _i64, _i64, _f32p, _f32p, _f32p, _i64, _i64] + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "mm_dwconv_silu_cross_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "mm_dwconv_silu_cross_strips": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "mm_dwconv_silu_cross_bwd": (ctypes.c_int, [_f32p, _i64, _i64, _f32p, _i64, _i64, _f32p, _i64, _i64, _f32p, _f32p, _f32p, _i64, _i64, _f32p]
                                 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "mm_cross_merge_fwd": (ctypes.c_int, [_f32p, _f32p, _i64, _i64] + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "mm_plane_transpose": (ctypes.c_int, [_f32p, _i64, _i64, _f32p, _i64, _i64] + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "mm_ln_gate_fwd": (ctypes.c_int, [_f32p, _i64, _i64, _f32p, _i64, _i64, _f32p, _f32p, ctypes.c_float, _f32p, _i64, _i64,
                                      _f32p, _f32p] + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "mm_ln_gate_bwd": (ctypes.c_int, [_f32p, _i64, _i64, _f32p, _i64, _i64, _f32p, _i64, _i64, _f32p, _f32p, _f32p, _f32p,
                                      _f32p, _i64, _i64, _f32p, _i64, _i64, _f32p] + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "mm_ln_gate_rows": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mm_block_split_fwd": (ctypes.c_int, [_f32p, _f32p, _f32p, ctypes.c_float, _f32p, _f32p, _f32p, _f32p, _f32p] + [ctypes.c_int] * 3
                           + [ctypes.c_void_p]),
    "mm_block_split_bwd": (ctypes.c_int, [_f32p] * 9 + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "mm_block_split_rows": (ctypes.c_int, [ctypes.c_int] * 3),
    "mm_shuffle_residual_bwd": (ctypes.c_int, [_f32p, _f32p, _f32p, _i64, _i64, _f32p, _f32p, _f32p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "mm_patch_merge_ln_supported": (ctypes.c_int, [ctypes.c_int]),
    "mm_patch_merge_ln_rows": (ctypes.c_int, [ctypes.c_int] * 3),
    "mm_patch_merge_ln_fwd": (ctypes.c_int, [_f32p, _f32p, _f32p, ctypes.c_float, _f32p, _f32p, _f32p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "mm_patch_merge_ln_bwd": (ctypes.c_int, [_f32p] * 7 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "mm_nchw_ln_rows_supported": (ctypes.c_int, [ctypes.c_int]),
    "mm_nchw_ln_rows_ws_rows": (ctypes.c_int, [ctypes.c_int] * 2),
    "mm_nchw_ln_rows_fwd": (ctypes.c_int, [_f32p, _f32p, _f32p, ctypes.c_float, _f32p, _f32p, _f32p] + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "mm_nchw_ln_rows_bwd": (ctypes.c_int, [_f32p] * 7 + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "mm_bn_splits": (ctypes.c_int, [ctypes.c_int] * 3),
    "mm_bn_relu_fwd": (ctypes.c_int, [_f32p, _f32p, _f32p, ctypes.c_float, ctypes.c_float, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p]
                       + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "mm_bn_relu_fwd_stats": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, _f32p, _f32p, ctypes.c_float, ctypes.c_float, _f32p, _f32p, _f32p, _f32p,
                                            _f32p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "mm_bn_fused": (ctypes.c_int, [ctypes.c_int] * 3),
    "mm_bn_relu_bwd": (ctypes.c_int, [_f32p] * 11 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "mm_channel_sum_nchw_split": (ctypes.c_int, [ctypes.c_int] * 2),
    "mm_channel_sum_nchw": (ctypes.c_int, [_f32p, _f32p] + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "mm_sum_lead": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, _i64, _i64, ctypes.c_void_p]),
    "mm_sum_lead_chunks": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, _i64, _i64, _i64, ctypes.c_void_p]),
    "mm_ss2d_pack_size": (ctypes.c_int, [ctypes.c_int] * 4),
    "mm_ss2d_pack_fwd": (ctypes.c_int, [_f32p] * 6 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "mm_ss2d_pack_bwd": (ctypes.c_int, [_f32p] * 4 + [ctypes.c_int] * 5 + [_f32p, ctypes.c_int, _f32p, _f32p, ctypes.c_int, ctypes.c_int,
                                                                              _f32p, ctypes.c_void_p]),
    "mm_ss2d_pack_parts_size": (ctypes.c_int, [ctypes.c_int] * 4),
    "mm_event_record": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "mm_im2col3x3": (ctypes.c_int, [_f32p, _f32p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "mm_adamw_chunk": (ctypes.c_int, []),
    "mm_adamw_max_tensors": (ctypes.c_int, []),
    "mm_adamw_step": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int]
                      + [ctypes.c_float] * 5 + [ctypes.c_double, ctypes.c_void_p]),
    "mm_blas_attach": (ctypes.c_int, [ctypes.c_char_p]),
    "mm_blas_attached": (ctypes.c_int, []),
    "mm_blas_set_atomics": (ctypes.c_int, [ctypes.c_int]),
    "mm_blas_last_status": (ctypes.c_int, []),
    "mm_gemm_f32": (ctypes.c_int, [ctypes.c_char, ctypes.c_char, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, _f32p, ctypes.c_int,
                                   _i64, _f32p, ctypes.c_int, _i64, ctypes.c_float, _f32p, ctypes.c_int, _i64, ctypes.c_int, ctypes.c_int32,
                                   ctypes.c_void_p]),
}

# entry points that only the experiments build exports (include/medmamba_hip.h, #ifdef MM_EXPERIMENTS)
EXP_SYMBOLS = {
    "mm_conv3x3_fwd_tiles": (ctypes.c_int, [ctypes.c_int] * 3),
    "mm_conv3x3_fwd": (ctypes.c_int, [_f32p] * 4 + [ctypes.c_int, _f32p, _f32p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "mm_conv3x3_v2_tiles": (ctypes.c_int, [ctypes.c_int] * 3),
    "mm_conv3x3_v2_fwd": (ctypes.c_int, [_f32p] * 4 + [ctypes.c_int, _f32p, _f32p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
}
EXP_SO_PATH = os.path.join(_HERE, "lib", "libmedmamba_hip_exp.so")

_lib = None
_exp = None


class MedMambaHipError(RuntimeError):
    pass


def lib():
    """Load the library (once). Raises MedMambaHipError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise MedMambaHipError(
                f"{SO_PATH} not found: build it with `python -m medmamba_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        try:
            handle = ctypes.CDLL(SO_PATH)
        except OSError as e:  # pragma: no cover
            raise MedMambaHipError(f"cannot load {SO_PATH}: {e}") from e
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)      # AttributeError if the .so is stale
            fn.restype, fn.argtypes = res, args
        if handle.mm_abi_version() != ABI_VERSION:
            raise MedMambaHipError("libmedmamba_hip.so ABI version mismatch; rebuild")
        _lib = handle
    return _lib


def _bind(path, symbols):
    handle = ctypes.CDLL(path)
    for name, (res, args) in symbols.items():
        fn = getattr(handle, name)      # AttributeError if the .so is stale
        fn.restype, fn.argtypes = res, args
    if handle.mm_abi_version() != ABI_VERSION:
        raise MedMambaHipError(f"{os.path.basename(path)} ABI version mismatch; rebuild")
    return handle


def exp_lib():
    """The EXPERIMENTS build of the library (`python -m medmamba_amd.build --experiments`): everything the product library has plus
    the measured-and-rejected experiments.  Only tests / tools / the opt-in MM_OWN_CONV=1 route call this; None when not built."""
    global _exp
    if _exp is None:
        if not os.path.exists(EXP_SO_PATH):
            return None
        _exp = _bind(EXP_SO_PATH, {**SYMBOLS, **EXP_SYMBOLS})
    return _exp


def check(rc, what):
    if rc != 0:
        msg = lib().mm_status_string(rc).decode()
        raise MedMambaHipError(f"{what} failed: status {rc} ({msg})")


def scan_chunk():
    return lib().mm_scan_chunk()


class device_guard:
    """`with torch.cuda.device(dev)` without its Python overhead: switches the current HIP device only when `dev` is not
    already current (the launch wrappers run once per kernel, ~1400 times per training step)."""
    __slots__ = ("idx", "prev")

    def __init__(self, device):
        self.idx = device.index if device.index is not None else torch._C._cuda_getDevice()

    def __enter__(self):
        self.prev = torch._C._cuda_getDevice()
        if self.prev != self.idx:
            torch._C._cuda_setDevice(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev != self.idx:
            torch._C._cuda_setDevice(self.prev)
        return False


def raw_stream():
    """hipStream_t of torch's current stream on the current device (the stream every kernel of this library is launched on)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
