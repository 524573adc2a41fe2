"""medmamba_amd — MI355X (gfx950) native implementation of MedMamba's SS2D / SS-Conv-SSM hot path.

Host side: Python on PyTorch-ROCm (device memory, streams, autograd, torch.distributed/RCCL).
Device side: hand-written HIP kernels in libmedmamba_hip.so behind a C ABI (include/medmamba_hip.h).
"""
from .selective_scan_interface import selective_scan_fn, cross_scan_fn, SelectiveScanFn, CrossScanFn  # noqa: F401

__all__ = ["selective_scan_fn", "cross_scan_fn", "SelectiveScanFn", "CrossScanFn"]
