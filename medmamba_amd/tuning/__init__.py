"""Measured GEMM kernel selection for the projections around the scan (PyTorch TunableOp on ROCm).

hipBLASLt's default heuristics pick poor kernels for several of the skinny fp32 GEMMs of SS2D (e.g. the dt projection
`(96 x 3) @ (3 x 3136)` batched 256 times: 0.167 ms by heuristic, 0.056 ms with the rocBLAS solution TunableOp finds).
`gemm_gfx950.csv` holds the winners for the MedMamba-T/S shapes at 64 x 224^2 and the MedMamba-B shapes at 32 x 384^2 per GPU, measured on an MI355X with this
image's ROCm 7.2 / hipBLASLt / rocBLAS builds (`tools/tune_gemms.py` regenerates it).  TunableOp validates the library
versions and the GPU architecture recorded in the file and ignores it on any mismatch; shapes that are not in the file
run with the default heuristic — nothing is tuned at run time unless `tune=True`.
"""
import os

import torch

DEFAULT_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_gfx950.csv")


def enable_tuned_gemms(path=None, tune=False):
    """Switch torch's GEMM dispatch to the recorded solutions (`tune=True`: also time unseen shapes and write `path` at
    process exit — single process only).  Returns the file in use, or None when there is nothing to load."""
    path = path or DEFAULT_FILE
    if not torch.cuda.is_available() or (not tune and not os.path.exists(path)):
        return None
    t = torch.cuda.tunable
    t.enable(True)
    t.tuning_enable(bool(tune))
    t.set_filename(path, insert_device_ordinal=False)
    from .. import blas
    blas.load_table(path)          # the same record drives the direct rocBLAS route of the package's own GEMM call sites
    return path
