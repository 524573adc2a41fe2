"""Measured GEMM kernel selection for the projections around the scan (PyTorch TunableOp on ROCm).

hipBLASLt's default heuristics pick poor kernels for several of the skinny fp32 GEMMs of SS2D (e.g. the dt projection
`(96 x 3) @ (3 x 3136)` batched 256 times: 0.167 ms by heuristic, 0.056 ms with the rocBLAS solution TunableOp finds).
`gemm_gfx950.csv` holds the winners for the MedMamba-T/S shapes at 64 x 224^2 and the MedMamba-B shapes at 32 x 384^2 per GPU, measured on an MI355X with this
image's ROCm 7.2 / hipBLASLt / rocBLAS builds (`tools/tune_gemms.py` regenerates it).  TunableOp validates the library
versions and the GPU architecture recorded in the file and ignores it on any mismatch; shapes that are not in the file
run with the default heuristic — nothing is tuned at run time unless `tune=True`.
`gemm_gfx950_rocblas.csv` (read by `blas.load_table` next to the default file) holds rocBLAS-ONLY winners for the weight-gradient GEMMs
of the channel-major SS2D blocks whose overall winner is a hipBLASLt kernel: what `MM_PARAM_STREAM=1` runs on its third stream, where
hipBLASLt kernels stop the GPU (`tools/tune_param_gemms.py`, `tools/tune_param_gemms_rocblas.py`).
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


# ---- MIOpen's solver picks for the conv branch's dense convolutions ------------------------------------------------------------
# MIOpen keeps what its solver search found in a per-user "find database" (plain text, one line per convolution problem, named
# after the GPU and the MIOpen build) and looks a problem up there before it measures anything.  Its search times every solver
# once, so near ties (Winograd vs implicit GEMM at the 14x14 and 7x7 stages) fall differently from box to box: 27.8 vs 28.7 ms per
# step of S / 64, and ~70 s of searching per process.  `miopen_gfx950/` is the database of a search on an MI355X
# with this image's MIOpen build for the shapes of BASELINE's single-GPU configurations — the conv-side counterpart of
# gemm_gfx950.csv.  MIOpen ignores files recorded by another build (the build id is part of the file name) and searches as before.
MIOPEN_DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_gfx950")


def seed_miopen_db(dst):
    """Copy the recorded find / perf database files into the directory `dst` (a MIOPEN_USER_DB_PATH) unless a file of that name is
    already there.  Safe when several ranks seed one directory at once (each file arrives by an atomic rename).  Returns the
    number of files written.  Call before the first convolution of the process (MIOpen reads the directory once)."""
    if not os.path.isdir(MIOPEN_DB_DIR):
        return 0
    os.makedirs(dst, exist_ok=True)
    n = 0
    for f in sorted(os.listdir(MIOPEN_DB_DIR)):
        if not f.endswith(".txt"):
            continue
        out = os.path.join(dst, f)
        if os.path.exists(out):
            continue
        import tempfile
        fd, tmp = tempfile.mkstemp(dir=dst, suffix=".tmp")      # a name of its own per caller (process or thread)
        with open(os.path.join(MIOPEN_DB_DIR, f), "rb") as src, os.fdopen(fd, "wb") as o:
            o.write(src.read())
        os.chmod(tmp, 0o644)
        os.replace(tmp, out)
        n += 1
    return n
