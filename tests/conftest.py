import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")


def load_golden(name):
    """Load a committed fixture (data only; made by tools/gen_golden.py from the reference)."""
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def split_sd(fx, prefix="sd/"):
    import torch
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in fx.items() if k.startswith(prefix)}


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """Build libmedmamba_hip.so (hipcc cross-compiles without a GPU) and the C oracle once per session.  build() is
    incremental on source / header mtimes, so edited kernels are never tested against a stale binary."""
    from medmamba_amd.build import build, build_host
    build()
    build_host()          # the C++ sequencing layer (g++, ~90 s when its source or the header changed, otherwise nothing)
    from oracle.scan_ref import build_c_oracle
    build_c_oracle()


@pytest.fixture(scope="session")
def golden():
    return load_golden
