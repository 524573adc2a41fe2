"""CPU: the C-ABI library builds/loads and exports every symbol include/medmamba_hip.h declares;
host-side argument checking of the operator mirrors the reference's (no compute without a GPU)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT
import medmamba_amd
from medmamba_amd import _lib


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "medmamba_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from medmamba_amd.build import build
    so = build()
    h = ctypes.CDLL(so)
    decl = _declared_symbols()
    assert "mm_scan_fwd" in decl and "mm_scan_bwd" in decl
    for name in decl:
        assert hasattr(h, name), name
        assert name in _lib.SYMBOLS, f"{name} declared in the header but not bound in _lib.SYMBOLS"
    assert set(_lib.SYMBOLS) <= set(decl)
    assert _lib.lib().mm_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define MM_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "medmamba_hip.h")).read()).group(1))
    assert _lib.scan_chunk() == 16
    assert b"unsupported" in _lib.lib().mm_status_string(-3)


def test_struct_layout_matches_header(tmp_path):
    """sizeof / offsetof of mm_scan_args as gcc sees include/medmamba_hip.h == the ctypes mirror, field by field."""
    import subprocess
    names = [f[0] for f in _lib.ScanArgs._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "medmamba_hip.h"\nint main(void){printf("%zu", sizeof(mm_scan_args));'
                   + "".join(f'printf(" %zu", offsetof(mm_scan_args, {n}));' for n in names) + 'return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got[0] == ctypes.sizeof(_lib.ScanArgs)
    assert got[1:] == [getattr(_lib.ScanArgs, n).offset for n in names]


def test_bad_arguments_are_rejected_without_launch():
    a = _lib.ScanArgs()
    assert _lib.lib().mm_scan_fwd(None, None) == -1
    a.batch, a.dim, a.L, a.N, a.G = 1, 6, 8, 16, 4        # dim % G != 0
    assert _lib.lib().mm_scan_fwd(a, None) == -2
    a.dim, a.N = 8, 8                                       # N != 16
    assert _lib.lib().mm_scan_fwd(a, None) == -3
    a.N = 16                                                # null operands
    assert _lib.lib().mm_scan_fwd(a, None) == -1
    with pytest.raises(_lib.MedMambaHipError):
        _lib.check(-3, "x")


def test_operator_has_no_cpu_fallback_and_mirrors_reference_errors():
    u = torch.zeros(1, 8, 8); A = torch.zeros(8, 16); B = torch.zeros(1, 4, 16, 8)
    with pytest.raises(RuntimeError, match="HIP device"):
        medmamba_amd.selective_scan_fn(u, u, A, B, B)
    with pytest.raises(NotImplementedError):
        medmamba_amd.selective_scan_fn(u, u, A, B, B, z=u)
    with pytest.raises(NotImplementedError):
        medmamba_amd.selective_scan_fn(u, u, A, B, B, return_last_state=True)
    with pytest.raises(NotImplementedError):
        medmamba_amd.selective_scan_fn(u, u, A, torch.zeros(1, 16, 8), B)


def test_tuned_gemm_table_ships_and_is_inert_without_a_device():
    from medmamba_amd.tuning import DEFAULT_FILE, enable_tuned_gemms
    rows = [l.strip().split(",") for l in open(DEFAULT_FILE) if l.strip()]
    assert any(r[0] == "Validator" and r[1] == "GCN_ARCH_NAME" and r[2].startswith("gfx950") for r in rows)
    assert sum(r[0].startswith("Gemm") for r in rows) >= 40          # the MedMamba-S projection shapes
    import torch
    if not torch.cuda.is_available():
        assert enable_tuned_gemms() is None


def test_graft_entry_build_runs():
    """The driver's build check: compiles the HIP library and the C oracle and imports the package."""
    import __graft_entry__
    __graft_entry__.build()
