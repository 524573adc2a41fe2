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


def _declared_symbols(experiments=False):
    """Entry points the header declares; the `#ifdef MM_EXPERIMENTS` section belongs to the experiments build only."""
    hdr = open(os.path.join(ROOT, "include", "medmamba_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    exp = "".join(re.findall(r"#ifdef MM_EXPERIMENTS(.*?)#endif", hdr, flags=re.S))
    if not experiments:
        hdr = re.sub(r"#ifdef MM_EXPERIMENTS.*?#endif", "", hdr, flags=re.S)
    syms = sorted(set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", hdr)))
    return (syms, sorted(set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", exp)))) if experiments else syms


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
    # the experiments (own dense convolutions, ablation bits) are NOT in the product library; the experiments build has both sets
    all_syms, exp_only = _declared_symbols(experiments=True)
    assert exp_only and set(exp_only) == set(_lib.EXP_SYMBOLS)
    for name in exp_only:
        assert not hasattr(h, name), f"{name} is an experiment and must not ship in libmedmamba_hip.so"
    he = ctypes.CDLL(build(experiments=True))
    for name in all_syms:
        assert hasattr(he, name), name
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


def _integration_snippet():
    """The ctypes binding INTEGRATION.md §3 shows a maintainer, executed verbatim."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = md.split("<!-- abi-snippet:begin")[1].split("<!-- abi-snippet:end -->")[0]
    code = block.split("```python\n", 1)[1].rsplit("```", 1)[0]
    ns = {}
    exec(compile(code, "INTEGRATION.md#abi-snippet", "exec"), ns)
    return ns


def test_integration_md_binding_matches_the_header(tmp_path):
    """The documented struct is the header's struct (size and every offset), and the documented call reaches the library's
    argument checks: struct_size accepted, NULL operands reported — not a read past the caller's buffer."""
    import subprocess
    ns = _integration_snippet()
    S = ns["mm_scan_args"]
    names = [f[0] for f in S._fields_]
    assert names == [f[0] for f in _lib.ScanArgs._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include "medmamba_hip.h"\nint main(void){printf("%zu %u %u %u", sizeof(mm_scan_args), '
                   'MM_SCAN_ARGS_SIZE_BASE, MM_SCAN_ARGS_SIZE_DBC, MM_SCAN_ARGS_SIZE_STRIDED);'
                   + "".join(f'printf(" %zu", offsetof(mm_scan_args, {n}));' for n in names) + 'return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got[0] == ctypes.sizeof(S)
    assert got[4:] == [getattr(S, n).offset for n in names]
    base, dbc, strided = got[1:4]
    v18 = S.dpar_sb.offset
    lib = ns["lib"]
    a = S(struct_size=ctypes.sizeof(S), batch=1, dim=8, L=8, N=16, G=4)
    assert lib.mm_scan_fwd(ctypes.byref(a), None) == -1             # MM_ERR_NULL: sizes fine, operands missing
    # struct sizes: every layout the header ever ended at is accepted, anything else is MM_ERR_SHAPE before any other check
    for sz, want in [(0, -2), (24, -2), (base, -1), (base + 8, -2), (dbc, -1), (strided, -1), (v18, -1), (ctypes.sizeof(S) - 4, -2),
                     (ctypes.sizeof(S), -1)]:
        a.struct_size = sz
        assert lib.mm_scan_fwd(ctypes.byref(a), None) == want, sz
    # a short (older) caller: the library must not look at the bytes behind struct_size — poison them
    raw = (ctypes.c_ubyte * ctypes.sizeof(S))()
    ctypes.memmove(raw, ctypes.byref(a), ctypes.sizeof(S))
    for i in range(strided, ctypes.sizeof(S)):
        raw[i] = 0xFF                                               # dt_w / dts / dt_rank garbage
    b = S.from_buffer(raw)
    b.struct_size = strided
    assert lib.mm_scan_fwd(ctypes.byref(b), None) == -1            # still "operands missing", not "unsupported dt projection"
    # a longer (newer) caller: fine while the unknown tail is zero, refused when it is used
    class Longer(ctypes.Structure):
        _fields_ = [("base", S), ("future", ctypes.c_int64)]
    c = Longer()
    ctypes.memmove(ctypes.byref(c), ctypes.byref(a), ctypes.sizeof(S))
    c.base.struct_size = ctypes.sizeof(Longer)
    fn = lib.mm_scan_fwd
    old = fn.argtypes
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    try:
        assert fn(ctypes.addressof(c), None) == -1
        c.future = 1
        assert fn(ctypes.addressof(c), None) == -3
    finally:
        fn.argtypes = old


@pytest.mark.gpu
def test_integration_md_binding_runs_a_forward():
    """One small forward through exactly the binding INTEGRATION.md shows, checked against the CPU oracle."""
    import numpy as np
    from oracle.scan_ref import c_scan_fwd
    ns = _integration_snippet()
    g = torch.Generator().manual_seed(3)
    b, G, H, L, N, R = 2, 4, 12, 68, 16, 2
    u, delta = torch.randn(b, G * H, L, generator=g), torch.randn(b, G * H, L, generator=g)
    A = -torch.exp(torch.randn(G * H, N, generator=g) * 0.5)
    xd = torch.randn(b, G, R + 2 * N, L, generator=g)
    D, bias = torch.randn(G * H, generator=g), torch.randn(G * H, generator=g) - 3
    dev = torch.device("cuda:0")
    xdd = xd.to(dev)
    out = ns["selective_scan_fwd"](u.to(dev), delta.to(dev), A.to(dev), xdd[:, :, R:R + N], xdd[:, :, R + N:], D.to(dev), bias.to(dev))
    torch.cuda.synchronize()
    ref = c_scan_fwd(u, delta, A, xd[:, :, R:R + N], xd[:, :, R + N:], D, bias, True, f64=True)
    assert np.abs(out.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max()


def test_bad_arguments_are_rejected_without_launch():
    a = _lib.ScanArgs()
    assert a.struct_size == ctypes.sizeof(_lib.ScanArgs)
    assert _lib.lib().mm_scan_fwd(None, None) == -1
    a.batch, a.dim, a.L, a.N, a.G = 1, 6, 8, 16, 4        # dim % G != 0
    assert _lib.lib().mm_scan_fwd(a, None) == -2
    a.dim, a.N = 8, 8                                       # N != 16
    assert _lib.lib().mm_scan_fwd(a, None) == -3
    a.N = 16                                                # null operands
    assert _lib.lib().mm_scan_fwd(a, None) == -1
    with pytest.raises(_lib.MedMambaHipError):
        _lib.check(-3, "x")


def test_gemm_entry_refuses_without_an_attached_blas():
    """mm_gemm_f32 links no BLAS of its own: until mm_blas_attach has been given the host process's rocBLAS it returns MM_ERR_BLAS
    (nothing is launched), and the Python route (medmamba_amd.blas.gemm) declines, so callers keep their torch GEMM."""
    from medmamba_amd import blas
    lib = _lib.lib()
    assert lib.mm_blas_attach(None) == -1
    assert lib.mm_blas_attach(b"/nonexistent/librocblas.so") == -6
    if not lib.mm_blas_attached():
        assert lib.mm_gemm_f32(b"N", b"N", 4, 4, 4, 1.0, 64, 4, 0, 64, 4, 0, 0.0, 64, 4, 0, 1, 0, None) == -6
    assert lib.mm_status_string(-6).decode().startswith("BLAS")
    if not torch.cuda.is_available():
        assert blas.load_table(os.path.join(os.path.dirname(blas.__file__), "tuning", "gemm_gfx950.csv")) == 0
        a = torch.zeros(4, 4)
        assert blas.gemm(torch.zeros(4, 4), a, a) is False
    assert blas._operand(torch.zeros(6, 5)) == ("n", 5) and blas._operand(torch.zeros(5, 6).t()) == ("t", 6)
    assert blas._operand(torch.zeros(6, 10)[:, ::2]) == (None, 0)


def test_host_side_helpers_validate_their_arguments_without_launching():
    """mm_adamw_step / mm_event_record (ABI 20): NULL tables, empty or oversized tensor groups and a step count below 1 are refused
    before anything touches the device."""
    lib = _lib.lib()
    assert lib.mm_adamw_chunk() >= 256 and lib.mm_adamw_max_tensors() * 8 < 4096       # the gradient pointers fit the kernel arguments
    ok = 64                                                                              # any non-NULL value for a table pointer
    arr = (ctypes.c_void_p * 2)(ok, ok)
    args = lambda **kw: [kw.get("P", ok), kw.get("G", arr), kw.get("t0", 0), kw.get("nt", 2), ok, ok, ok, ok, ok, kw.get("nchunks", 2),
                         1e-3, 0.9, 0.999, 1e-8, 1e-2, kw.get("step", 1.0), None]
    assert lib.mm_adamw_step(*args(P=None)) == -1
    assert lib.mm_adamw_step(*args(G=None)) == -1
    assert lib.mm_adamw_step(*args(nchunks=0)) == -2
    assert lib.mm_adamw_step(*args(nt=0)) == -2
    assert lib.mm_adamw_step(*args(nt=lib.mm_adamw_max_tensors() + 1)) == -2
    assert lib.mm_adamw_step(*args(step=0.0)) == -2
    assert lib.mm_adamw_step(*args(G=(ctypes.c_void_p * 2)(ok, None))) == -1             # a NULL gradient pointer in the group
    assert lib.mm_event_record(None, None) == -1


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
