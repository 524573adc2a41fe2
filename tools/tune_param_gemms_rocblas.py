#!/usr/bin/env python3
"""Best EXPLICIT rocBLAS solution for the weight-gradient GEMMs of the channel-major SS2D blocks for which TunableOp (hipBLASLt off)
answers "Default" — which may be a hipBLASLt kernel behind rocBLAS's back: every solution index of the Tensile library of the
layout is tried through mm_gemm_f32 (invalid ones are refused by the library), checked against torch, timed.  Appends
`Gemm_Rocblas_<index>` lines to medmamba_amd/tuning/gemm_gfx950_rocblas.csv.  usage: python tools/tune_param_gemms_rocblas.py [csv]"""
import glob, os, re, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medmamba_amd import _lib, blas
import msgpack
out_csv = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "medmamba_amd", "tuning", "gemm_gfx950_rocblas.csv")
assert blas.attach()
lib = _lib.lib()
dev = torch.device("cuda:0")
libdir = os.path.join(os.path.dirname(torch.__file__), "lib", "rocblas", "library")
cands = []
for f in glob.glob(os.path.join(libdir, "TensileLibrary_Type_SS_Contraction_l_*_gfx950.dat")):
    for s in msgpack.unpack(open(f, "rb"), raw=False, strict_map_key=False)["solutions"]:
        cands.append(s["index"])
cands = sorted(set(cands))
print(len(cands), "candidate solutions", flush=True)
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)


def problems():
    for (Bsz, L, dm, D, R) in ((64, 196, 192, 384, 12),):
        Q, C = Bsz * L, R + 32
        yield "d(out_proj.weight)", r(dm, Q), r(D, Q).t(), torch.empty(dm, D, device=dev)
        yield "d(dt_projs_weight)", r(4, D, Q), r(4, C, Q).narrow(1, 0, R).transpose(1, 2), torch.empty(4, D, R, device=dev)
        yield "d(x_proj_weight)", r(2, 2 * C, Q), r(2, D, Q).transpose(1, 2), torch.empty(2, 2 * C, D, device=dev)


def run(a, b, out, sol):
    m, k, n = a.shape[-2], a.shape[-1], b.shape[-1]
    opa, lda = blas._operand(b); opb, ldb = blas._operand(a)
    batch = out.shape[0] if out.dim() == 3 else 1
    sa = b.stride(0) if b.dim() == 3 else 0; sb = a.stride(0) if a.dim() == 3 else 0; sc = out.stride(0) if out.dim() == 3 else 0
    key = f"{opa}{opb}_{n}_{m}_{k}" + (f"_B_{batch}" if out.dim() == 3 else "") + f"_ld_{lda}_{ldb}_{out.stride(-2)}"
    rc = lib.mm_gemm_f32(opa.upper().encode(), opb.upper().encode(), n, m, k, 1.0, b.data_ptr(), lda, sa, a.data_ptr(), ldb, sb, 0.0,
                         out.data_ptr(), out.stride(-2), sc, batch, sol, _lib.raw_stream())
    return rc, key


lines = []
for name, a, b, out in problems():
    ref = torch.matmul(a.double(), b.double()).float()
    best = (1e9, None)
    nvalid = 0
    for idx in cands:
        sol = -idx
        out.zero_()
        rc, key = run(a, b, out, sol)
        if rc != 0:
            continue
        torch.cuda.synchronize()
        if not torch.allclose(out, ref, rtol=2e-3, atol=2e-2):
            continue
        nvalid += 1
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            run(a, b, out, sol)
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) / 5
        if t < best[0]:
            best = (t, sol)
    print(f"{name}: {key}: {nvalid} valid solutions, best {best[1]} at {best[0] * 1e3:.1f} us", flush=True)
    kind = "GemmStridedBatchedTunableOp_float_" if out.dim() == 3 else "GemmTunableOp_float_"
    lines.append(f"{kind}{key[:2].upper()},{key},Gemm_Rocblas_{best[1]},{best[0]:.6f}")
rows = [l for l in open(out_csv).read().splitlines() if l.split(",")[1] not in {x.split(",")[1] for x in lines}]
open(out_csv, "w").write("\n".join(rows + lines) + "\n")
print(open(out_csv).read())
