"""GPU: the direct rocBLAS route of the projections (medmamba_amd.blas / mm_gemm_f32) against torch's own GEMMs.

blas.any_shape(True) sends every GEMM of the package's call sites through mm_gemm_f32 (unrecorded shapes with rocBLAS's default
solution), so the small shapes used here exercise the same operand descriptions (transposes, leading dimensions, broadcast
operands, strided outputs) as the recorded full-size ones."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def direct():
    from medmamba_amd import blas
    assert blas.any_shape(True), "rocBLAS of this process could not be attached"
    blas.STATS["direct"] = 0
    yield blas
    blas.any_shape(False)


def test_operand_forms_match_torch(direct):
    blas = direct
    g = torch.Generator(device=DEV).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
    m, n, k, B = 37, 52, 19, 5
    cases = []
    for ta in (False, True):
        for tb in (False, True):
            a2 = rnd(k, m).t() if ta else rnd(m, k)
            b2 = rnd(n, k).t() if tb else rnd(k, n)
            a3 = rnd(B, k, m).transpose(1, 2) if ta else rnd(B, m, k)
            b3 = rnd(B, n, k).transpose(1, 2) if tb else rnd(B, k, n)
            cases += [(a2, b2), (a3, b3), (a2, b3), (a3, b2)]
    for a, b in cases:
        want = torch.matmul(a, b)
        out = torch.full_like(want, float("nan"))
        assert blas.gemm(out, a, b)
        assert torch.allclose(out, want, rtol=1e-4, atol=1e-4), (a.shape, a.stride(), b.shape, b.stride())
    # strided output rows (a row block of a wider buffer), operands that are row blocks themselves, beta = 1
    big = rnd(4, 10, 64)
    a, b = rnd(4, 3, 7), rnd(4, 7, 64)
    ref = big.clone()
    ref[:, 2:5] = torch.bmm(a, b)
    assert blas.gemm(big[:, 2:5], a, b)
    assert torch.allclose(big, ref, rtol=1e-4, atol=1e-4)
    acc = rnd(4, 3, 64)
    ref = acc + torch.bmm(a, big[:, :7])
    assert blas.gemm(acc, a, big[:, :7], beta=1.0)
    assert torch.allclose(acc, ref, rtol=1e-4, atol=1e-4)
    # a layout rocBLAS cannot take (no unit stride in either matrix dimension) is declined, nothing is written
    out = torch.zeros(m, n, device=DEV)
    assert not blas.gemm(out, rnd(m, 2 * k)[:, ::2], rnd(k, n))
    assert float(out.abs().max()) == 0.0
    assert blas.STATS["direct"] >= len(cases) + 2


@pytest.mark.parametrize("layout", ["bm", "cm"])
def test_tiny_model_direct_route_matches_torch_route(layout, monkeypatch, capsys):
    from medmamba_amd import blas, modules, ops
    # the Python route of the projections is what calls blas.gemm (the C++-sequenced branches issue their GEMMs through ATen)
    monkeypatch.setattr(ops, "ss2d_branch_native_ok", lambda *a: False)
    monkeypatch.setattr(ops, "conv_branch_native", lambda *a: None)
    torch.manual_seed(5)
    net = modules.VSSM(num_classes=4, depths=[1, 1, 1, 1], dims=[16, 32, 64, 128], drop_path_rate=0.0).to(DEV).train()
    x = torch.randn(3, 3, 64, 64, device=DEV)
    y = torch.randint(0, 4, (3,), device=DEV)
    monkeypatch.setattr(ops, "_LAYOUT", layout)

    def run():
        net.zero_grad(set_to_none=True)
        logits = net(x)
        torch.nn.functional.cross_entropy(logits, y).backward()
        return logits.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}

    l0, g0 = run()
    assert blas.any_shape(True)
    blas.STATS["direct"] = 0
    try:
        l1, g1 = run()
    finally:
        blas.any_shape(False)
    assert blas.STATS["direct"] >= 4 * (14 if layout == "cm" else 8), blas.STATS     # GEMMs per block that went the direct way
    assert (l1 - l0).abs().max().item() <= 1e-4 * max(1.0, l0.abs().max().item())
    for k in g0:
        scale = max(1e-4, g0[k].abs().max().item())
        assert (g1[k] - g0[k]).abs().max().item() <= 2e-3 * scale, (layout, k)


def test_recorded_solutions_are_used_at_full_size(monkeypatch):
    """The shapes of MedMamba-S at 64 x 224^2 are in the table with the rocBLAS build of this image: one 14x14 block (channel-major
    planes) sends its projections through the recorded solutions and agrees with the torch route."""
    from medmamba_amd import blas, modules, ops
    from medmamba_amd.tuning import enable_tuned_gemms
    monkeypatch.setattr(ops, "ss2d_branch_native_ok", lambda *a: False)
    monkeypatch.setattr(ops, "conv_branch_native", lambda *a: None)
    enable_tuned_gemms()
    if not blas._TABLE:
        pytest.skip("GEMM table not recorded for this rocBLAS build")
    try:
        torch.backends.cudnn.deterministic = True      # the reference's mode (train.py:28-29): bitwise comparisons are meaningful
        torch.manual_seed(0)
        blk = modules.SS_Conv_SSM(hidden_dim=384, drop_path=0.0, norm_layer=torch.nn.LayerNorm).to(DEV).train()
        x = torch.randn(64, 14, 14, 384, device=DEV, requires_grad=True)
        g = torch.randn(64, 14, 14, 384, device=DEV)

        def run():
            blk.zero_grad(set_to_none=True)
            x.grad = None
            out = blk(x)
            out.backward(g)
            return out.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()}

        blas.STATS["direct"] = 0
        o1, dx1, g1 = run()
        n_direct = blas.STATS["direct"]
        saved = dict(blas._TABLE)
        blas._TABLE.clear()
        try:
            o0, dx0, g0 = run()                  # (the Python route is still patched in: its GEMMs go through torch now)
        finally:
            blas._TABLE.update(saved)
        assert n_direct >= 10, n_direct
        # the C++-sequenced routes issue the same recorded rocBLAS solutions through mm_gemm_f32 themselves (csrc_host gemm_out):
        # identical bits to the Python route with the same table
        monkeypatch.undo()
        from medmamba_amd import _host
        assert _host.module() is not None
        o2, dx2, g2 = run()
        # (the whole test runs under cudnn.deterministic — see above —: MIOpen's default picks for the dense convolutions are not
        # reproducible run to run on every box of the pool; in the reference's mode every gradient is)
        assert torch.equal(o2, o1) and torch.equal(dx2, dx1)
        for k in g1:
            assert torch.equal(g2[k], g1[k]), k
        assert torch.allclose(o1, o0, rtol=1e-4, atol=1e-4)
        assert (dx1 - dx0).abs().max().item() <= 2e-3 * dx0.abs().max().item()
        for k in g0:
            if k in ("conv33conv33conv11.1.bias", "conv33conv33conv11.4.bias"):
                continue          # a conv bias in front of a BatchNorm: the true gradient is exactly zero, what is computed is rounding noise
            assert (g1[k] - g0[k]).abs().max().item() <= 2e-3 * max(1e-4, g0[k].abs().max().item()), k
    finally:
        torch.backends.cudnn.deterministic = False
        torch.cuda.tunable.enable(False)
        blas.clear()


def _operands_of(key, batched, g):
    """Dense device operands for a TunableOp key `tn_m_n_k[_B_b]_ld_lda_ldb_ldc` (rocBLAS column-major terms)."""
    p = key.split("_")
    opa, opb = p[0][0].upper(), p[0][1].upper()
    m, n, k = int(p[1]), int(p[2]), int(p[3])
    batch = int(p[5]) if batched else 1
    lda, ldb, ldc = (int(v) for v in p[-3:])
    na = lda * (k if opa == "N" else m)
    nb = ldb * (n if opb == "N" else k)
    A = torch.randn(batch * na, device=DEV, generator=g)
    B = torch.randn(batch * nb, device=DEV, generator=g)
    return dict(opa=opa, opb=opb, m=m, n=n, k=k, batch=batch, lda=lda, ldb=ldb, ldc=ldc, sa=na, sb=nb, sc=ldc * n, A=A, B=B)


def _ref_product(o):
    """A(op) @ B(op) per batch item with torch (row-major views of the column-major operands), as (batch, n, m) = C^T."""
    b = o["batch"]
    At = o["A"].view(b, -1, o["lda"])           # column-major (lda x cols): row-major view is (cols, lda)
    Bt = o["B"].view(b, -1, o["ldb"])
    Aop = At[:, :o["k"], :o["m"]].transpose(1, 2) if o["opa"] == "N" else At[:, :o["m"], :o["k"]]       # (b, m, k)
    Bop = Bt[:, :o["n"], :o["k"]].transpose(1, 2) if o["opb"] == "N" else Bt[:, :o["k"], :o["n"]]       # (b, k, n)
    return torch.bmm(Aop, Bop).transpose(1, 2)                                                           # (b, n, m)


def test_recorded_split_k_solutions_on_two_streams_at_once():
    """a block's two branches issue GEMMs on two streams from one thread.  A rocBLAS handle owns ONE device
    workspace, so mm_gemm_f32 keeps a handle per (thread, device, stream).  Here the recorded solutions of the large-K
    weight-gradient shapes (K = B*L >= 12544: the ones rocBLAS runs split-K) are issued on two streams from this thread at the same
    time, with different operands, and must give what the same calls give one after the other."""
    from medmamba_amd import _lib, blas
    from medmamba_amd.tuning import DEFAULT_FILE
    if blas.load_table(DEFAULT_FILE) == 0:
        pytest.skip("GEMM table not recorded for this rocBLAS build / GPU")
    lib = _lib.lib()
    try:
        keys = sorted(((int(k.split("_")[3]), b, k, s) for (b, k), s in blas._TABLE.items()), reverse=True)
        picked = [x for x in keys if x[0] >= 12544][:6] + [x for x in keys if x[1] and x[0] >= 3136][:3]
        assert len(picked) >= 3, picked
        g = torch.Generator(device=DEV).manual_seed(0)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for K, batched, key, sol in picked:
            ops_ = [_operands_of(key, batched, g) for _ in range(2)]

            def launch(o, C, stream):
                rc = lib.mm_gemm_f32(o["opa"].encode(), o["opb"].encode(), o["m"], o["n"], o["k"], 1.0, o["A"].data_ptr(), o["lda"], o["sa"],
                                     o["B"].data_ptr(), o["ldb"], o["sb"], 0.0, C.data_ptr(), o["ldc"], o["sc"], o["batch"], sol,
                                     stream.cuda_stream)
                assert rc == 0, (key, rc, lib.mm_blas_last_status())

            new_c = lambda o: torch.full((o["batch"] * o["sc"],), float("nan"), device=DEV)
            serial = []
            for rep in range(2):                                   # one after the other (also: is this solution reproducible at all?)
                cs = [new_c(o) for o in ops_]
                for o, C in zip(ops_, cs):
                    launch(o, C, torch.cuda.current_stream())
                    torch.cuda.synchronize()
                serial.append(cs)
            same = lambda a, b: torch.equal(torch.nan_to_num(a, nan=7.0), torch.nan_to_num(b, nan=7.0))    # ldc > m leaves NaN pads
            reproducible = all(same(a, b) for a, b in zip(*serial))
            for o, C in zip(ops_, serial[0]):                      # and right
                got = C.view(o["batch"], o["n"], o["ldc"])[:, :, :o["m"]]
                ref = _ref_product(o)
                assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), key
            s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
            rounds = [[new_c(o) for o in ops_] for _ in range(6)]
            for cs in rounds:                                      # 12 launches in flight, alternating streams, no sync in between
                launch(ops_[0], cs[0], s1)
                launch(ops_[1], cs[1], s2)
            torch.cuda.synchronize()
            for cs in rounds:
                for C, want in zip(cs, serial[0]):
                    if reproducible:
                        assert same(C, want), f"{key}: concurrent result differs from the serial one"
                    else:                                          # a solution that accumulates with atomics: equal up to summation order
                        scale = float(want[torch.isfinite(want)].abs().max())
                        ok = torch.isfinite(want)
                        assert torch.equal(torch.isfinite(C), ok) and float((C[ok] - want[ok]).abs().max()) <= 1e-5 * scale, key
    finally:
        blas.clear()


def test_rocblas_only_records_cover_the_channel_major_blocks_of_S_and_nothing_else():
    """MM_PARAM_STREAM (off by default) may move a block's weight-gradient GEMMs to a third stream only when each of the
    four has an EXPLICIT rocBLAS solution on record (rocBLAS's own pick may be a hipBLASLt kernel, and those stop the GPU from a third
    queue): the records (gemm_gfx950.csv + gemm_gfx950_rocblas.csv) cover the 14x14 and 7x7 blocks of MedMamba-T / S at 64 images, not
    a shape nobody tuned, and every rocBLAS-only record runs and agrees with torch."""
    import csv
    import os
    from medmamba_amd import _host, _lib, blas
    from medmamba_amd.tuning import DEFAULT_FILE, enable_tuned_gemms
    enable_tuned_gemms()
    if not blas._TABLE or _host.module() is None:
        pytest.skip("GEMM table not recorded for this rocBLAS build / no C++ layer")
    try:
        cov = _host.module().ss2d_params_covered
        assert cov(64, 196, 192, 384, 44, 12) and cov(64, 49, 384, 768, 56, 24)
        assert not cov(32, 576, 256, 512, 48, 16) and not cov(64, 196, 192, 384, 44, 11) and not cov(7, 196, 192, 384, 44, 12)
        rows = list(csv.reader(open(os.path.join(os.path.dirname(DEFAULT_FILE), "gemm_gfx950_rocblas.csv"))))
        recs = [r for r in rows if len(r) >= 3 and r[2].startswith("Gemm_Rocblas_")]
        assert len(recs) >= 8
        g = torch.Generator(device=DEV).manual_seed(1)
        for r in recs:
            p = r[1].split("_")
            opa, opb, n, m, k = p[0][0], p[0][1], int(p[1]), int(p[2]), int(p[3])
            batch = int(p[5]) if p[4] == "B" else 1
            lda, ldb, ldc = (int(v) for v in p[-3:])
            A = torch.randn(batch, (k if opa == "n" else n) * lda, device=DEV, generator=g)       # column-major operands, as rocBLAS sees them
            B = torch.randn(batch, (m if opb == "n" else k) * ldb, device=DEV, generator=g)
            C = torch.empty(batch, m * ldc, device=DEV)
            rc = _lib.lib().mm_gemm_f32(opa.upper().encode(), opb.upper().encode(), n, m, k, 1.0, A.data_ptr(), lda, A.stride(0), B.data_ptr(),
                                        ldb, B.stride(0), 0.0, C.data_ptr(), ldc, C.stride(0), batch, int(r[2][len("Gemm_Rocblas_"):]),
                                        _lib.raw_stream())
            assert rc == 0, (r[1], rc)
            Aop = A.view(batch, -1, lda)[:, :k, :n].transpose(1, 2) if opa == "n" else A.view(batch, -1, lda)[:, :n, :k]       # (b, n, k)
            Bop = B.view(batch, -1, ldb)[:, :m, :k].transpose(1, 2) if opb == "n" else B.view(batch, -1, ldb)[:, :k, :m]       # (b, k, m)
            want = torch.bmm(Aop.double(), Bop.double()).transpose(1, 2).float()                  # C^T rows: (b, m, n)
            got = C.view(batch, m, ldc)[:, :, :n]
            assert (got - want).abs().max().item() <= 2e-3 * max(1.0, want.abs().max().item()), r[1]
    finally:
        blas.clear()
        torch.cuda.tunable.enable(False)
