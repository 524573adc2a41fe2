#!/usr/bin/env python3
"""Scan-kernel micro-benchmark on one MI355X: per MedMamba stage shape, per variant, forward and
backward time (hipEvents on torch's current stream, which is the stream the kernels are launched on),
algorithmic bytes and the implied HBM GB/s.  Usage: python tools/bench_scan.py [S|B] [batch]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.selective_scan_interface import SelectiveScanFn  # noqa: E402

STAGES = {"S": [(96, 3136, 2), (192, 784, 2), (384, 196, 8), (768, 49, 2)],
          "B": [(128, 9216, 2), (256, 2304, 2), (512, 576, 12), (1024, 144, 2)]}


def bytes_fwd(Bz, K, D, N, L):
    return 4 * Bz * L * (3 * K * D + 2 * K * N) + 4 * (K * D * N + 2 * K * D)


def bytes_bwd(Bz, K, D, N, L):
    return 4 * Bz * L * (5 * K * D + 4 * K * N) + 8 * (K * D * N + 2 * K * D)


def timeit(fn, iters=20, warmup=12):     # the first launches of a process run at ramping clocks
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "S"
    Bz = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2, 4]
    stages = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 1, 2, 3]
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
    do_bwd = (sys.argv[6] != "nobwd") if len(sys.argv) > 6 else True
    dev = torch.device("cuda:0")
    K, N = 4, 16
    rows = []
    for si, (D, L, nblk) in enumerate(STAGES[model]):
        if si not in stages:
            continue
        R = max(1, (D // 2 + 15) // 16)
        g = torch.Generator(device=dev).manual_seed(0)
        u = torch.randn(Bz, K * D, L, device=dev, generator=g)
        delta = torch.randn(Bz, K * D, L, device=dev, generator=g)
        A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(K * D, 1)
        x_dbl = torch.randn(Bz, K, R + 2 * N, L, device=dev, generator=g)
        Bs, Cs = x_dbl[:, :, R:R + N], x_dbl[:, :, R + N:]
        Dp = torch.ones(K * D, device=dev)
        bias = torch.randn(K * D, device=dev, generator=g) - 4.0
        dout = torch.randn(Bz, K * D, L, device=dev, generator=g)
        bf, bb = bytes_fwd(Bz, K, D, N, L), bytes_bwd(Bz, K, D, N, L)
        for v in variants:
            try:
                med, mn = timeit(lambda: SelectiveScanFn.apply(u, delta, A, Bs, Cs, Dp, bias, True, v), iters=iters)
            except Exception as e:  # noqa
                print(f"D={D} L={L} variant={v}: {e}")
                continue
            row = dict(model=model, batch=Bz, D=D, L=L, blocks=nblk, variant=v, fwd_ms=med, fwd_min_ms=mn,
                       fwd_GBs=bf / med / 1e6, fwd_frac_8TBs=bf / med / 1e6 / 8000)
            rows.append(row)
            print(json.dumps(row), flush=True)
        if not do_bwd:
            continue
        # training forward (writes checkpoints) + backward, default variant
        ins = [t.detach().requires_grad_() for t in (u, delta, A, Bs, Cs, Dp, bias)]
        med_f, _ = timeit(lambda: SelectiveScanFn.apply(*ins, True, 0))
        out = SelectiveScanFn.apply(*ins, True, 0)
        med_b, mn_b = timeit(lambda: torch.autograd.grad(out, ins, dout, retain_graph=True), iters=10)
        if os.environ.get("MM_BWD_WAVES"):
            pass
        row = dict(model=model, batch=Bz, D=D, L=L, blocks=nblk, variant="train", fwd_chk_ms=med_f, bwd_ms=med_b,
                   bwd_min_ms=mn_b, bwd_GBs=bb / med_b / 1e6, note="bwd_ms includes zero-fill of dA/dB/dC")
        rows.append(row)
        print(json.dumps(row), flush=True)
    tot_f = sum(r["fwd_ms"] * r["blocks"] for r in rows if r["variant"] == 0)
    tot_bytes = sum(bytes_fwd(Bz, K, D, N, L) * n for si, (D, L, n) in enumerate(STAGES[model]) if si in stages)
    if tot_f > 0:
      print(json.dumps(dict(summary="scan fwd, all blocks of one model forward", model=model, batch=Bz, ms=tot_f,
                          GB=tot_bytes / 1e9, GBs=tot_bytes / tot_f / 1e6, frac_8TBs=tot_bytes / tot_f / 1e6 / 8000)))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rows, open(f"gpurun_out/bench_scan_{model}_{Bz}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
