#!/usr/bin/env python3
"""Backward scan kernel alone (hipEvents around mm_scan_bwd only — no partial-plane sums, no allocation), per MedMamba stage
shape in the storage layout the model uses there (batch-major for long sequences, channel-major for L <= 256), SS2D operand
sharing (2 u blocks, reversed directions), for a list of `variant` words (selective_scan_interface._BWD_VARIANT: bits 8-15
passes, 16-23 waves per workgroup, 24 / 25 = 2 / 4 states per lane).

    python tools/bench_scan_bwd.py [S|B] [batch] [variant,variant,...] [stage,stage,...]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd import ops, selective_scan_interface as ssi  # noqa: E402

STAGES = {"S": [(96, 3136, 2), (192, 784, 2), (384, 196, 8), (768, 49, 2)],
          "B": [(128, 9216, 2), (256, 2304, 2), (512, 576, 12), (1024, 144, 2)]}


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "S"
    Bz = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    variants = [int(v, 0) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
    stages = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 1, 2, 3]
    dev = torch.device("cuda:0")
    K, N = 4, 16
    total = {v: 0.0 for v in variants}
    tot_bytes = 0
    for si, (D, L, nblk) in enumerate(STAGES[model]):
        if si not in stages:
            continue
        R = max(1, (D // 2 + 15) // 16)
        C = R + 2 * N
        cm = ops.channel_major(Bz, L)
        g = torch.Generator(device=dev).manual_seed(0)
        pl = (lambda ch: torch.randn(ch, Bz, L, device=dev, generator=g).permute(1, 0, 2)) if cm else \
            (lambda ch: torch.randn(Bz, ch, L, device=dev, generator=g))
        u2, delta, dout2 = pl(2 * D), pl(K * D), pl(2 * D)
        A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(K * D, 1)
        if cm:
            x_dbl = torch.randn(K, C, Bz * L, device=dev, generator=g)
            xb = x_dbl.view(K, C, Bz, L).permute(2, 0, 1, 3)
            dxb = torch.empty_like(x_dbl).view(K, C, Bz, L).permute(2, 0, 1, 3)
        else:
            xb = torch.randn(Bz, K, C, L, device=dev, generator=g)
            dxb = torch.empty_like(xb)
        Dp = torch.ones(K * D, device=dev)
        bias = torch.randn(K * D, device=dev, generator=g) - 4.0
        _, x_chk = ssi._launch_fwd(u2, delta, A, xb[:, :, R:R + N], xb[:, :, R + N:], Dp, bias, True, True, 0, ssi._CROSS_SHARED)
        if os.environ.get("FWD_VARIANTS"):      # forward kernel in the same operand layouts: FWD_VARIANTS=0,2,4,0x1000004 ...
            fb = ssi.scan_bytes_fwd(Bz, K * D, L, N, K)
            for fv in [int(v, 0) for v in os.environ["FWD_VARIANTS"].split(",")]:
                for chk in ((False, True) if "FWD_CHK" not in os.environ else (os.environ["FWD_CHK"] == "1",)):
                    runf = lambda: ssi._launch_fwd(u2, delta, A, xb[:, :, R:R + N], xb[:, :, R + N:], Dp, bias, True, chk, fv or 0,
                                                   ssi._CROSS_SHARED)
                    ssi._FWD_VARIANT = fv
                    for _ in range(8):
                        runf()
                    torch.cuda.synchronize()
                    ssi.KERNEL_TIMER.records.clear()
                    ssi.KERNEL_TIMER.enabled = True
                    for _ in range(12):
                        runf()
                    torch.cuda.synchronize()
                    ssi.KERNEL_TIMER.enabled = False
                    ts = sorted(s.elapsed_time(e) for tag, s, e, *_ in ssi.KERNEL_TIMER.records if tag == "scan_fwd")
                    med = ts[len(ts) // 2]
                    print(json.dumps(dict(model=model, batch=Bz, D=D, L=L, layout="cm" if cm else "bm", fwd_variant=hex(fv), chk=chk,
                                          fwd_kernel_ms=round(med, 4), frac_8TBs=round(fb / med / 8e9, 4))), flush=True)
            ssi._FWD_VARIANT = 0
        nbytes = ssi.scan_bytes_bwd(Bz, K * D, L, N, K)
        tot_bytes += nbytes * nblk
        for v in variants:
            ssi._BWD_VARIANT = v
            run = lambda: ssi._launch_bwd(u2, delta, A, xb[:, :, R:R + N], xb[:, :, R + N:], Dp, bias, x_chk, dout2, True,
                                          ssi._CROSS_SHARED, dBC_dst=dxb[:, :, R:], channel_major=cm)
            try:
                for _ in range(8):
                    run()
                torch.cuda.synchronize()
                ssi.KERNEL_TIMER.records.clear()
                ssi.KERNEL_TIMER.enabled = True
                for _ in range(12):
                    run()
                torch.cuda.synchronize()
                ssi.KERNEL_TIMER.enabled = False
                ts = sorted(s.elapsed_time(e) for tag, s, e, *_ in ssi.KERNEL_TIMER.records if tag == "scan_bwd")
                med = ts[len(ts) // 2]
            except Exception as e:  # noqa
                print(f"D={D} L={L} variant={v:#x}: {e}")
                continue
            total[v] += med * nblk
            print(json.dumps(dict(model=model, batch=Bz, D=D, L=L, layout="cm" if cm else "bm", variant=hex(v), bwd_kernel_ms=round(med, 4),
                                  min_ms=round(ts[0], 4), GBs=round(nbytes / med / 1e6, 1), frac_8TBs=round(nbytes / med / 8e9, 4))), flush=True)
    for v in variants:
        if total[v] > 0:
            print(json.dumps(dict(summary="scan bwd kernel, all blocks of one model backward", model=model, batch=Bz, variant=hex(v),
                                  ms=round(total[v], 3), GB=round(tot_bytes / 1e9, 3), frac_8TBs=round(tot_bytes / total[v] / 8e9, 4))))


if __name__ == "__main__":
    main()
