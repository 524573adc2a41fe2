#!/usr/bin/env python3
"""csrc/conv.hip against MIOpen on the conv-branch shapes of MedMamba-S at 64 images (MedMamba.py:339-343): what one
"conv + bias -> BatchNorm statistics" costs each way (the BatchNorm apply pass is common to both and not timed).
  miopen: F.conv2d(x, w, b) [MIOpen solver + its layout transposes + bias add] + bn_stats (mm_bn_relu_fwd's first kernel is not
          separable through the ABI, so the whole mm_bn_relu_fwd is timed and the apply-only mm_bn_relu_fwd_stats subtracted)
  own   : mm_conv3x3_fwd with bias and statistics in the epilogue.
Run under rocprofv3 --kernel-trace --stats for the per-kernel view (profiles/r3_conv_branch.txt)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd import _lib  # noqa: E402


def timeit(fn, iters=30, warmup=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return 1e3 * ts[len(ts) // 2]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda:0")
    lib = _lib.exp_lib()      # python -m medmamba_amd.build --experiments
    st = torch.cuda.current_stream().cuda_stream
    print(f"{'shape':<22} {'miopen conv+bias us':>20} {'bn (stats+apply) us':>20} {'bn apply only us':>17} {'own conv+bias+stats us':>23} {'TFLOP/s own':>12}")
    for C, HW in [(48, 56), (96, 28), (192, 14), (384, 7)]:
        x = torch.randn(B, C, HW, HW, device=dev)
        w = torch.randn(C, C, 3, 3, device=dev) / (3.0 * C ** 0.5)
        b = torch.randn(C, device=dev)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        y = torch.empty_like(x); z = torch.empty_like(x)
        mean, rstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
        ws = torch.empty(3 * C * lib.mm_bn_splits(B, C, HW * HW), device=dev)
        stats = torch.empty(lib.mm_conv3x3_fwd_tiles(B, HW, HW), C, 3, device=dev)
        t_mi = timeit(lambda: torch.nn.functional.conv2d(x, w, b, padding=1))
        t_bn = timeit(lambda: lib.mm_bn_relu_fwd(y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, None, None, z.data_ptr(),
                                                 mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), None, 1, B, C, HW * HW, st))
        t_own = timeit(lambda: lib.mm_conv3x3_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, 0, y.data_ptr(), stats.data_ptr(),
                                                  B, C, C, HW, HW, st))
        t_ap = timeit(lambda: lib.mm_bn_relu_fwd_stats(y.data_ptr(), stats.data_ptr(), stats.shape[0], gamma.data_ptr(), beta.data_ptr(),
                                                       1e-5, 0.1, None, None, z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 1, B, C,
                                                       HW * HW, st))
        wt = w.permute(1, 2, 3, 0).reshape(C * 9, C).contiguous()
        stats2 = torch.empty(lib.mm_conv3x3_v2_tiles(B, HW, HW), C, 3, device=dev)
        t_v2 = timeit(lambda: lib.mm_conv3x3_v2_fwd(x.data_ptr(), wt.data_ptr(), b.data_ptr(), None, 0, y.data_ptr(), stats2.data_ptr(),
                                                    B, C, C, HW, HW, st))
        flop = 2.0 * B * HW * HW * C * C * 9
        print(f"{B}x{C}x{HW}x{HW:<12} {t_mi:>20.1f} {t_bn:>20.1f} {t_ap:>17.1f} {t_own:>23.1f} {flop / t_own / 1e6:>12.1f}   v2: {t_v2:7.1f} us {flop / t_v2 / 1e6:6.1f} TFLOP/s")


if __name__ == "__main__":
    main()
