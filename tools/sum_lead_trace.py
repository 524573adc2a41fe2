#!/usr/bin/env python3
"""Run under `rocprofv3 --kernel-trace --stats`: ops.sum_lead and torch.sum(0) on batch-sum and partial-row shapes, 20 calls each,
each shape bracketed by a marker copy so that the kernels can be told apart by order."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd import ops
dev = torch.device("cuda:0")
for shape in [(64, 2, 70, 96), (64, 768, 384), (1024, 96), (1024, 384), (1024, 768), (4096, 768), (3136, 384), (448, 768)]:
    x = torch.randn(*shape, device=dev); out = torch.empty(shape[1:], device=dev)
    for _ in range(20): ops.sum_lead(x, out=out)
    torch.cuda.synchronize()
    for _ in range(20): torch.sum(x, 0, out=out)
    torch.cuda.synchronize()
