"""Data parallelism for the MedMamba hot path: one process per GPU, replicated weights, bucketed gradient
all-reduce (sum / world) overlapped with backward — torch.distributed over RCCL (backend "nccl" on ROCm),
which rides xGMI inside a node.  The reference is single-GPU (train.py:64); this is the only strategy added
(SURVEY §8e): every (batch, direction, channel) sequence is independent, so the batch shards with no
data-path collective; the single exchange per step is the gradient all-reduce (74.5 MB fp32 for MedMamba-S).

BatchNorm statistics stay per-GPU (plain DDP semantics, no SyncBN: the reference has no multi-GPU behaviour
to match) and BN buffers are not re-broadcast every step.  Buckets: xGMI is point-to-point (7 links x ~153 GB/s
per GPU), a ring all-reduce is per-link bound, so a few large buckets (32 MB -> 3 buckets for S) amortise the
per-collective latency while still overlapping with the backward of the earlier stages.
"""
import os

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel


def init_distributed(backend=None):
    """Initialise the default process group from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    return world


def wrap_ddp(module, device=None, bucket_cap_mb=32):
    """Replicate `module` across the default process group (weights are broadcast from rank 0 once)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return module
    kw = dict(broadcast_buffers=False, gradient_as_bucket_view=True, bucket_cap_mb=bucket_cap_mb)
    if device is not None and device.type == "cuda":
        kw.update(device_ids=[device.index], output_device=device.index)
    return DistributedDataParallel(module, **kw)
