"""Data parallelism for the MedMamba hot path: one process per GPU, replicated weights, bucketed gradient
all-reduce (sum / world) overlapped with backward — torch.distributed over RCCL (backend "nccl" on ROCm),
which rides xGMI inside a node.  The reference is single-GPU (train.py:64); this is the only strategy added
(SURVEY §8e): every (batch, direction, channel) sequence is independent, so the batch shards with no
data-path collective; the single exchange per step is the gradient all-reduce (74.5 MB fp32 for MedMamba-S).

BatchNorm statistics stay per-GPU (plain DDP semantics, no SyncBN: the reference has no multi-GPU behaviour
to match) and BN buffers are not re-broadcast every step.

Two ways to do the exchange:
* `GradSync` (what bench.py uses): ONE all-reduce of the flattened gradients after backward.  The step is bound by the
  host's launch rate, not by the link: DistributedDataParallel's per-parameter hooks and bucket bookkeeping cost
  3.9 ms of host time per step on this model (265 parameters; tools/ddp_overhead.py, world-size-1 RCCL group: 35.1 ->
  39.0 ms), while the whole 74.5 MB all-reduce is ~0.5-1 ms on xGMI (7 links x ~153 GB/s per GPU) — hiding it behind the
  backward buys less than the hooks cost.  flatten (one `cat`), all-reduce, scale, scatter back (one multi-tensor copy).
* `wrap_ddp`: torch's DistributedDataParallel, 32 MB buckets overlapped with backward — for models whose gradient
  exchange is long enough to be worth hiding.

The process group is initialised WITHOUT `device_id=`: eager communicator binding costs 6 ms of host time on every
step (tools/pg_overhead.py: 35.2 -> 41.2 ms); the communicator is created lazily by the first collective instead.
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
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (see bench.py); read when RCCL first shares memory
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend)
    return world


def wrap_ddp(module, device=None, bucket_cap_mb=32):
    """Replicate `module` across the default process group (weights are broadcast from rank 0 once)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return module
    _broadcast_from_rank0(list(module.buffers()))      # once; DDP itself skips buffers when broadcast_buffers=False
    kw = dict(broadcast_buffers=False, gradient_as_bucket_view=True, bucket_cap_mb=bucket_cap_mb)
    if device is not None and device.type == "cuda":
        kw.update(device_ids=[device.index], output_device=device.index)
    return DistributedDataParallel(module, **kw)


def _broadcast_from_rank0(tensors, group=None):
    """One broadcast per dtype (fp32 weights / statistics, int64 BatchNorm counters) of the flattened tensors."""
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t.data)
    with torch.no_grad():
        for ts in by_dtype.values():
            flat = torch._utils._flatten_dense_tensors(ts)
            dist.broadcast(flat, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            torch._foreach_copy_(ts, list(torch._utils._unflatten_dense_tensors(flat, ts)))


class GradSync:
    """Replicated data parallelism with one collective per step.

        sync = GradSync(net)            # broadcasts rank 0's parameters and buffers once
        loss.backward(); sync(); optimizer.step()

    `sync()` averages the gradients of all ranks: flatten -> all-reduce(sum) -> * 1/world -> copy back.  Parameters whose
    .grad is None on this rank take part with zeros (every rank must contribute the same layout)."""

    def __init__(self, module, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        if self.world > 1:
            _broadcast_from_rank0(list(module.parameters()) + list(module.buffers()), process_group)

    def __call__(self):
        if self.world == 1:
            return
        grads = []
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            grads.append(p.grad)
        flat = torch._utils._flatten_dense_tensors(grads)
        dist.all_reduce(flat, group=self.group)
        flat.mul_(1.0 / self.world)
        torch._foreach_copy_(grads, list(torch._utils._unflatten_dense_tensors(flat, grads)))
