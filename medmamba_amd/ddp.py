"""Data parallelism for the MedMamba hot path: one process per GPU, replicated weights, bucketed gradient
all-reduce (sum / world) overlapped with backward — torch.distributed over RCCL (backend "nccl" on ROCm),
which rides xGMI inside a node.  The reference is single-GPU (train.py:64); this is the only strategy added
(SURVEY §8e): every (batch, direction, channel) sequence is independent, so the batch shards with no
data-path collective; the single exchange per step is the gradient all-reduce (74.5 MB fp32 for MedMamba-S).

BatchNorm statistics stay per-GPU (plain DDP semantics, no SyncBN: the reference has no multi-GPU behaviour
to match) and BN buffers are not re-broadcast every step.

Two ways to do the exchange:
* `GradSync` (what bench.py and train.py use): one bucket per model stage, its all-reduce started by ONE tensor hook at the
  stage boundary while backward continues — the bucketed, overlapped exchange of SURVEY §8e without per-parameter hooks
  (DistributedDataParallel's 365 hooks and bucket bookkeeping cost 4-5 ms of host time per step on this model,
  tools/ddp_overhead.py; four boundary hooks cost microseconds).  Per bucket: flatten (one `cat`), all-reduce on RCCL's stream,
  scale, scatter back (one multi-tensor copy).  `GradSync(net, overlap=False)` = one flat all-reduce after backward.
* `wrap_ddp`: torch's DistributedDataParallel, 32 MB buckets overlapped with backward (MM_DDP=torch).

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
    """Replicated data parallelism: bucketed gradient all-reduce (sum / world), started while backward is still running.

        sync = GradSync(net)            # broadcasts rank 0's parameters and buffers once
        loss.backward(); sync(); optimizer.step()

    Buckets follow the order in which backward finishes the gradients: the parameters are cut at the model's stage boundaries
    (`net.layers`, MedMamba.py:466-481), last stage (+ head) first.  Backward reaches the input of stage i only after everything
    behind it has been differentiated, so ONE tensor hook per boundary (on the stage's input activation) — not one hook per
    parameter, which is what makes DistributedDataParallel cost 4-5 ms of host time on this 365-parameter model — starts the
    all-reduce of that bucket (flatten with one `cat`, `all_reduce(async_op=True)`: RCCL runs it on its own stream beside the rest of
    backward).  A hook only acts when every gradient of its bucket is present (each parameter is used once per forward, so a
    gradient that is there is final); whatever is still pending when backward returns goes out in `sync()`, which then waits for
    all buckets, scales and copies back (one multi-tensor copy per bucket).  `overlap=False` (or a module without `.layers`) is the
    single flat all-reduce after backward.  Parameters whose .grad is None on this rank take part with zeros (every rank must
    contribute the same layout).  `self.stats` = dict(buckets, early: how many of them went out during backward) of the last
    step; with `timing=True`, `allreduce_ms()` reads device events around the collectives (rank-local, after a synchronize)."""

    def __init__(self, module, process_group=None, overlap=True, timing=False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.timing = timing
        self.events = []                # (start, end) device events of the collectives of the last call(s) when timing
        self.last = dict(buckets=0, early=0)
        self.stats = dict(buckets=0, early=0)
        self._pending = []              # buckets in flight: (flat, grads, work)
        self._launched = set()
        self._hold = False              # inside no_sync()
        self._stale = False             # this step's forward saw gradients of an earlier backward: no early launches
        self._hooks = []
        self.buckets = [self.params]
        if self.world > 1:
            _broadcast_from_rank0(list(module.parameters()) + list(module.buffers()), process_group)
            layers = getattr(module, "layers", None)
            if overlap and layers is not None and len(layers) > 1:
                self._cut_at_stages(module, list(layers))

    # ---- bucket layout: [stage n-1 + everything behind it] [stage n-2] ... [stage 0 + everything in front of it]
    def _cut_at_stages(self, module, layers):
        owner = {}
        for i, layer in enumerate(layers):
            for p in layer.parameters():
                owner[p] = i
        first = {i: None for i in range(len(layers))}
        order = {p: n for n, p in enumerate(module.parameters())}
        for p, i in owner.items():
            first[i] = order[p] if first[i] is None else min(first[i], order[p])
        buckets = [[] for _ in layers]
        for p in self.params:
            if p in owner:
                buckets[owner[p]].append(p)
            else:       # patch embed (in front of stage 0) -> stage 0's bucket; head (behind the last stage) -> the last stage's
                buckets[0 if order[p] < first[0] else len(layers) - 1].append(p)
        stages = [i for i in reversed(range(len(layers))) if buckets[i]]          # bucket n holds stage stages[n]
        self.buckets = [buckets[i] for i in stages]
        # the INPUT of stage i (i >= 1) has its gradient once every stage >= i has been differentiated: it closes those buckets
        for i in range(1, len(layers)):
            closes = [n for n, st in enumerate(stages) if st >= i]
            self._hooks.append(layers[i].register_forward_pre_hook(self._make_pre_hook(closes)))

    def _make_pre_hook(self, closes):
        def pre_hook(mod, args):
            x = args[0]
            if self.world > 1 and torch.is_grad_enabled() and isinstance(x, torch.Tensor) and x.requires_grad:
                # "gradient present" only means "gradient final" when the step started without gradients (zero_grad(set_to_none=True),
                # train.py:281): with gradients left from an earlier backward (accumulation over micro-batches) nothing goes out early
                if any(b[0].grad is not None or b[-1].grad is not None for b in self.buckets):
                    self._stale = True
                x.register_hook(lambda g: self._boundary(closes))
        return pre_hook

    def _boundary(self, closes):
        """Runs inside backward (autograd thread), when the gradient of a stage's input has been computed."""
        if self._stale or self._hold:
            return None
        for n in closes:
            if n not in self._launched and all(p.grad is not None for p in self.buckets[n]):
                self._launch(n, early=True)
        return None

    def _launch(self, n, early):
        grads = []
        for p in self.buckets[n]:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            grads.append(p.grad)
        if early and grads[0].is_cuda:
            # gradients of conv-branch parameters are produced on the block schedule's side stream (modules._side_stream): the
            # engine joins the streams only when backward returns, so order this stream behind the side stream's queue first
            from . import modules
            side = modules._side_stream(grads[0].device) if modules._TWO_STREAMS else None
            if side is not None:
                torch.cuda.current_stream(grads[0].device).wait_stream(side)
            from . import ops
            ops.join_param_stream(grads[0].device)      # (MM_PARAM_STREAM experiment: the SS2D parameter gradients' own stream)
        flat = torch._utils._flatten_dense_tensors(grads)
        ev = None
        if self.timing and flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        work = dist.all_reduce(flat, group=self.group, async_op=True)
        self._pending.append((flat, grads, work, ev, [g._version for g in grads] if early else None))
        self._launched.add(n)
        self.last["early"] += int(early)

    def __call__(self):
        if self.world == 1:
            return
        for n in range(len(self.buckets)):
            if n not in self._launched:
                self._launch(n, early=False)
        for flat, grads, work, ev, versions in self._pending:
            work.wait()                                  # device tensors: the current stream waits for the collective
            if versions is not None and versions != [g._version for g in grads]:
                self._pending, self._launched, self.last, self._stale = [], set(), dict(buckets=0, early=0), False
                raise RuntimeError("GradSync: gradients changed after their bucket's all-reduce had started (a second backward without "
                                   "sync() in between?) — run the backward passes of all but the last micro-batch under no_sync()")
            if ev is not None:
                ev[1].record()
                self.events.append(ev)
            flat.mul_(1.0 / self.world)
            torch._foreach_copy_(grads, list(torch._utils._unflatten_dense_tensors(flat, grads)))
        self.stats = dict(buckets=len(self._pending), early=self.last["early"])      # of the step just finished
        self._pending, self._launched, self.last, self._stale = [], set(), dict(buckets=0, early=0), False

    def no_sync(self):
        """Context for gradient accumulation: backward passes inside it start no all-reduce (the gradients stay local); the last
        micro-batch runs outside it and is followed by sync() as usual — it sees the accumulated gradients at its forward pass and
        therefore sends everything after backward (one launch per bucket, nothing early)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev, self._hold = self._hold, True
            try:
                yield
            finally:
                self._hold = prev
        return ctx()

    def allreduce_ms(self):
        """Sum of the device time between the start of each collective and the point where the step waited for it, over the calls
        since the last read (timing=True; call after torch.cuda.synchronize()).  With overlap this INCLUDES the backward work that
        ran meanwhile — it is the window, not the link time."""
        ms = sum(s.elapsed_time(e) for s, e in self.events)
        self.events = []
        return ms
