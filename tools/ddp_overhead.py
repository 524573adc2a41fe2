#!/usr/bin/env python3
"""Diagnostic: cost of the DistributedDataParallel machinery itself on ONE GPU — a world-size-1 RCCL process group
(all-reduce of one rank), same model / batch / optimizer as bench.py, with and without the DDP wrapper."""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
from medmamba_amd.tuning import enable_tuned_gemms
from torch.nn.parallel import DistributedDataParallel
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1)      # lazy init: device_id= costs 6 ms per step (tools/pg_overhead.py)
enable_tuned_gemms()
class _ForcedSync:          # GradSync with the world-size-1 short-cut removed, to time its launches (flat or per-stage buckets)
    def __init__(self, net, bucketed):
        from medmamba_amd.ddp import GradSync
        self.s = GradSync(net); self.s.world = 2
        if bucketed:
            self.s._cut_at_stages(net, list(net.layers))
    def __call__(self):
        self.s();


def run(ddp, bucket_mb=32, flat=False, bucketed=False):
    torch.manual_seed(42)
    net = VSSM(num_classes=6, **MEDMAMBA_CONFIGS["S"]).to(dev).train()
    model = DistributedDataParallel(net, device_ids=[0], output_device=0, broadcast_buffers=False, gradient_as_bucket_view=True,
                                    bucket_cap_mb=bucket_mb) if ddp else net
    from medmamba_amd.optim import FusedAdamW
    opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)          # bench.py's optimizer
    x = torch.randn(64, 3, 224, 224, device=dev); y = torch.randint(0, 6, (64,), device=dev)
    sync = _ForcedSync(net, bucketed) if (flat or bucketed) else None
    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model(x), y); loss.backward()
        if sync is not None: sync()
        opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t2 - t0) / 20 * 1e3, (t1 - t0) / 20 * 1e3
import warnings; warnings.filterwarnings("ignore")
for ddp, mb, flat, bucketed in ((False, 0, False, False), (True, 32, False, False), (False, 0, True, False), (False, 0, False, True),
                                (False, 0, False, False)):
    ms, enq = run(ddp, mb, flat, bucketed)
    print(f"ddp={ddp} bucket_cap_mb={mb} gradsync_flat={flat} gradsync_bucketed={bucketed}: {ms:.2f} ms/step (host enqueue {enq:.2f} ms)", flush=True)
dist.destroy_process_group()
