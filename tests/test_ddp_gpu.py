"""GPU rehearsal of the data-parallel path: two ranks (gloo rendezvous, both on cuda:0 — a 1-GPU box has no second device
and RCCL needs one device per rank) run GradSync on HIP tensors with the two-stream block schedule.  Checks what the CPU
gloo test cannot: gradients produced on the side stream are complete when the flat all-reduce reads them, the parameter
and buffer broadcast (fp32 + int64 BatchNorm counters) works on device tensors, and both replicas stay identical after
optimizer steps.  The 8-GPU RCCL run itself is the driver's (SCALE_rNN.json)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, outdir, mode="flat"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from medmamba_amd import modules
    from medmamba_amd.ddp import GradSync, init_distributed, wrap_ddp
    from medmamba_amd.trainer import offset_device_rng
    assert init_distributed("gloo") == world
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(7 + rank)                 # different initial replicas: the broadcast must make them equal
    net = modules.VSSM(num_classes=3, depths=[1, 1, 1, 1], dims=[16, 32, 64, 128], drop_path_rate=0.0).to(dev).train()
    assert modules._TWO_STREAMS                     # the block schedule under test: conv branch on the side stream
    if mode == "torch":                             # DistributedDataParallel (MM_DDP=torch): bucketed, overlapped with backward
        model, sync = wrap_ddp(net, dev, bucket_cap_mb=1), None      # small buckets: several all-reduces start mid-backward
    else:
        model, sync = net, GradSync(net, overlap=(mode == "bucketed"), timing=True)
    offset_device_rng(rank, 7)                      # per-rank device RNG (DropPath masks) after the replicas are identical
    draw = torch.rand(8, device=dev).cpu()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, fused=True)
    g = torch.Generator().manual_seed(100)
    x = torch.randn(8, 3, 64, 64, generator=g)[4 * rank:4 * rank + 4].to(dev)
    y = torch.tensor([0, 1, 2, 1, 2, 0, 1, 1])[4 * rank:4 * rank + 4].to(dev)
    losses = []
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model(x), y)
        loss.backward()
        if sync is not None:
            sync()
        opt.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    if sync is not None:
        assert sync.stats == (dict(buckets=4, early=3) if mode == "bucketed" else dict(buckets=1, early=0)), sync.stats
        assert sync.allreduce_ms() > 0.0
    torch.save(dict(params={k: p.detach().cpu() for k, p in net.named_parameters()}, losses=losses, draw=draw,
                    grads={k: p.grad.detach().cpu() for k, p in net.named_parameters()}), os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["bucketed", "flat", "torch"])
def test_two_ranks_on_one_gpu_stay_identical(tmp_path, mode):
    """mode "bucketed": GradSync as bench.py uses it — per-stage buckets whose all-reduce starts at the stage boundary inside
    backward, while gradients of conv-branch parameters are still being produced on the side stream; mode "flat": one all-reduce
    after backward; mode "torch": wrap_ddp = DistributedDataParallel on HIP tensors with the
    two-stream block schedule (gradients of conv-branch parameters are produced on the side stream while buckets are being
    reduced).  Same assertions: bitwise identical gradients and parameters on both ranks."""
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), mode), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "r1.pt", weights_only=True)
    for k in r0["params"]:
        assert torch.equal(r0["grads"][k], r1["grads"][k]), k       # all-reduced gradients: bitwise identical
        assert torch.equal(r0["params"][k], r1["params"][k]), k     # same start (broadcast) + same updates
    assert all(abs(a) < 1e3 for a in r0["losses"] + r1["losses"])
    assert r0["losses"] != r1["losses"]                             # different data shards
    assert not torch.equal(r0["draw"], r1["draw"])                  # per-rank device RNG: DropPath masks differ across ranks
