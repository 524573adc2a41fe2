"""CPU, world_size 2, gloo: the data-parallel wrapper averages gradients across ranks exactly like a
single-process run on the concatenated batch does for batch-independent parameters.  The scan inside the
model is the oracle injected by the TEST in each worker (no GPU here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build_model():
    from medmamba_amd import modules as M
    from oracle.scan_ref import c_cross_scan_fn, c_selective_scan_fn
    M.selective_scan_fn = c_selective_scan_fn          # test doubles; the product has no CPU scan
    M.cross_scan_fn = c_cross_scan_fn
    from oracle.model_ref import (block_split_ref, dwconv_silu_cross_ref, in_proj_cf_ref, shuffle_residual_ref,
                                  ss2d_conv_core_ref, ss2d_core_ref)
    M.shuffle_residual, M.dwconv_silu_cross, M.ss2d_core = shuffle_residual_ref, dwconv_silu_cross_ref, ss2d_core_ref
    M.ss2d_conv_core = ss2d_conv_core_ref
    M.block_split, M.in_proj_cf = block_split_ref, in_proj_cf_ref
    torch.manual_seed(7)
    net = M.VSSM(num_classes=3, depths=[1, 1], dims=[16, 32], drop_path_rate=0.0)
    net.eval()       # BatchNorm uses running stats -> per-sample independence -> DDP mean == big-batch gradient
    return net


def _worker(rank, world, port, outdir, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from medmamba_amd.ddp import GradSync, init_distributed, wrap_ddp
    assert init_distributed("gloo") == world
    net = _build_model()
    if rank == 1:                      # prove the initial broadcast: perturb rank 1's copy before wrapping
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
            for b in net.buffers():
                if b.is_floating_point():
                    b.add_(0.5)
    model, sync = (wrap_ddp(net), None) if mode == "torch" else (net, GradSync(net, overlap=(mode == "bucketed")))
    g = torch.Generator().manual_seed(100)
    x = torch.randn(4, 3, 16, 16, generator=g)[2 * rank:2 * rank + 2]
    y = torch.tensor([0, 1, 2, 1])[2 * rank:2 * rank + 2]
    stats = []
    for _ in range(2):                 # twice: the boundary hooks are per forward pass
        net.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model(x), y)
        loss.backward()
        if sync is not None:
            sync()
            stats.append(dict(sync.stats))
    grads = {k: p.grad.clone() for k, p in net.named_parameters()}
    accum, accum_stats, misuse = {}, None, ""
    if sync is not None:               # gradient accumulation: two micro-batches, one exchange
        net.zero_grad(set_to_none=True)
        with sync.no_sync():
            torch.nn.functional.cross_entropy(model(x), y).backward()
        torch.nn.functional.cross_entropy(model(x), y).backward()
        sync()
        accum = {k: p.grad.clone() for k, p in net.named_parameters()}
        accum_stats = dict(sync.stats)
        if mode == "bucketed":         # the misuse the version check is there for: a second backward after an early launch, no no_sync()
            net.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(model(x), y).backward()
            torch.nn.functional.cross_entropy(model(x), y).backward()
            try:
                sync()
            except RuntimeError as e:
                misuse = str(e)
    torch.save(dict(grads=grads, accum=accum, accum_stats=accum_stats, misuse=misuse, loss=float(loss), stats=stats,
                    nbuckets=0 if sync is None else len(sync.buckets)), os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["bucketed", "flat", "torch"])
def test_ddp_gradients_match_single_process(tmp_path, monkeypatch, mode):
    """mode "bucketed": medmamba_amd.ddp.GradSync as bench.py / train.py use it (one bucket per stage, its all-reduce started by the
    stage-boundary hook during backward); mode "flat": GradSync(overlap=False), one all-reduce after backward;
    mode "torch": DistributedDataParallel through wrap_ddp."""
    from medmamba_amd import modules as M
    for name in ("selective_scan_fn", "cross_scan_fn", "shuffle_residual", "dwconv_silu_cross", "ss2d_core", "ss2d_conv_core",
                 "block_split", "in_proj_cf"):
        monkeypatch.setattr(M, name, getattr(M, name))      # restore the product functions after this test
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), mode), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "r1.pt", weights_only=True)
    for k in r0["grads"]:
        assert torch.equal(r0["grads"][k], r1["grads"][k]), k          # all-reduced: identical on both ranks
    if mode == "bucketed":             # two stages -> two buckets; the last stage's went out DURING backward, in both passes
        assert r0["nbuckets"] == 2 and r0["stats"] == [dict(buckets=2, early=1)] * 2 == r1["stats"], r0["stats"]
        assert "no_sync" in r0["misuse"] and "no_sync" in r1["misuse"]
    elif mode == "flat":
        assert r0["nbuckets"] == 1 and r0["stats"] == [dict(buckets=1, early=0)] * 2
    if mode != "torch":                # accumulation over two equal micro-batches under no_sync(): twice the gradient, nothing early
        assert r0["accum_stats"]["early"] == 0
        for k in r0["grads"]:
            assert torch.allclose(r0["accum"][k], 2 * r0["grads"][k], rtol=1e-5, atol=1e-7), k
            assert torch.equal(r0["accum"][k], r1["accum"][k]), k
    net = _build_model()
    g = torch.Generator().manual_seed(100)
    x = torch.randn(4, 3, 16, 16, generator=g)
    y = torch.tensor([0, 1, 2, 1])
    torch.nn.functional.cross_entropy(net(x), y).backward()           # mean over 4 == mean of the two rank means
    for k, p in net.named_parameters():
        w = p.grad
        err = (r0["grads"][k] - w).abs().max().item()
        assert err <= 1e-5 * max(1.0, w.abs().max().item()), (k, err)


def test_wrap_is_identity_without_process_group():
    from medmamba_amd.ddp import GradSync, wrap_ddp
    m = torch.nn.Linear(2, 2)
    assert wrap_ddp(m) is m
    m(torch.ones(1, 2)).sum().backward()
    g = m.weight.grad.clone()
    GradSync(m)()                       # world size 1: nothing to exchange
    assert torch.equal(m.weight.grad, g)


def _shard_worker(rank, world, port, root, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from medmamba_amd import trainer as T
    from medmamba_amd.ddp import init_distributed
    assert init_distributed("gloo") == world
    tr = T.NpzBatches(root, "train", 4, 8, torch.device("cpu"), shuffle=True, seed=5, rank=rank, world=world)
    va = T.NpzBatches(root, "val", 4, 8, torch.device("cpu"), shuffle=False, rank=rank, world=world)
    epochs = [tr.shard_indices(e).tolist() for e in range(2)]
    steps = [sum(1 for _ in tr) for _ in range(2)]                 # two epochs: the reader advances its own epoch counter
    # a "model" that predicts the label stored in pixel (0, 0): right on even sample ids, wrong on odd ones (see the fixture)
    class Net(torch.nn.Module):
        def forward(self, x):
            lab = ((x[:, 0, 0, 0] * 0.5 + 0.5) * 255.0).round().long()
            return torch.nn.functional.one_hot(lab, 4).float()
    correct, seen = T.evaluate(Net(), va, return_counts=True)
    acc = T.sharded_accuracy(correct, seen)
    torch.save(dict(epochs=epochs, steps=steps, len=len(tr), val=va.shard_indices(0).tolist(), correct=correct, seen=seen, acc=acc),
               os.path.join(outdir, f"s{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_npz_shards_are_disjoint_and_cover_the_set(tmp_path):
    """ADVICE r2 (train.py under torchrun): one epoch = one pass over the training set split into disjoint per-rank shards of a
    permutation shared by all ranks (padded to equal step counts); validation shards are disjoint, unpadded, and the accuracy
    is the ratio of the all-reduced COUNTS."""
    n_train, n_val = 37, 21
    rng = np.random.default_rng(0)
    root = tmp_path / "data"
    root.mkdir()
    np.save(root / "train_images.npy", rng.integers(0, 255, (n_train, 8, 8), dtype=np.uint8))
    np.save(root / "train_labels.npy", rng.integers(0, 4, n_train))
    vl = rng.integers(0, 4, n_val)
    vi = rng.integers(0, 255, (n_val, 8, 8), dtype=np.uint8)
    pred = np.where(np.arange(n_val) % 2 == 0, vl, (vl + 1) % 4)     # what the test "model" will answer: right on even ids
    vi[:, 0, 0] = pred
    np.save(root / "val_images.npy", vi)
    np.save(root / "val_labels.npy", vl)
    world, port = 2, _free_port()
    out = tmp_path / "out"
    out.mkdir()
    mp.spawn(_shard_worker, args=(world, port, str(root), str(out)), nprocs=world, join=True)
    r = [torch.load(out / f"s{k}.pt", weights_only=True) for k in range(world)]
    for e in range(2):
        a, b = r[0]["epochs"][e], r[1]["epochs"][e]
        assert len(a) == len(b) == (n_train + 1) // 2                       # padded to equal length
        assert set(a) | set(b) == set(range(n_train))                        # together: the whole set
        assert len(set(a) & set(b)) <= 1                                     # disjoint up to the one wrap-around sample
    assert r[0]["epochs"][0] != r[0]["epochs"][1]                            # a new permutation every epoch
    assert r[0]["steps"] == r[1]["steps"] == [r[0]["len"]] * 2 == [5, 5]     # same number of optimizer steps on every rank
    assert sorted(r[0]["val"] + r[1]["val"]) == list(range(n_val))           # validation: disjoint, complete, unpadded
    assert r[0]["seen"] + r[1]["seen"] == n_val
    want = float((pred == vl).sum()) / n_val
    assert r[0]["acc"] == r[1]["acc"] == pytest.approx(want)
    assert r[0]["correct"] / r[0]["seen"] != pytest.approx(want)             # a single shard's accuracy is NOT the answer
