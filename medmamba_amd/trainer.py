"""Training / evaluation loop and checkpoint format of the reference trainer, on the HIP model.

What is mirrored (so that a user of the reference's `train.py` can switch and keep their files):
  * flag names and per-dataset-type defaults            train.py:38-55, 71-85
  * optimizer / scheduler choice                        train.py:187-201  (NPZ: AdamW(lr) + MultiStepLR([50, 75], 0.1);
                                                                           ImageFolder: AdamW(lr, wd 1e-4), no scheduler)
  * the step loop                                       train.py:277-288  (zero_grad, forward, CE, backward, step)
  * per-epoch validation accuracy                       train.py:293-308
  * the checkpoint dict and file names                  train.py:310-339, 349-362
        {epoch, model_state_dict, optimizer_state_dict, best_acc, num_classes, class_indices[, scheduler_state_dict]}
        best -> {model_name}_epoch_{e}_best.pth (previous best removed), last -> {model_name}_epoch_{e}_last.pth
  * resume with per-key fallbacks                       train.py:208-260
  * early stopping on stale validation accuracy         train.py:31-36, 345-347
What is ours: one process per GPU under torchrun with `medmamba_amd.ddp.GradSync` (one flat gradient all-reduce per step over
RCCL), `--synthetic` data (tensors resident in HBM, no files), losses accumulated on the device (the reference calls
`loss.item()` twice per step — a host sync per step), and checkpoints read with `weights_only=True`.
Dataset file handling (ImageFolder / NPZ readers, augmentation) is outside the hot path: NPZ arrays are read
with numpy, ImageFolder needs torchvision and is refused with a clear message when that package is absent.
"""
import json
import logging
import os
import random

import numpy as np
import torch
import torch.nn as nn

from .modules import MEDMAMBA_CONFIGS, VSSM

log = logging.getLogger("medmamba_amd.trainer")

CHECKPOINT_KEYS = ("epoch", "model_state_dict", "optimizer_state_dict", "best_acc", "num_classes", "class_indices")


def set_seed(seed):
    """train.py:21-29 (the cudnn flags select deterministic MIOpen solvers on ROCm)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False


def offset_device_rng(rank, seed):
    """After the (rank-identical) weight initialisation: give every rank its own device RNG stream, so that DropPath masks and
    Dropout differ across the replicas of a data-parallel job — the model replicas stay identical, the
    stochastic-depth draws become independent samples like the data shards."""
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed + 1000003 * (int(rank) + 1))


def dataset_defaults(is_npz, epochs=None, batch_size=None, lr=None):
    """(epochs, batch_size, lr, lr_decay_epochs) — train.py:71-85."""
    if is_npz:
        return (100 if epochs is None else epochs, 100 if batch_size is None else batch_size, 1e-3 if lr is None else lr, [50, 75])
    return (150 if epochs is None else epochs, 64 if batch_size is None else batch_size, 1e-4 if lr is None else lr, [])


def build_model(size, num_classes, attn_drop_rate=0.0, **kw):
    """train.py:179-182."""
    return VSSM(num_classes=num_classes, attn_drop_rate=attn_drop_rate, **MEDMAMBA_CONFIGS[size], **kw)


def make_optimizer(net, is_npz, lr, lr_decay_epochs, fused=None):
    """train.py:187-201.  `fused`: AdamW's multi-tensor kernel (same update rule); default = on for HIP parameters."""
    params = list(net.parameters())
    if fused is None:
        fused = bool(params) and params[0].is_cuda
    if fused:                     # same optimizer and state, per-step Python bookkeeping cached (optim.FusedAdamW)
        from .optim import FusedAdamW as AdamW
    else:
        AdamW = torch.optim.AdamW
    if is_npz:
        opt = AdamW(params, lr=lr)
    else:
        opt = AdamW(params, lr=lr, betas=(0.9, 0.999), weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=lr_decay_epochs, gamma=0.1) if (is_npz and lr_decay_epochs) else None
    return opt, sched


def checkpoint_dict(epoch, net, optimizer, scheduler, best_acc, num_classes, class_indices):
    """The dict of train.py:310-319 / 351-360."""
    d = {"epoch": int(epoch), "model_state_dict": net.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
         "best_acc": float(best_acc), "num_classes": int(num_classes), "class_indices": dict(class_indices)}
    if scheduler is not None:
        d["scheduler_state_dict"] = scheduler.state_dict()
    return d


def save_checkpoint(path, **kw):
    torch.save(checkpoint_dict(**kw), path)
    return path


def load_checkpoint(path, net, optimizer=None, scheduler=None, map_location="cpu"):
    """Resume logic of train.py:208-260 with its per-key fallbacks.  Returns (start_epoch, best_acc, checkpoint).
    The file is read with weights_only=True: tensors, numbers, strings, dicts and lists only (reference checkpoints hold
    nothing else)."""
    ck = torch.load(path, map_location=map_location, weights_only=True)
    net.load_state_dict(ck["model_state_dict"])
    if optimizer is not None:
        if "optimizer_state_dict" in ck:
            optimizer.load_state_dict(ck["optimizer_state_dict"])
            log.info("Optimizer state loaded.")
        else:
            log.warning("Optimizer state not found in checkpoint, starting optimizer from scratch.")
    if scheduler is not None:
        if "scheduler_state_dict" in ck:
            scheduler.load_state_dict(ck["scheduler_state_dict"])
            log.info("Scheduler state loaded.")
        else:
            log.warning("Scheduler state not found in checkpoint. Scheduler will start without loaded state.")
    if "epoch" in ck:
        start_epoch = int(ck["epoch"]) + 1
    else:
        log.warning("Epoch number not found in checkpoint, starting from epoch 1.")
        start_epoch = 1
    if "best_acc" in ck:
        best_acc = float(ck["best_acc"])
    else:
        log.warning("Best accuracy not found in checkpoint, starting best_acc from 0.0.")
        best_acc = 0.0
    return start_epoch, best_acc, ck


# ---- data (outside the hot path: kept minimal) ------------------------------------------------------------------------
class SyntheticBatches:
    """`steps` batches of N(0,1) images (the value range of Normalize(0.5, 0.5), train.py:103) and uniform labels, generated
    once on the device — the metric's synthetic workload."""

    def __init__(self, steps, batch_size, num_classes, res, device, seed=0, distinct=2):
        g = torch.Generator(device=device).manual_seed(1234 + seed)
        self.items = [(torch.randn(batch_size, 3, res, res, device=device, generator=g),
                       torch.randint(0, num_classes, (batch_size,), device=device, generator=g)) for _ in range(max(1, distinct))]
        self.steps, self.batch_size = steps, batch_size

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            yield self.items[i % len(self.items)]

    @property
    def num_samples(self):
        return self.steps * self.batch_size


class NpzBatches:
    """`{split}_images.npy` / `{split}_labels.npy` (datasets.py:7-54 of the reference): uint8 images (N,H,W) or (N,H,W,3) ->
    float RGB in [-1, 1] at res x res (Resize + ToTensor + Normalize(0.5, 0.5), train.py:100-110), batched in order or
    shuffled per epoch.  Arrays are memory-mapped; a batch is converted on the fly.

    Under torchrun every rank passes its `rank` / `world`: ONE permutation per epoch (seeded by seed + epoch, identical on all
    ranks) is dealt out round-robin — rank r takes order[r::world] — so an epoch is one pass over the set, split into disjoint
    shards.  Training shards are padded (by wrapping around) to equal length, because every rank must run the same number of
    steps for the gradient all-reduce; evaluation shards are NOT padded (every sample is counted exactly once)."""

    def __init__(self, root_dir, split, batch_size, res, device, shuffle, seed=0, rank=0, world=1, pad_to_equal=None):
        self.images = np.load(os.path.join(root_dir, f"{split}_images.npy"), mmap_mode="r")
        self.labels = np.load(os.path.join(root_dir, f"{split}_labels.npy")).reshape(-1).astype(np.int64)
        self.batch_size, self.res, self.device, self.shuffle = batch_size, res, device, shuffle
        self.seed, self.epoch = seed, 0
        self.rank, self.world = int(rank), max(1, int(world))
        assert 0 <= self.rank < self.world
        self.pad_to_equal = shuffle if pad_to_equal is None else pad_to_equal
        self.classes = sorted(int(c) for c in np.unique(self.labels))

    @property
    def num_samples(self):
        """Samples this rank iterates over per epoch."""
        return len(self.shard_indices(self.epoch))

    def shard_indices(self, epoch):
        n = len(self.labels)
        order = np.random.default_rng(self.seed + epoch).permutation(n) if self.shuffle else np.arange(n)
        if self.world == 1:
            return order
        if self.pad_to_equal and n % self.world:
            order = np.concatenate([order, order[:self.world - n % self.world]])
        return order[self.rank::self.world]

    def __len__(self):
        per = (len(self.labels) + self.world - 1) // self.world if self.pad_to_equal else len(self.shard_indices(0))
        return (per + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = self.shard_indices(self.epoch)
        self.epoch += 1
        for i in range(0, len(order), self.batch_size):
            idx = np.sort(order[i:i + self.batch_size])
            x = torch.from_numpy(np.ascontiguousarray(self.images[idx])).to(self.device).float().div_(255.0)
            x = x.unsqueeze(1).expand(-1, 3, -1, -1) if x.dim() == 3 else x.permute(0, 3, 1, 2)
            if x.shape[-2:] != (self.res, self.res):
                x = torch.nn.functional.interpolate(x, size=(self.res, self.res), mode="bilinear", align_corners=False)
            yield x.sub(0.5).div(0.5).contiguous(), torch.from_numpy(self.labels[idx]).to(self.device)


def is_npz_dir(path, split):
    return (os.path.exists(os.path.join(path, f"{split}_images.npy")) and os.path.exists(os.path.join(path, f"{split}_labels.npy")))


# ---- loops ------------------------------------------------------------------------------------------------------------
def train_one_epoch(net, batches, optimizer, loss_fn, sync=None, on_step=None):
    """train.py:271-288.  Returns the mean loss (one device->host copy per epoch, not two per step)."""
    net.train()
    total = None
    n = 0
    for images, labels in batches:
        optimizer.zero_grad(set_to_none=True)
        loss = loss_fn(net(images), labels)
        loss.backward()
        # drop the autograd graph NOW: a `loss` that lives into the next iteration keeps this step's AccumulateGrad nodes alive, the
        # next forward then reuses them with the stream they were created on, and the engine serialises the block's two streams at
        # every parameter of the conv branch ("AccumulateGrad node's stream does not match ...": 48 instead of 30 ms per step)
        loss = loss.detach()
        if sync is not None:
            sync()
        optimizer.step()
        total = loss if total is None else total + loss
        n += 1
        if on_step is not None:
            on_step(n, loss)
    return float(total) / max(1, n) if total is not None else 0.0


@torch.no_grad()
def evaluate(net, batches, return_counts=False):
    """Validation accuracy, train.py:293-304 (correct predictions / number of samples).  return_counts: (correct, seen) of this
    process instead — under torchrun every rank evaluates its own shard and the COUNTS are summed over the ranks (fit())."""
    net.eval()
    correct, seen = None, 0
    for images, labels in batches:
        c = torch.eq(net(images).argmax(dim=1), labels).sum()
        correct = c if correct is None else correct + c
        seen += int(labels.numel())
    if return_counts:
        return (float(correct) if correct is not None else 0.0), seen
    return (float(correct) / seen) if seen else 0.0


def sharded_accuracy(correct, seen, device="cpu"):
    """Accuracy over the whole validation set when every rank evaluated its own shard: the (correct, seen) COUNTS are summed
    over the ranks (an average of per-rank accuracies would weight unequal shards wrongly and, with unsharded readers, count
    every sample world-size times)."""
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        t = torch.tensor([correct, seen], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t)
        correct, seen = float(t[0]), float(t[1])
    return correct / seen if seen else 0.0


def fit(net, train_batches, val_batches, optimizer, scheduler, *, epochs, start_epoch=1, best_acc=0.0, num_classes, class_indices,
        save_dir=".", model_name="Medmamba", patience=25, use_early_stopping=False, sync=None, is_main=True):
    """Epoch loop with the reference's checkpoint policy (train.py:271-369).  Returns (final_epoch, best_acc, paths)."""
    loss_fn = nn.CrossEntropyLoss()
    best_path, stale, final_epoch = None, 0, start_epoch - 1
    paths = {"best": None, "last": None}
    for epoch in range(start_epoch, epochs + 1):
        final_epoch = epoch
        mean_loss = train_one_epoch(net, train_batches, optimizer, loss_fn, sync)
        if scheduler is not None:
            scheduler.step()
        acc = sharded_accuracy(*evaluate(net, val_batches, return_counts=True), device=next(net.parameters()).device)
        log.info("[Epoch %d/%d] Train Loss: %.3f | Val Accuracy: %.3f", epoch, epochs, mean_loss, acc)
        if acc > best_acc:
            best_acc, stale = acc, 0
            if is_main:
                new_best = os.path.join(save_dir, f"{model_name}_epoch_{epoch}_best.pth")
                save_checkpoint(new_best, epoch=epoch, net=net, optimizer=optimizer, scheduler=scheduler, best_acc=best_acc,
                                num_classes=num_classes, class_indices=class_indices)
                if best_path and os.path.exists(best_path) and best_path != new_best:
                    os.remove(best_path)                       # train.py:333-337: only the newest best is kept
                best_path = paths["best"] = new_best
        else:
            stale += 1
            log.info("Validation accuracy did not improve. Patience: %d/%d", stale, patience)
        if use_early_stopping and stale >= patience:
            log.info("Early stopping triggered after %d epochs without improvement at epoch %d/%d.", patience, epoch, epochs)
            break
    if is_main:
        paths["last"] = save_checkpoint(os.path.join(save_dir, f"{model_name}_epoch_{final_epoch}_last.pth"), epoch=final_epoch,
                                        net=net, optimizer=optimizer, scheduler=scheduler, best_acc=best_acc,
                                        num_classes=num_classes, class_indices=class_indices)
    return final_epoch, best_acc, paths


def write_class_indices(save_dir, class_indices):
    """train.py:142-146."""
    with open(os.path.join(save_dir, "class_indices.json"), "w") as f:
        json.dump(class_indices, f, indent=4)
