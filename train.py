#!/usr/bin/env python3
"""Flag-compatible trainer for MedMamba on MI355X (reference: train.py:38-55 for the flags, :57-369 for the flow).

    python train.py --train_dir D --val_dir D [--medmb_size S] [--resume ckpt.pth] ...          # the reference's CLI
    python train.py --synthetic --steps 20 --epochs 2 --medmb_size T --num_classes 6            # no files: synthetic batches
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...     # one process per GPU (RCCL)

Same flags, per-dataset defaults, optimizer / scheduler, checkpoint dict and file names as the reference (see
medmamba_amd/trainer.py); checkpoints written by either trainer resume in the other.  Extra flags: --synthetic, --steps,
--val_steps, --res, --drop_path_rate.
"""
import argparse
import logging
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="Train a MedMamba model on MI355X.")
    ap.add_argument("--medmb_size", type=str, default="T", choices=["T", "S", "B", "Te"])
    ap.add_argument("--train_dir", type=str, default=None, help="training set: folder with train_images.npy / train_labels.npy, or an ImageFolder tree")
    ap.add_argument("--val_dir", type=str, default=None, help="validation set (val_images.npy / val_labels.npy, or an ImageFolder tree)")
    ap.add_argument("--num_classes", type=int, default=None)
    ap.add_argument("--model_name", type=str, default="Medmamba")
    ap.add_argument("--batch_size", type=int, default=None, help="per process (GPU)")
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--lr", type=float, default=None)
    ap.add_argument("--resume", type=str, default=None)
    ap.add_argument("--patience", type=int, default=25)
    ap.add_argument("--save_dir", type=str, default=".")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--augmentation", action="store_true", default=False)
    ap.add_argument("--use_early_stopping", action="store_true", default=False)
    ap.add_argument("--attn_drop_rate", type=float, default=0.0)
    # ours
    ap.add_argument("--synthetic", action="store_true", help="synthetic batches resident in HBM instead of a dataset")
    ap.add_argument("--steps", type=int, default=20, help="--synthetic: training steps per epoch")
    ap.add_argument("--val_steps", type=int, default=2, help="--synthetic: validation batches per epoch")
    ap.add_argument("--res", type=int, default=224)
    ap.add_argument("--drop_path_rate", type=float, default=0.1)
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    from medmamba_amd import trainer as T
    from medmamba_amd.ddp import GradSync, init_distributed

    if not torch.cuda.is_available():
        raise SystemExit("train.py needs a HIP device (the MedMamba hot path has no CPU implementation)")
    world = init_distributed()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    T.set_seed(args.seed)
    if os.environ.get("MM_TUNED_GEMMS", "1") == "1":
        from medmamba_amd.tuning import enable_tuned_gemms
        enable_tuned_gemms()        # recorded GEMM kernels for the shapes in the table (64 x 224^2 T/S, 32 x 384^2 B); others: heuristics
    os.makedirs(args.save_dir, exist_ok=True)

    if args.synthetic:
        is_npz = False
        epochs, batch_size, lr, decay = T.dataset_defaults(False, args.epochs, args.batch_size, args.lr)
        num_classes = args.num_classes if args.num_classes is not None else 6
        class_indices = {str(i): f"class_{i}" for i in range(num_classes)}
        train_batches = T.SyntheticBatches(args.steps, batch_size, num_classes, args.res, device, seed=rank)
        val_batches = T.SyntheticBatches(args.val_steps, batch_size, num_classes, args.res, device, seed=1000 + rank)
    else:
        if not args.train_dir or not args.val_dir:
            raise SystemExit("--train_dir and --val_dir are required (or use --synthetic)")
        is_npz = T.is_npz_dir(args.train_dir, "train")
        epochs, batch_size, lr, decay = T.dataset_defaults(is_npz, args.epochs, args.batch_size, args.lr)
        if not is_npz or not T.is_npz_dir(args.val_dir, "val"):
            raise SystemExit("ImageFolder datasets need torchvision, which is not part of this build; convert to "
                             "{split}_images.npy / {split}_labels.npy (the reference's NPZ layout) or use --synthetic")
        if args.augmentation:
            logging.warning("--augmentation is accepted for flag compatibility; the NPZ reader applies Resize + Normalize only")
        # one pass over the set per epoch, split over the ranks (disjoint shards of one shared permutation per epoch)
        train_batches = T.NpzBatches(args.train_dir, "train", batch_size, args.res, device, shuffle=True, seed=args.seed, rank=rank,
                                     world=world)
        val_batches = T.NpzBatches(args.val_dir, "val", batch_size, args.res, device, shuffle=False, rank=rank, world=world)
        num_classes = args.num_classes if args.num_classes is not None else len(train_batches.classes)
        class_indices = {str(c): str(c) for c in train_batches.classes}
    if rank == 0:
        T.write_class_indices(args.save_dir, class_indices)
    logging.info("Epochs: %d, Batch Size: %d per GPU x %d GPU(s), Initial LR: %g, model %s, %d classes", epochs, batch_size, world, lr,
                 args.medmb_size, num_classes)

    net = T.build_model(args.medmb_size, num_classes, args.attn_drop_rate, drop_path_rate=args.drop_path_rate).to(device)
    if world > 1:
        T.offset_device_rng(rank, args.seed)      # identical weights above, per-rank DropPath / Dropout draws from here on
    optimizer, scheduler = T.make_optimizer(net, is_npz, lr, decay)
    start_epoch, best_acc = 1, 0.0
    if args.resume:
        if os.path.isfile(args.resume):
            start_epoch, best_acc, _ = T.load_checkpoint(args.resume, net, optimizer, scheduler, map_location=device)
            logging.info("Resuming training from epoch %d (best accuracy so far %.3f)", start_epoch, best_acc)
        else:
            logging.error("Checkpoint file not found: %s. Starting training from scratch.", args.resume)
    if epochs < start_epoch:
        logging.warning("Target epochs (%d) is less than start epoch (%d). No training will occur.", epochs, start_epoch)
        return 0
    sync = GradSync(net) if world > 1 else None          # broadcasts rank 0's (possibly resumed) weights once
    final_epoch, best_acc, paths = T.fit(net, train_batches, val_batches, optimizer, scheduler, epochs=epochs, start_epoch=start_epoch,
                                         best_acc=best_acc, num_classes=num_classes, class_indices=class_indices,
                                         save_dir=args.save_dir, model_name=args.model_name, patience=args.patience,
                                         use_early_stopping=args.use_early_stopping, sync=sync, is_main=rank == 0)
    logging.info("Finished Training. Final Epoch Reached: %d. Best validation accuracy: %.3f (last: %s)", final_epoch, best_acc,
                 paths["last"])
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
