"""GPU: the flag-compatible trainer end to end on synthetic batches — checkpoint files as the reference names them, and a
resumed run continues like the uninterrupted one (reference flow: train.py:208-260, 271-362)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(tmp, name, epochs, resume=None):
    import train
    save = os.path.join(tmp, name)
    argv = ["--synthetic", "--steps", "3", "--val_steps", "1", "--epochs", str(epochs), "--medmb_size", "T", "--res", "64",
            "--batch_size", "4", "--num_classes", "5", "--drop_path_rate", "0", "--save_dir", save, "--seed", "7"]
    if resume:
        argv += ["--resume", resume]
    assert train.main(argv) == 0
    return save


def test_synthetic_training_checkpoints_and_resume(tmp_path):
    tmp = str(tmp_path)
    a = _run(tmp, "a", 2)                                           # two epochs in one go
    b1 = _run(tmp, "b", 1)                                          # one epoch ...
    last1 = os.path.join(b1, "Medmamba_epoch_1_last.pth")
    assert os.path.isfile(last1) and os.path.isfile(os.path.join(b1, "class_indices.json"))
    _run(tmp, "b", 2, resume=last1)                                 # ... resumed for the second
    ca = torch.load(os.path.join(a, "Medmamba_epoch_2_last.pth"), weights_only=True)
    cb = torch.load(os.path.join(b1, "Medmamba_epoch_2_last.pth"), weights_only=True)
    assert set(ca) == {"epoch", "model_state_dict", "optimizer_state_dict", "best_acc", "num_classes", "class_indices"}
    assert ca["epoch"] == cb["epoch"] == 2 and ca["num_classes"] == 5
    # same weights up to what the run-to-run noise of the fp32 atomics in the backward scan (1e-7 relative on a gradient)
    # can do through AdamW: its update is lr * m / sqrt(v), so noise on a near-zero gradient moves a weight by a fraction of
    # lr = 1e-4 per step regardless of the gradient's size.  Worst case: a conv bias in front of a training-mode BatchNorm
    # has a true gradient of exactly zero, so its updates are +-lr of pure noise in both runs -> up to 2 * lr per step apart
    # (observed 2.5e-4 after the 3 resumed steps; weights with real gradients: 1e-5)
    lr, resumed_steps = 1e-4, 3
    for k, va in ca["model_state_dict"].items():
        vb = cb["model_state_dict"][k]
        if va.dtype.is_floating_point:
            # BatchNorm running statistics follow the activations, which follow every noisy weight before them: relative 1e-3
            rel = 1e-3 if "running_" in k else 1e-4
            assert float((va - vb).abs().max()) <= 2 * lr * resumed_steps + rel * float(va.abs().max()), k
        else:
            assert torch.equal(va, vb), k                            # BatchNorm step counters
    # ... and that is the exception: on average a weight of the first block ends within a few 1e-6 of the uninterrupted run (single
    # elements whose gradient is at the noise level take Adam's +-lr steps in either direction, hence no tight bound on the max)
    w = "layers.0.blocks.0.self_attention.in_proj.weight"
    assert float((ca["model_state_dict"][w] - cb["model_state_dict"][w]).abs().mean()) <= 1e-5
    sa, sb = ca["optimizer_state_dict"]["state"], cb["optimizer_state_dict"]["state"]
    assert len(sa) == len(sb) and all(float(sa[i]["step"]) == float(sb[i]["step"]) == 6.0 for i in sa)
    bests = [f for f in os.listdir(a) if f.endswith("_best.pth")]
    assert len(bests) <= 1                                           # only the newest best checkpoint is kept (train.py:333-337)
