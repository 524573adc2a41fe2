"""CPU: checkpoint format / resume logic / defaults of medmamba_amd.trainer against the reference trainer's contract
(train.py:71-85 defaults, :187-201 optimizer, :208-260 resume fallbacks, :310-362 checkpoint dict and file names).
No forward pass runs here (the hot path is HIP-only); gradients are synthetic."""
import os

import numpy as np
import pytest
import torch

from medmamba_amd import trainer as T


def _tiny(seed=0):
    torch.manual_seed(seed)
    return T.VSSM(num_classes=3, depths=[1, 1], dims=[16, 32], drop_path_rate=0.0)


def _fake_steps(net, opt, sched=None, n=2, seed=1):
    g = torch.Generator().manual_seed(seed)
    for _ in range(n):
        for p in net.parameters():
            p.grad = torch.randn(p.shape, generator=g) * 1e-2
        opt.step()
    if sched is not None:
        sched.step()


def test_defaults_and_optimizer_match_the_reference_contract():
    assert T.dataset_defaults(True) == (100, 100, 1e-3, [50, 75])
    assert T.dataset_defaults(False) == (150, 64, 1e-4, [])
    assert T.dataset_defaults(False, epochs=3, batch_size=8, lr=0.5)[:3] == (3, 8, 0.5)
    net = _tiny()
    opt, sched = T.make_optimizer(net, True, 1e-3, [50, 75], fused=False)
    assert isinstance(opt, torch.optim.AdamW) and opt.defaults["weight_decay"] == 1e-2 and sched is not None
    assert list(sched.milestones) == [50, 75] and sched.gamma == 0.1
    opt, sched = T.make_optimizer(net, False, 1e-4, [], fused=False)
    assert opt.defaults["weight_decay"] == 1e-4 and opt.defaults["betas"] == (0.9, 0.999) and sched is None


def test_checkpoint_round_trip(tmp_path):
    net = _tiny()
    opt, sched = T.make_optimizer(net, True, 1e-3, [50, 75], fused=False)
    _fake_steps(net, opt, sched)
    path = T.save_checkpoint(str(tmp_path / "Medmamba_epoch_7_best.pth"), epoch=7, net=net, optimizer=opt, scheduler=sched, best_acc=0.625,
                             num_classes=3, class_indices={"0": "a", "1": "b", "2": "c"})
    ck = torch.load(path, weights_only=True)
    assert set(ck) == set(T.CHECKPOINT_KEYS) | {"scheduler_state_dict"}          # exactly the reference's keys (train.py:310-321)
    net2 = _tiny(seed=5)
    opt2, sched2 = T.make_optimizer(net2, True, 1e-3, [50, 75], fused=False)
    start, best, _ = T.load_checkpoint(path, net2, opt2, sched2)
    assert start == 8 and best == 0.625
    for (k, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), k
    sa, sb = opt.state_dict(), opt2.state_dict()
    assert sa["param_groups"] == sb["param_groups"]
    for i in sa["state"]:
        for k in sa["state"][i]:
            assert torch.equal(torch.as_tensor(sa["state"][i][k]), torch.as_tensor(sb["state"][i][k])), (i, k)
    assert sched2.state_dict() == sched.state_dict()
    # both replicas continue identically
    _fake_steps(net, opt, seed=9)
    _fake_steps(net2, opt2, seed=9)
    for (k, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), k


def test_loads_a_dict_with_exactly_the_reference_keys_and_falls_back(tmp_path):
    """A checkpoint as the reference writes it (int class-index keys of the ImageFolder branch, train.py:128; no scheduler
    entry) and stripped-down ones exercising every fallback of train.py:215-250."""
    net = _tiny()
    opt, _ = T.make_optimizer(net, False, 1e-4, [], fused=False)
    _fake_steps(net, opt)
    ref = {"epoch": 3, "model_state_dict": net.state_dict(), "optimizer_state_dict": opt.state_dict(), "best_acc": 0.5,
           "num_classes": 3, "class_indices": {0: "akiec", 1: "bcc", 2: "mel"}}
    torch.save(ref, tmp_path / "ref.pth")
    net2 = _tiny(seed=3)
    opt2, sched2 = T.make_optimizer(net2, True, 1e-3, [50, 75], fused=False)     # a scheduler exists, the file has no state for it
    start, best, ck = T.load_checkpoint(str(tmp_path / "ref.pth"), net2, opt2, sched2)
    assert (start, best) == (4, 0.5) and ck["class_indices"][1] == "bcc" and ck["num_classes"] == 3
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), net2.state_dict().values()))
    torch.save({"model_state_dict": net.state_dict()}, tmp_path / "bare.pth")          # weights only
    net3 = _tiny(seed=4)
    opt3, _ = T.make_optimizer(net3, False, 1e-4, [], fused=False)
    assert T.load_checkpoint(str(tmp_path / "bare.pth"), net3, opt3, None)[:2] == (1, 0.0)
    assert len(opt3.state_dict()["state"]) == 0
    with pytest.raises(KeyError):
        torch.save({"epoch": 1}, tmp_path / "nomodel.pth")
        T.load_checkpoint(str(tmp_path / "nomodel.pth"), net3)


def test_state_dict_is_interchangeable_with_the_reference_layout():
    """Key names / shapes of a reference T model (355 entries) — what `net.load_state_dict(checkpoint[...])`
    of test.py:76-77 relies on."""
    net = T.build_model("T", 6)
    sd = net.state_dict()
    assert len(sd) == 355
    assert sd["layers.0.blocks.0.self_attention.x_proj_weight"].shape == (4, 35, 96)
    assert sd["layers.0.blocks.0.self_attention.dt_projs_weight"].shape == (4, 96, 3)
    assert sd["layers.3.blocks.1.conv33conv33conv11.1.weight"].shape == (384, 384, 3, 3)
    assert sd["layers.0.downsample.reduction.weight"].shape == (192, 384)


def test_npz_batches(tmp_path):
    rng = np.random.default_rng(0)
    np.save(tmp_path / "train_images.npy", rng.integers(0, 256, (10, 28, 28), dtype=np.uint8))
    np.save(tmp_path / "train_labels.npy", rng.integers(0, 3, (10, 1)))
    assert T.is_npz_dir(str(tmp_path), "train") and not T.is_npz_dir(str(tmp_path), "val")
    b = T.NpzBatches(str(tmp_path), "train", 4, 32, torch.device("cpu"), shuffle=True, seed=1)
    seen = 0
    for x, y in b:
        assert x.shape[1:] == (3, 32, 32) and x.dtype == torch.float32 and float(x.min()) >= -1.0 and float(x.max()) <= 1.0
        assert torch.equal(x[:, 0], x[:, 1]) and y.dtype == torch.int64
        seen += len(y)
    assert seen == 10 and len(b) == 3 and b.num_samples == 10


def test_cli_flags_match_the_reference(monkeypatch):
    import train
    a = train.parse_args(["--train_dir", "x", "--val_dir", "y", "--medmb_size", "B", "--resume", "c.pth", "--use_early_stopping",
                          "--augmentation", "--attn_drop_rate", "0.1", "--patience", "3", "--model_name", "M", "--save_dir", "s",
                          "--seed", "1", "--num_classes", "4", "--batch_size", "2", "--epochs", "5", "--lr", "0.01"])
    assert (a.medmb_size, a.resume, a.use_early_stopping, a.augmentation, a.attn_drop_rate, a.patience) == ("B", "c.pth", True, True, 0.1, 3)
    assert train.parse_args([]).medmb_size == "T" and train.parse_args([]).seed == 42 and train.parse_args([]).patience == 25


def test_recorded_miopen_database_seeds_a_directory_once(tmp_path):
    """tuning.seed_miopen_db (what bench.py starts its private MIOpen database from): every recorded file arrives whole under its own
    name (MIOpen matches GPU and build by file name), nothing is overwritten, a second call — or a second rank — is a no-op, and
    every line of the find database has the `problem=solver:time,workspace,algorithm;...` shape MIOpen parses."""
    import threading
    from medmamba_amd import tuning
    names = sorted(f for f in os.listdir(tuning.MIOPEN_DB_DIR) if f.endswith(".txt"))
    assert any(f.endswith(".ufdb.txt") for f in names) and any(f.endswith(".udb.txt") for f in names)
    assert all(f.startswith("gfx950") for f in names)
    d = str(tmp_path / "db")
    counts = []
    ts = [threading.Thread(target=lambda: counts.append(tuning.seed_miopen_db(d))) for _ in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert sorted(os.listdir(d)) == names and sum(counts) >= len(names)
    for f in names:
        assert open(os.path.join(d, f), "rb").read() == open(os.path.join(tuning.MIOPEN_DB_DIR, f), "rb").read()
    assert tuning.seed_miopen_db(d) == 0
    mine = os.path.join(d, names[0])
    open(mine, "w").write("kept")
    assert tuning.seed_miopen_db(d) == 0 and open(mine).read() == "kept"
    ufdb = [f for f in names if f.endswith(".ufdb.txt")][0]
    keys = set()
    for line in open(os.path.join(tuning.MIOPEN_DB_DIR, ufdb)):
        key, val = line.rstrip("\n").split("=", 1)
        assert key.endswith(("-F", "-B", "-W")) and "NCHW-FP32" in key
        for ent in val.split(";"):
            solver, rec = ent.split(":")
            t_ms, ws, algo = rec.split(",")
            assert float(t_ms) > 0 and int(ws) >= 0 and algo.startswith("miopenConvolution")
        keys.add(key)
    # the dense 3x3 convolutions of S at 64 images (BASELINE config 3), forward / data / weight gradient at all four stages
    for c, hw in ((48, 56), (96, 28), (192, 14), (384, 7)):
        for d_ in "FBW":
            assert f"{c}-{hw}-{hw}-3x3-{c}-{hw}-{hw}-64-1x1-1x1-1x1-0-NCHW-FP32-{d_}" in keys
