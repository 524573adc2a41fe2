#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json: "images/sec fwd+bwd MedMamba-S 224^2 @1/2/4/8 MI355X; SS2D scan
HBM GB/s vs roofline").

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one full training step of MedMamba-S (train.py:277-288: zero_grad, forward, CrossEntropy,
backward, AdamW step) on a synthetic 224x224x3 batch of 64 images per GPU that is already resident in HBM
(BASELINE config 3; N > 1 = config 4).  Weak scaling: per-GPU batch fixed, one process per GPU, gradients averaged by
ONE flat all-reduce after backward (medmamba_amd.ddp.GradSync) over RCCL (backend "nccl"); MM_DDP=torch selects
DistributedDataParallel instead.  Rank 0 prints ONE JSON line.

Other single-GPU configurations of BASELINE.json (metric / config.workload follow the flags):
    --mode fwd --batch 32            config 2: MedMamba-S, 32 images, forward only (eval, no_grad)
    --size B --res 384 --batch 32    config 5: MedMamba-B at 384x384 (L = 9216 at stage 1)

Extra objects on the line:
  roofline     — the selective-scan forward kernel (the north-star kernel): algorithmic bytes
                 of every forward scan call in the timed steps / their hipEvent-measured duration, vs 8 TB/s.
                 In the timed steps the conv branch of every block runs beside the scan on a second HIP stream, so
                 `achieved` is the kernel's rate while sharing the GPU; `achieved_alone` / `frac_alone` repeat the
                 measurement in 3 extra untimed steps with that overlap switched off (N=1 only).  `frac` / `frac_alone`
                 are the raw event times; an event pair adds its own ~5 us to every bracketed launch (measured live as
                 `event_bracket_us`, checked against rocprofv3's kernel durations in profiles/r4_event_bracket_vs_kernel_trace.txt):
                 `frac_net` / `frac_alone_net` have it removed and are what a kernel trace of the same launches shows.
  roofline_bwd — same for the backward scan kernel.
  cpu_baseline — the CPU restatement of the reference path (oracle/: torch-CPU glue + C selective_scan_ref)
                 timed on this box's host cores on a bounded sample (rank 0, N=1 only).
  cpu_baseline_ref_loop — the reference's literal CPU path (north_star: "selective_scan_ref CPU path", temp.py:57-139):
                 the pure-PyTorch time loop inside the same model, MedMamba-T, 1 x 3 x 224 x 224, forward
                 (BASELINE config 1), 1 warm-up + median of 3.
"""
import argparse
import json
import os
import sys
import time

# this pool's host driver supports only dmabuf IPC: without this RCCL / cross-process HIP memory sharing fails with
# "hipIpcGetMemHandle: invalid argument".  Already exported on the boxes; kept as a default for any other launcher.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# MIOpen keeps what its solver searches found in a per-user database and later processes inherit it — including the quick-search
# picks of whatever ran before on the box (a test suite, a profiling run), which this process's own full search (cudnn.benchmark,
# below) would then not replace: 31 instead of 28 ms per step, measured.  The benchmark searches for itself, in a database of its own.
_PRIVATE_MIOPEN_DB = os.environ.get("MM_MIOPEN_BENCHMARK", "1") == "1" and "MIOPEN_USER_DB_PATH" not in os.environ
if _PRIVATE_MIOPEN_DB:
    import tempfile
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # one database per launch, shared by its ranks (same parent = the launcher): rank 0 searches first (set-up passes below),
        # the others then find its results — the same solvers on every rank instead of N independent searches with N outcomes
        _db = os.path.join(tempfile.gettempdir(), f"mm_miopen_db_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")
        os.makedirs(_db, exist_ok=True)
        os.environ["MIOPEN_USER_DB_PATH"] = _db
    else:
        os.environ["MIOPEN_USER_DB_PATH"] = tempfile.mkdtemp(prefix="mm_miopen_db_")

import torch
import torch.distributed as dist
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# ... and that database starts from the recorded search of medmamba_amd/tuning/miopen_gfx950/ (the conv-side counterpart of the GEMM
# table: same solvers on every box, no 70 s search for the recorded shapes; MIOpen ignores it when its build differs and searches as
# before; MM_MIOPEN_SEED_DB=0 starts from an empty database).  Before the first convolution: MIOpen reads the directory once.
MIOPEN_DB_SEEDED = 0
if _PRIVATE_MIOPEN_DB and os.environ.get("MM_MIOPEN_SEED_DB", "1") == "1":
    from medmamba_amd.tuning import seed_miopen_db
    MIOPEN_DB_SEEDED = seed_miopen_db(os.environ["MIOPEN_USER_DB_PATH"])

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md §Chip-level parameters)
# Secondary ceiling of the scan kernels ("fp32 VALU / v_exp_f32 issue"): the irreducible VALU work per state-step —
# forward v_mul, v_exp_f32, v_mul, v_fma, v_fma = 5 instructions; backward: recompute 4 + adjoint 10 + the second v_exp_f32 = 15,
# counting a v_exp_f32 once — at the issue rate tools/ubench/valu_rate.cpp measures for that mix on a saturated SIMD (7.5 ns per
# wavefront-state-step of 5 instructions), over the chip's 1024 SIMDs.  valu_floor_frac = that time / measured time.
VALU_FLOOR_NS_PER_WAVE_STEP = {"scan_fwd": 7.5, "scan_bwd": 7.5 * 15 / 5}
N_SIMD = 1024


def cpu_baseline(size, res, nimg=32, mode="train"):
    """fwd+bwd (mode "fwd": forward only) of the same model through the CPU oracle (kind "port"): bounded sample of `nimg`
    images (fewer at 384x384, where an image costs ~6x the work)."""
    if res > 256:
        nimg = max(4, nimg // 4)
    from oracle import model_ref as R
    from oracle.scan_ref import c_selective_scan_fn, build_c_oracle
    from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
    build_c_oracle()
    # the box's CPU share for one GPU is 16 cores; more threads than that only oversubscribes (256-CPU hosts)
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    torch.set_num_threads(cores)
    from oracle.scan_ref import _lib as _olib
    _olib().oracle_set_threads(cores)
    torch.manual_seed(42)
    cfg = MEDMAMBA_CONFIGS[size]
    net = VSSM(num_classes=6, **cfg)          # CPU copy, only used as a parameter container
    p = {k: (v.detach().clone().requires_grad_() if v.dtype.is_floating_point and "running" not in k else v.clone())
         for k, v in net.state_dict().items()}
    x = torch.randn(nimg, 3, res, res)
    y = torch.randint(0, 6, (nimg,))

    def step():
        if mode == "fwd":
            with torch.no_grad():
                return float(nn.functional.cross_entropy(R.vssm_forward(p, x, cfg["depths"], c_selective_scan_fn, training=False), y))
        for v in p.values():
            if v.requires_grad:
                v.grad = None
        loss = nn.functional.cross_entropy(R.vssm_forward(p, x, cfg["depths"], c_selective_scan_fn, training=True), y)
        loss.backward()
        return float(loss.detach())

    step()                                     # warm-up (first call is ~3x slower: allocator growth)
    t0 = time.perf_counter()
    step()
    dt = time.perf_counter() - t0
    return dict(value=nimg / dt, unit="images/s", cores=cores, kind="port",
                sample=f"{nimg} images, MedMamba-{size} {res}x{res} {'fwd+bwd (no optimizer step)' if mode == 'train' else 'forward only'}, torch-CPU glue + "
                       f"oracle/selective_scan_ref.c scan on {cores} threads, 1 warm-up + 1 timed pass "
                       f"({dt:.1f} s)")


def cpu_baseline_ref_loop(reps=3):
    """BASELINE config 1 on the host: MedMamba-T, 1 x 3 x 224 x 224, forward, the pure-PyTorch selective_scan_ref loop
    (oracle.scan_ref.selective_scan_ref, restated from temp.py:57-139) as the scan — the reference's own CPU path."""
    from oracle import model_ref as R
    from oracle.scan_ref import selective_scan_ref
    from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    cfg = MEDMAMBA_CONFIGS["T"]
    p = {k: v.clone() for k, v in VSSM(num_classes=6, **cfg).state_dict().items()}
    x = torch.randn(1, 3, 224, 224)
    ts = []
    with torch.no_grad():
        for i in range(reps + 1):                       # first call is ~3x slower (allocator growth): warm-up
            t0 = time.perf_counter()
            R.vssm_forward(p, x, cfg["depths"], selective_scan_ref, training=False)
            ts.append(time.perf_counter() - t0)
    med = sorted(ts[1:])[len(ts[1:]) // 2]
    return dict(value=1.0 / med, unit="images/s", cores=cores, kind="port",
                sample=f"MedMamba-T 1x3x224x224 forward, pure-PyTorch selective_scan_ref time loop (temp.py:57-139 restated) "
                       f"on {cores} torch threads, 1 warm-up ({ts[0]:.1f} s) + median of {reps} ({med:.2f} s)")


def cpu_baseline_subprocess(size, res, timeout_s=240, fn="cpu_baseline", mode="train"):
    """Run the CPU leg in a child process (own thread pools, hard time bound); never blocks the GPU result."""
    import subprocess
    call = "bench.cpu_baseline(%r, %d, 32, %r)" % (size, res, mode) if fn == "cpu_baseline" else "bench.cpu_baseline_ref_loop()"
    code = ("import json,sys; sys.path.insert(0, %r); import bench; "
            "print('CPUBASE ' + json.dumps(%s))" % (ROOT, call))
    env = dict(os.environ, OMP_NUM_THREADS="16", MKL_NUM_THREADS="16", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s, env=env)
        for line in r.stdout.splitlines():
            if line.startswith("CPUBASE "):
                return json.loads(line[8:])
        return dict(value=None, unit="images/s", cores=0, kind="port", sample="cpu leg failed: " + r.stderr[-200:])
    except subprocess.TimeoutExpired:
        return dict(value=None, unit="images/s", cores=0, kind="port", sample=f"cpu leg exceeded {timeout_s} s")


def step_census(step):
    """One profiled step (torch.profiler, device activities only): kernel launches per step and which MIOpen solver family ran the
    dense convolutions on THIS box (MIOpen picks by on-the-spot timing; Winograd and implicit GEMM are a near tie at 14x14)."""
    import collections
    import re
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    fam = collections.Counter()
    launches = 0
    families = [(r"miopenSp3AsmConv|[Ww]inograd", "winograd"), (r"igemm_fwd", "igemm_fwd"), (r"igemm_bwd", "igemm_bwd"),
                (r"igemm_wrw", "igemm_wrw"), (r"naive_conv|Conv.*[Dd]irect|gcnAsmConv", "direct"), (r"batched_transpose", "layout_transpose"),
                (r"Im2[dD]?[cC]ol|im2col", "im2col")]
    for e in prof.key_averages():
        if e.device_type != DeviceType.CUDA or re.search(r"[Mm]emcpy|[Mm]emset", e.key):
            continue
        launches += e.count
        for pat, name in families:
            if re.search(pat, e.key):
                fam[name] += e.count
                break
    return {"launches_per_step": launches, "conv_kernel_families": dict(fam)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU (BASELINE config 3/4: 64)")
    ap.add_argument("--size", default="S", choices=["T", "S", "B", "Te"])
    ap.add_argument("--res", type=int, default=224)
    ap.add_argument("--mode", default="train", choices=["train", "fwd"], help="train: fwd + CE + bwd + AdamW (config 3/4/5); "
                    "fwd: eval-mode forward under no_grad (config 2)")
    ap.add_argument("--graph", action="store_true", help="--mode fwd only: replay the forward from one hipGraph "
                    "(medmamba_amd.graphs.GraphedInference) instead of ~550 eager launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse)")
    ap.add_argument("--no-alone-pass", action="store_true", help="skip the 3 untimed steps that re-measure the scan kernels "
                    "without the side-stream overlap (use under rocprofv3 so that the trace holds overlapped steps only)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  No device_id=: eager communicator binding costs 6 ms of host time per step
        # (tools/pg_overhead.py); the communicator comes up with the first collective (GradSync's parameter broadcast)
        dist.init_process_group(backend=args.backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS
    from medmamba_amd.selective_scan_interface import KERNEL_TIMER
    from medmamba_amd.tuning import enable_tuned_gemms
    if os.environ.get("MM_TUNED_GEMMS", "1") == "1":
        enable_tuned_gemms()                        # recorded rocBLAS / hipBLASLt solutions per GEMM shape; no tuning at run time
    from medmamba_amd.ddp import GradSync, wrap_ddp

    if os.environ.get("MM_MIOPEN_BENCHMARK", "1") == "1":
        # MIOpen times every applicable solver per conv shape at its first use (the first, single-stream warm-up step) instead of
        # taking its quick hybrid search's pick: on some boxes of the pool that pick is Winograd for the 7x7-stage convolutions,
        # 4x slower there than the implicit-GEMM kernels (31.4 vs 28.0 ms per step, same GPU); with the full search every box
        # lands on the same solvers.  Costs ~70 s of warm-up once per process.  MM_MIOPEN_BENCHMARK=0: off.
        torch.backends.cudnn.benchmark = True
    torch.manual_seed(42)                      # identical replicas; random-init weights (no checkpoints offline)
    net = VSSM(num_classes=6, **MEDMAMBA_CONFIGS[args.size]).to(dev)
    net = net.train() if args.mode == "train" else net.eval()
    if world > 1:
        from medmamba_amd.trainer import offset_device_rng
        offset_device_rng(rank, 42)             # identical weights above; DropPath masks drawn per rank from here on
    # gradient exchange: GradSync (bucketed, overlapped with backward) unless MM_DDP=torch asks for DistributedDataParallel
    use_torch_ddp = world > 1 and os.environ.get("MM_DDP", "bucketed") == "torch"
    model = wrap_ddp(net, dev) if use_torch_ddp else net
    # MM_DDP=flat: GradSync(overlap=False), one all-reduce after backward; default: per-stage buckets started during backward
    sync = GradSync(net, overlap=os.environ.get("MM_DDP", "bucketed") != "flat", timing=True) if (world > 1 and not use_torch_ddp) else None
    # train.py:189-192 (ImageFolder branch); fused=True: same update rule, one multi-tensor kernel per step
    # MM_FUSED_ADAMW=1 (default): optim.FusedAdamW = torch.optim.AdamW(fused=True) with its per-step tensor lists cached;
    # =torch: torch.optim.AdamW(fused=True) itself; =0: torch's default (foreach) implementation
    fa = os.environ.get("MM_FUSED_ADAMW", "1")
    if fa == "1":
        from medmamba_amd.optim import FusedAdamW
        opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, **({"fused": True} if fa == "torch" else {}))
    loss_fn = nn.CrossEntropyLoss()
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randn(args.batch, 3, args.res, args.res, device=dev, generator=g)   # resident in HBM
    labels = torch.randint(0, 6, (args.batch,), device=dev, generator=g)

    def train_step():
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(model(images), labels)
        loss.backward()
        if sync is not None:
            sync()
        opt.step()
        return loss.detach()            # do not keep the autograd graph (and its AccumulateGrad nodes) alive across steps

    graphed = None
    if args.graph:
        assert args.mode == "fwd" and world == 1, "--graph is for the single-GPU inference forward"
        from medmamba_amd.graphs import GraphedInference
        graphed = GraphedInference(net, images)

    def fwd_step():
        with torch.no_grad():
            return loss_fn(graphed(images) if graphed is not None else model(images), labels)

    step = train_step if args.mode == "train" else fwd_step

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[local_rank]) if args.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    # set-up, not warm-up: the first passes over the model run MIOpen's solver search per convolution shape (the forward and the
    # backward shapes, one stream first, then the two-stream schedule) — part of building the workload like the GEMM table above
    if torch.backends.cudnn.benchmark:
        def setup_pass():       # forward (+ backward): no collective, no optimizer step — the replicas stay identical
            if args.mode != "train":
                return fwd_step()
            import contextlib
            opt.zero_grad(set_to_none=True)
            with (sync.no_sync() if sync is not None else (model.no_sync() if use_torch_ddp else contextlib.nullcontext())):
                loss_fn(model(images), labels).backward()
            opt.zero_grad(set_to_none=True)
        for turn in range(2 if world > 1 else 1):       # N > 1: rank 0 first, then everybody else out of its database
            if world == 1 or (rank == 0) == (turn == 0):
                for _ in range(2):
                    setup_pass()
                torch.cuda.synchronize()
            if world > 1:
                fence()
    for _ in range(args.warmup):
        step()
    KERNEL_TIMER.enabled = True
    KERNEL_TIMER.records.clear()
    fence()
    if sync is not None:
        sync.allreduce_ms()         # drop the warm-up steps' collective windows: the figure below is per TIMED step
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    KERNEL_TIMER.enabled = False
    ks = KERNEL_TIMER.summary() if rank == 0 else {}
    ks_iso = {}
    if rank == 0 and world == 1 and not args.no_alone_pass:
        # untimed extra pass: the same kernels on the same shapes WITHOUT the conv branch running beside them on the
        # side stream (modules._TWO_STREAMS) — the kernel's own rate, next to the rate it gets inside the overlapped step
        from medmamba_amd import modules as _modules
        two = _modules._TWO_STREAMS
        _modules._TWO_STREAMS = False
        KERNEL_TIMER.records.clear()
        KERNEL_TIMER.enabled = True
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        KERNEL_TIMER.enabled = False
        _modules._TWO_STREAMS = two
        ks_iso = KERNEL_TIMER.summary()
    # what the event pair itself adds to a bracketed launch: the empty bracket behind a running kernel (profiles/
    # r4_event_bracket_vs_kernel_trace.txt: event time - kernel-trace duration = the empty bracket, 5.0-5.2 us, whatever the kernel)
    bracket_us = None
    if rank == 0:
        probe = torch.zeros(1 << 20, device=dev)
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
        for s_, e_ in pairs:
            probe.add_(1.0)
            s_.record()
            e_.record()
        torch.cuda.synchronize()
        bracket_us = 1e3 * sorted(s_.elapsed_time(e_) for s_, e_ in pairs)[len(pairs) // 2]
    dist_info = None
    if world > 1:
        # every rank's own wall time of the timed region and the window its gradient all-reduces were in flight (device events of
        # GradSync; with overlap this includes the backward work that ran meanwhile): a first multi-GPU run then says WHERE it loses
        ar_ms = sync.allreduce_ms() / args.steps if sync is not None else -1.0
        mine = torch.tensor([dt, ar_ms], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                     "exchange": "DistributedDataParallel" if use_torch_ddp else ("GradSync bucketed" if len(sync.buckets) > 1 else "GradSync flat"),
                     "per_rank_ms_per_step": [round(1e3 * float(t[0]) / args.steps, 3) for t in allr],
                     "allreduce_window_ms_per_step": [round(float(t[1]), 3) for t in allr]}
        if sync is not None:
            dist_info.update(buckets=sync.stats["buckets"], buckets_started_in_backward=sync.stats["early"],
                             gradient_MB=round(sum(p.numel() for p in sync.params) * 4 / 1e6, 1))
        dt = max(float(t[0]) for t in allr)
    assert torch.isfinite(loss).item(), "loss is not finite"

    # ---- untimed extras of the single-GPU training line: what one step launches, which MIOpen solver family
    # this box picked per convolution kernel, and the step in the reference's own mode (cudnn.deterministic, train.py:28-29)
    extras = {}
    if rank == 0 and world == 1 and args.mode == "train" and not args.no_alone_pass:
        try:
            extras.update(step_census(step))
        except Exception as e:  # noqa: BLE001 — diagnostics must never cost the bench line
            extras["launches_per_step"] = None
            extras["census_error"] = repr(e)[:200]
        try:
            prev = torch.backends.cudnn.deterministic
            torch.backends.cudnn.deterministic = True
            for _ in range(2):
                step()
            nd = max(3, args.steps // 2)
            fence()
            t0d = time.perf_counter()
            for _ in range(nd):
                step()
            fence()
            dtd = time.perf_counter() - t0d
            torch.backends.cudnn.deterministic = prev
            extras["deterministic"] = {"ms_per_step": round(1e3 * dtd / nd, 3), "value": round(args.batch * nd / dtd, 2), "steps": nd,
                                       "note": "torch.backends.cudnn.deterministic=True, the mode the reference trains in (train.py:28-29): "
                                               "the dense convs' weight gradient is an im2col + batched GEMM + ordered sum; every kernel "
                                               "of this library is deterministic in both modes"}
        except Exception as e:  # noqa: BLE001
            extras["deterministic"] = {"ms_per_step": None, "error": repr(e)[:200]}

    if rank == 0:
        try:      # PMC-derived HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/scan_traffic.json)
            traffic = json.load(open(os.path.join(ROOT, "profiles", "scan_traffic.json")))
        except Exception:
            traffic = {}
        std_workload = (args.size == "S" and args.batch == 64 and args.res == 224 and args.mode == "train")

        def roof(tag):
            d = ks.get(tag)
            if not d or d["ms"] <= 0:
                return None
            gbs = d["bytes"] / d["ms"] / 1e6
            tr = traffic.get(tag, {}).get("bytes_per_launch") if std_workload else None
            tr = None if tr is None else round(tr * d["calls"] / d["ms"] / 1e6, 1)      # GB/s, same basis as `achieved`
            r = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": tr, "kernel": tag,
                 "calls": d["calls"], "avg_us_per_call": round(1e3 * d["ms"] / d["calls"], 2),
                 "algorithmic_MB_per_call": round(d["bytes"] / d["calls"] / 1e6, 2)}
            floor_ms = d["state_steps"] / 64.0 * VALU_FLOOR_NS_PER_WAVE_STEP[tag] / N_SIMD * 1e-6
            r["valu_floor_frac"] = round(floor_ms / d["ms"], 4)
            net = lambda k: k["bytes"] / (k["ms"] - k["calls"] * bracket_us * 1e-3) / 1e6 / HBM_PEAK_GBS      # the event pair's own time removed
            if bracket_us is not None and 0 < bracket_us < 20:
                r["event_bracket_us"] = round(bracket_us, 2)
                r["frac_net"] = round(net(d), 4)
            i = ks_iso.get(tag)
            if i and i["ms"] > 0:       # same kernel, same shapes, nothing else on the GPU (see above)
                r["achieved_alone"] = round(i["bytes"] / i["ms"] / 1e6, 1)
                r["frac_alone"] = round(r["achieved_alone"] / HBM_PEAK_GBS, 4)
                if "frac_net" in r:
                    r["frac_alone_net"] = round(net(i), 4)
                r["valu_floor_frac_alone"] = round(i["state_steps"] / 64.0 * VALU_FLOOR_NS_PER_WAVE_STEP[tag] / N_SIMD * 1e-6 / i["ms"], 4)
            return r

        what = "fwd+bwd" if args.mode == "train" else "fwd"
        work = "training step (fwd + CE loss + bwd + AdamW)" if args.mode == "train" else \
            ("inference forward (eval, no_grad" + (", replayed from one hipGraph" if args.graph else "") + ") + CE loss")
        out = {
            "metric": f"images/sec {what} MedMamba-{args.size} {args.res}^2", "value": round(args.batch * world * args.steps / dt, 2),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"MedMamba-{args.size} {args.res}x{args.res}x3 {work}, "
                                   f"{args.batch} images per GPU resident in HBM, random-init weights",
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "parallelism": (f"dp{world} (replicas, " + (dist_info["exchange"] + " gradient all-reduce") + " over "
                                       + ("RCCL" if dist_info["backend"] == "nccl" else dist_info["backend"]) + ")") if world > 1 else "single GPU"},
            "roofline": roof("scan_fwd"), "roofline_bwd": roof("scan_bwd"),
            "final_loss": round(float(loss.detach()), 5),
        }
        if dist_info is not None:
            out["distributed"] = dist_info
        out.update(extras)
        out["miopen_find_db"] = (f"private, seeded with {MIOPEN_DB_SEEDED} recorded files of medmamba_amd/tuning/miopen_gfx950 (MIOpen searches only "
                                 "the shapes they do not hold)" if MIOPEN_DB_SEEDED else
                                 ("private, searched by this process" if _PRIVATE_MIOPEN_DB else "MIOPEN_USER_DB_PATH of the caller"))
        if out["roofline"] is not None and out["roofline"].get("traffic") is not None:
            out["roofline"]["traffic_source"] = "profiles/scan_traffic.json @ " + str(traffic.get("measured_at_commit", "unrecorded"))
        if out["roofline_bwd"] is None:
            del out["roofline_bwd"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_subprocess(args.size, args.res, mode=args.mode)
            out["cpu_baseline_ref_loop"] = cpu_baseline_subprocess(args.size, args.res, fn="cpu_baseline_ref_loop")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
