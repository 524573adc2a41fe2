#!/usr/bin/env python3
"""A/B on one box: the SS2D backward's parameter half on a third stream (MM_PARAM_STREAM=1).  bench.py alternating over
the variants, REPS rounds; every run under a timeout, the first failure ends the script.
usage: tools/param_stream_ab.py [REPS] [bench args...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
extra = sys.argv[2:]
VARIANTS = [("plain", {"MM_PARAM_STREAM": "0"}), ("third stream (channel-major blocks, rocBLAS kernels)", {"MM_PARAM_STREAM": "1"})]
res = {n: [] for n, _ in VARIANTS}
for rep in range(REPS):
    for name, env in VARIANTS:
        e = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "MM_PARAM_STREAM")}
        e.update(env)
        try:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-alone-pass", *extra], env=e,
                               capture_output=True, text=True, timeout=90)
            j = json.loads(p.stdout.strip().splitlines()[-1])
        except Exception as ex:  # noqa: BLE001
            print(f"rep {rep} {name}: FAILED {type(ex).__name__}", flush=True)
            sys.exit(1)
        res[name].append(j["ms_per_step"])
        print(f"rep {rep} {name[:14]:14s} {j['ms_per_step']:.3f} ms  loss {j['final_loss']}  det {j.get('deterministic', {}).get('ms_per_step')}", flush=True)
for n, x in res.items():
    print(f"{n[:14]:14s} " + " ".join(f"{t:.3f}" for t in x) + f"   median {sorted(x)[len(x) // 2]:.3f}")
