#!/usr/bin/env python3
"""Which solver should the recorded MIOpen database name where its search found a near tie?  For the 14x14-stage forward / data-gradient
convolutions of S / 64 (Winograd f2x3 107.6 / 105.1 us vs the NHWC implicit-GEMM kernel 114.9 / 111.7 us alone) this writes copies of
medmamba_amd/tuning/miopen_gfx950/ whose rankings are edited (the chosen solver's time set below the winner's) and runs bench.py on each,
alternating, REPS times: the step time decides, not the kernel alone.  usage: tools/miopen_pick_ab.py [REPS]"""
import json, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "medmamba_amd", "tuning", "miopen_gfx950")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
F, B = "192-14-14-3x3-192-14-14-64-1x1-1x1-1x1-0-NCHW-FP32-F", "192-14-14-3x3-192-14-14-64-1x1-1x1-1x1-0-NCHW-FP32-B"
VARIANTS = {"recorded (Winograd F, Winograd B)": {}, "igemm F": {F: "ConvAsmImplicitGemmGTCDynamicFwdXdlopsNHWC"},
            "igemm B": {B: "ConvAsmImplicitGemmGTCDynamicBwdXdlopsNHWC"},
            "igemm F + B": {F: "ConvAsmImplicitGemmGTCDynamicFwdXdlopsNHWC", B: "ConvAsmImplicitGemmGTCDynamicBwdXdlopsNHWC"}}


def make_db(edits):
    d = tempfile.mkdtemp(prefix="mm_pick_")
    for f in os.listdir(SRC):
        if not f.endswith(".ufdb.txt"):
            shutil.copy(os.path.join(SRC, f), d)
            continue
        out = []
        for line in open(os.path.join(SRC, f)):
            key, val = line.rstrip("\n").split("=", 1)
            if key in edits:
                ents = [e.split(":", 1) for e in val.split(";")]
                best = min(float(r.split(",")[0]) for _, r in ents)
                ents = [(n, (f"{best * 0.5:g}," + r.split(",", 1)[1]) if n == edits[key] else r) for n, r in ents]
                val = ";".join(f"{n}:{r}" for n, r in ents)
            out.append(f"{key}={val}\n")
        open(os.path.join(d, f), "w").writelines(out)
    return d


dbs = {name: make_db(ed) for name, ed in VARIANTS.items()}
res = {name: [] for name in VARIANTS}
for rep in range(REPS):
    for name, d in dbs.items():
        env = dict(os.environ, MIOPEN_USER_DB_PATH=d)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-alone-pass"], env=env, capture_output=True, text=True)
        j = json.loads(p.stdout.strip().splitlines()[-1])
        res[name].append(j["ms_per_step"])
        print(f"rep {rep} {name:36s} {j['ms_per_step']:.3f} ms  {j.get('conv_kernel_families')}", flush=True)
for name, v in res.items():
    print(f"{name:36s} " + " ".join(f"{x:.3f}" for x in v) + f"   min {min(v):.3f}")
