"""Build libmedmamba_hip.so for gfx950 with hipcc (no torch headers, no pybind; plain C ABI).

    python -m medmamba_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU; the .so lands in medmamba_amd/lib/ (git-ignored, travels with
gpurun snapshots).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "libmedmamba_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=fast", "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "medmamba_hip.h"))
    return hdrs


def build(force=False, verbose=False, extra=()):
    os.makedirs(LIBDIR, exist_ok=True)
    objs, dep_m = [], max(os.path.getmtime(h) for h in _deps())
    procs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), dep_m):
            cmd = [HIPCC, *FLAGS, *extra, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if procs or not os.path.exists(SO):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
