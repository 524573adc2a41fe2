"""Build libmedmamba_hip.so for gfx950 with hipcc (no torch headers, no pybind; plain C ABI).

    python -m medmamba_amd.build [--force] [--verbose] [--experiments]

--experiments additionally builds lib/libmedmamba_hip_exp.so from the same sources with -DMM_EXPERIMENTS: the product library plus
the measured-and-rejected experiments (own dense 3x3 convolutions, csrc/conv.hip; the forward scan's timing-ablation bits).  Only
tests / tools that exercise those load it (_lib.exp_lib()); the product path never does.

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


EXP_SO = os.path.join(LIBDIR, "libmedmamba_hip_exp.so")


def build(force=False, verbose=False, extra=(), experiments=False):
    """The product library; experiments=True: the experiments variant (own object directory, own .so) instead."""
    os.makedirs(LIBDIR, exist_ok=True)
    objdir, so = (os.path.join(LIBDIR, "exp"), EXP_SO) if experiments else (LIBDIR, SO)
    os.makedirs(objdir, exist_ok=True)
    if experiments:
        extra = tuple(extra) + ("-DMM_EXPERIMENTS",)
    objs, dep_m = [], max(os.path.getmtime(h) for h in _deps())
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), dep_m):
            cmd = [HIPCC, *FLAGS, *extra, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if procs or not os.path.exists(so):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return so


# ---- the C++ sequencing layer (csrc_host/*.cpp): a torch extension without device code, compiled with g++ against torch's
# headers and linked against the library above (ops.SS2DBranchFn loads it; everything still works without it, from Python)
HOST_SRC = os.path.join(HERE, "csrc_host")
HOST_SO = os.path.join(LIBDIR, "_mm_host.so")
CXX = os.environ.get("CXX", "g++")


def build_host(force=False, verbose=False):
    import sysconfig

    import torch
    from torch.utils import cpp_extension as ce
    build(force=False, verbose=verbose)
    srcs = sorted(os.path.join(HOST_SRC, f) for f in os.listdir(HOST_SRC) if f.endswith(".cpp"))
    newest = max([os.path.getmtime(f) for f in srcs] + [os.path.getmtime(os.path.join(ROOT, "include", "medmamba_hip.h"))])
    if not force and os.path.exists(HOST_SO) and os.path.getmtime(HOST_SO) >= newest:
        return HOST_SO
    inc = [p for p in ce.include_paths() if os.path.isdir(p)] + [sysconfig.get_paths()["include"], os.path.join(ROOT, "include")]
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [CXX, "-O2", "-std=c++17", "-fPIC", "-shared", "-DTORCH_EXTENSION_NAME=_mm_host", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wall", "-Wno-unused-function",
           *["-I" + p for p in inc], *srcs, "-o", HOST_SO, "-L" + LIBDIR, "-lmedmamba_hip", "-Wl,-rpath,$ORIGIN",
           "-L" + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python", "-Wl,-rpath," + tlib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return HOST_SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
    if "--experiments" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, experiments=True))
