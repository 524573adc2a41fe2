"""Loader of the C++ sequencing layer (csrc_host/, built by medmamba_amd.build.build_host into lib/_mm_host.so).

Optional by design: it issues the same kernels and GEMMs as the Python route in ops.py, only without the interpreter in
between; when it has not been built (or MM_HOST_CPP=0) `module()` returns None and the Python route runs."""
import importlib.util
import os

from . import _lib

SO_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "_mm_host.so")
_ENABLED = os.environ.get("MM_HOST_CPP", "1") == "1"
_mod = None
_tried = False


def module():
    global _mod, _tried
    if not _tried:
        _tried = True
        if _ENABLED and os.path.exists(SO_PATH):
            import torch  # noqa: F401  (libtorch must be loaded before the extension)
            _lib.lib()                 # libmedmamba_hip.so first: the extension resolves the C ABI from the copy already loaded
            spec = importlib.util.spec_from_file_location("_mm_host", SO_PATH)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            if mod.abi_version() != _lib.ABI_VERSION:
                raise _lib.MedMambaHipError("_mm_host.so was built against another ABI of libmedmamba_hip.so; rebuild "
                                            "(python -m medmamba_amd.build)")
            _mod = mod
    return _mod
