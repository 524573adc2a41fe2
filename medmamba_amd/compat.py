"""Import-time shim: lets an UNCHANGED reference model file run on the HIP path.

The reference does `from timm.layers import DropPath, trunc_normal_` (MedMamba.py:11) and
`from mamba_ssm.ops.selective_scan_interface import selective_scan_fn` (MedMamba.py:12).  Importing this
module registers module objects under those names in sys.modules — only when the real packages are absent —
whose attributes are the implementations in this package.  Nothing here computes anything.
"""
import importlib.util
import sys
import types

from .modules import DropPath, trunc_normal_
from .selective_scan_interface import selective_scan_fn


def _absent(name):
    if name in sys.modules:
        return False
    try:
        return importlib.util.find_spec(name) is None
    except (ImportError, ValueError):
        return True


def install(force=False):
    done = []
    if force or _absent("mamba_ssm"):
        pkg, ops = types.ModuleType("mamba_ssm"), types.ModuleType("mamba_ssm.ops")
        ssi = types.ModuleType("mamba_ssm.ops.selective_scan_interface")
        ssi.selective_scan_fn = selective_scan_fn
        pkg.ops, ops.selective_scan_interface = ops, ssi
        pkg.__path__, ops.__path__ = [], []
        sys.modules.update({"mamba_ssm": pkg, "mamba_ssm.ops": ops, "mamba_ssm.ops.selective_scan_interface": ssi})
        done.append("mamba_ssm")
    if force or _absent("timm"):
        timm, layers = types.ModuleType("timm"), types.ModuleType("timm.layers")
        layers.DropPath, layers.trunc_normal_ = DropPath, trunc_normal_
        timm.layers = layers
        timm.__path__ = []
        sys.modules.update({"timm": timm, "timm.layers": layers})
        done.append("timm")
    return done


INSTALLED = install()
