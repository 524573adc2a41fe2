"""AdamW for the training step of train.py:187-201 with the per-step Python bookkeeping done once.

`torch.optim.AdamW(fused=True)` updates all parameters with one multi-tensor kernel family, but every `step()` still walks the
~365 parameters of MedMamba-S in Python to rebuild six tensor lists, groups them by device and dtype, and `zero_grad()` walks them
again: 1.9 + 0.6 ms of host time per step (`tools/host_op_profile.py`) on a step that is within 10 % of being bound by the host's
launch rate.  `FusedAdamW` is the same optimizer — same state (`step`, `exp_avg`, `exp_avg_sq` per parameter: checkpoints are
interchangeable with torch.optim.AdamW's), same order of operations — whose first step runs torch's own code (which creates the
state) and whose later steps reuse the lists.

On HIP fp32 parameters the update itself is one launch of `mm_adamw_step` (csrc/adamw.hip) instead of torch's 11 multi-tensor
launches: torch cuts the list into 65536-element chunks, 300 workgroups for MedMamba-S = a quarter of an MI355X streaming at
1.1 TB/s (0.47 ms per step, at the serial tail where nothing overlaps it); 2048-element chunks fill the chip.  `MM_HIP_ADAMW=0`
keeps torch's kernel.
"""
import ctypes
import os

import torch

_HIP_ADAMW = os.environ.get("MM_HIP_ADAMW", "1") == "1"


class FusedAdamW(torch.optim.AdamW):
    def __init__(self, params, **kw):
        kw.setdefault("fused", True)
        super().__init__(params, **kw)
        self._plans = None

    # anything that can change parameters, state tensors or their identity drops the cached lists
    def load_state_dict(self, state_dict):
        self._plans = None
        return super().load_state_dict(state_dict)

    def add_param_group(self, param_group):
        self._plans = None
        return super().add_param_group(param_group)

    def _build_plans(self):
        plans = []
        for group in self.param_groups:
            if (not group.get("fused") or group.get("amsgrad") or group.get("maximize") or group.get("differentiable")
                    or group.get("capturable") or isinstance(group["lr"], torch.Tensor)):
                return None
            params = [p for p in group["params"]]
            st = [self.state.get(p) for p in params]
            if not params or any(s is None or "exp_avg" not in s for s in st):
                return None
            if len({(p.device, p.dtype) for p in params}) != 1 or not all(s["step"].device == params[0].device for s in st):
                return None
            plans.append([group, params, [s["exp_avg"] for s in st], [s["exp_avg_sq"] for s in st], [s["step"] for s in st], None])
        if _HIP_ADAMW:
            for plan in plans:
                plan[5] = self._build_hip(plan)
        return plans

    @staticmethod
    def _build_hip(plan):
        """Device tables for mm_adamw_step, or None when the group is not dense fp32 on a HIP device with one common step count."""
        _, params, exp_avgs, exp_avg_sqs, steps, _ = plan
        ts = params + exp_avgs + exp_avg_sqs
        if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts):
            return None
        from . import _lib
        lib = _lib.lib()
        dev = params[0].device
        host_steps = torch.stack([s.detach().reshape(()) for s in steps]).cpu()
        if not bool((host_steps == host_steps[0]).all()):
            return None
        chunk, max_t = lib.mm_adamw_chunk(), lib.mm_adamw_max_tensors()
        numel = [p.numel() for p in params]
        ct, ci, groups = [], [], []
        for t0 in range(0, len(params), max_t):
            t1 = min(len(params), t0 + max_t)
            c0 = len(ct)
            for t in range(t0, t1):
                nch = (numel[t] + chunk - 1) // chunk
                ct += [t] * nch
                ci += list(range(nch))
            groups.append((t0, t1 - t0, c0, len(ct) - c0))
        i64 = lambda v: torch.tensor(v, dtype=torch.int64, device=dev)
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
        return dict(P=i64([p.data_ptr() for p in params]), M=i64([t.data_ptr() for t in exp_avgs]), V=i64([t.data_ptr() for t in exp_avg_sqs]),
                    N=i64(numel), ct=i32(ct), ci=i32(ci), groups=groups, ptrs=tuple(p.data_ptr() for p in params), numel=numel,
                    step=float(host_steps[0]), dev=dev, lib=lib, _lib=_lib, written=tuple(ts))

    def _hip_step(self, plan, grads):
        """One mm_adamw_step launch per <= 448 tensors.  False (nothing done) if a tensor moved or a gradient is not dense fp32."""
        group, params, _, _, steps, h = plan
        if tuple(p.data_ptr() for p in params) != h["ptrs"]:
            return False
        gp = []
        for g, n in zip(grads, h["numel"]):
            if g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != n or g.device != h["dev"]:
                return False
            gp.append(g.data_ptr())
        beta1, beta2 = group["betas"]
        torch._foreach_add_(steps, 1)
        h["step"] += 1.0
        _lib, lib = h["_lib"], h["lib"]
        with _lib.device_guard(h["dev"]):
            stream = _lib.raw_stream()
            for t0, nt, c0, nc in h["groups"]:
                arr = (ctypes.c_void_p * nt)(*gp[t0:t0 + nt])
                rc = lib.mm_adamw_step(h["P"].data_ptr(), arr, t0, nt, h["M"].data_ptr(), h["V"].data_ptr(), h["N"].data_ptr(),
                                       h["ct"].data_ptr() + 4 * c0, h["ci"].data_ptr() + 4 * c0, nc, group["lr"], beta1, beta2,
                                       group["eps"], group["weight_decay"], h["step"], stream)
                _lib.check(rc, "mm_adamw_step")
        # the kernel wrote parameters and moments through raw pointers: bump their version counters as torch's _fused_adamw_ does —
        # SS_Conv_SSM._eval_fold and GraphedInference key their caches on (data_ptr, _version) and must see an optimizer step
        torch.autograd.graph.increment_version(h["written"])
        return True

    def _torch_step(self, closure=None):
        """torch's own step, then the version bump the inference caches rely on (SS_Conv_SSM._eval_fold, GraphedInference: keyed on
        (data_ptr, _version)).  Measured on this build (tests/test_optim_gpu.py): after torch's fused multi-tensor update the
        folded constants were NOT rebuilt, i.e. `_fused_adamw_` does not advance the parameters' version counters here."""
        out = super().step(closure)
        ps = [p for group in self.param_groups for p in group["params"]]
        if ps:
            torch.autograd.graph.increment_version(ps)
        return out

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or getattr(self, "grad_scale", None) is not None or getattr(self, "found_inf", None) is not None:
            self._plans = None
            return self._torch_step(closure)
        if self._plans is None:
            self._plans = self._build_plans()          # state that load_state_dict brought along: the cached path from this step on
            if self._plans is None:                    # (a resumed run then takes the same kernels as the run it continues)
                out = self._torch_step()               # torch's own step: creates / validates the state
                self._plans = self._build_plans() or False
                return out
        if self._plans is False:
            return self._torch_step()
        all_grads = []
        for _, params, *_ in self._plans:
            grads = [p.grad for p in params]
            for g in grads:
                if g is None:                          # a parameter without a gradient this step: torch's general path
                    self._plans = None                 # (its step counters then run apart: rebuild the lists afterwards)
                    return self._torch_step()
            all_grads.append(grads)
        for plan, grads in zip(self._plans, all_grads):
            group, params, exp_avgs, exp_avg_sqs, steps, hip = plan
            if hip is not None:
                if self._hip_step(plan, grads):
                    continue
                plan[5] = None                         # a tensor moved / odd gradient: torch's kernel from here on for this group
            beta1, beta2 = group["betas"]
            torch._foreach_add_(steps, 1)
            torch._fused_adamw_(params, grads, exp_avgs, exp_avg_sqs, [], steps, amsgrad=False, lr=group["lr"], beta1=beta1,
                                beta2=beta2, weight_decay=group["weight_decay"], eps=group["eps"], maximize=False,
                                grad_scale=None, found_inf=None)
            torch.autograd.graph.increment_version(params)
        return None

    def zero_grad(self, set_to_none=True):
        if not set_to_none or not self._plans:
            return super().zero_grad(set_to_none)
        for _, params, *_ in self._plans:
            for p in params:
                p.grad = None
