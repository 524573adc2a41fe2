"""AdamW for the training step of train.py:187-201 with the per-step Python bookkeeping done once.

`torch.optim.AdamW(fused=True)` updates all parameters with one multi-tensor kernel family, but every `step()` still walks the
~365 parameters of MedMamba-S in Python to rebuild six tensor lists, groups them by device and dtype, and `zero_grad()` walks them
again: 1.9 + 0.6 ms of host time per step (`tools/host_op_profile.py`) on a step that is within 10 % of being bound by the host's
launch rate.  `FusedAdamW` is the same optimizer — same state (`step`, `exp_avg`, `exp_avg_sq` per parameter: checkpoints are
interchangeable with torch.optim.AdamW's), same `torch._fused_adamw_` kernel, same order of operations — whose first step runs
torch's own code (which creates the state) and whose later steps reuse the lists.
"""
import torch


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
            plans.append((group, params, [s["exp_avg"] for s in st], [s["exp_avg_sq"] for s in st], [s["step"] for s in st]))
        return plans

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or getattr(self, "grad_scale", None) is not None or getattr(self, "found_inf", None) is not None:
            self._plans = None
            return super().step(closure)
        if self._plans is None:
            out = super().step()                      # torch's own step: creates / validates the state
            self._plans = self._build_plans() or False
            return out
        if self._plans is False:
            return super().step()
        all_grads = []
        for _, params, *_ in self._plans:
            grads = [p.grad for p in params]
            for g in grads:
                if g is None:                          # a parameter without a gradient this step: torch's general path
                    return super().step()
            all_grads.append(grads)
        for (group, params, exp_avgs, exp_avg_sqs, steps), grads in zip(self._plans, all_grads):
            beta1, beta2 = group["betas"]
            torch._foreach_add_(steps, 1)
            torch._fused_adamw_(params, grads, exp_avgs, exp_avg_sqs, [], steps, amsgrad=False, lr=group["lr"], beta1=beta1,
                                beta2=beta2, weight_decay=group["weight_decay"], eps=group["eps"], maximize=False,
                                grad_scale=None, found_inf=None)
        return None

    def zero_grad(self, set_to_none=True):
        if not set_to_none or not self._plans:
            return super().zero_grad(set_to_none)
        for _, params, *_ in self._plans:
            for p in params:
                p.grad = None
