#!/usr/bin/env python3
"""Generate tests/golden/* by EXECUTING the real reference (/root/reference/MedMamba.py) on CPU.

Runs only where /root/reference exists (this build container); it is a no-op anywhere else (the
reference never travels to the GPU box — only the data files written here do).

The reference imports two packages that are absent from the image:
  * timm.layers (DropPath, trunc_normal_)  — MedMamba.py:11.  Stubbed: DropPath restated (per-sample
    stochastic depth; identity in eval / p = 0), trunc_normal_ = torch.nn.init.trunc_normal_.
  * mamba_ssm.ops.selective_scan_interface.selective_scan_fn — MedMamba.py:12 (CUDA-only, third
    party).  Stubbed with oracle.scan_ref.selective_scan_ref, the restatement of the loop quoted in
    temp.py:57-139.  So: scan-level vectors are pinned to the quoted text only ("parity unpinned" by
    any reference test), every vector ABOVE the scan is produced by the reference's own code.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def load_reference():
    from oracle.scan_ref import selective_scan_ref

    class DropPath(nn.Module):  # timm.layers.DropPath semantics
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__()
            self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            if keep > 0.0 and self.scale_by_keep:
                mask.div_(keep)
            return x * mask

    timm, layers = types.ModuleType("timm"), types.ModuleType("timm.layers")
    layers.DropPath, layers.trunc_normal_ = DropPath, torch.nn.init.trunc_normal_
    timm.layers = layers
    ms, ops = types.ModuleType("mamba_ssm"), types.ModuleType("mamba_ssm.ops")
    ssi = types.ModuleType("mamba_ssm.ops.selective_scan_interface")
    ssi.selective_scan_fn = selective_scan_ref
    ms.ops, ops.selective_scan_interface = ops, ssi
    sys.modules.update({"timm": timm, "timm.layers": layers, "mamba_ssm": ms, "mamba_ssm.ops": ops,
                        "mamba_ssm.ops.selective_scan_interface": ssi})
    sys.path.insert(0, REF)
    import MedMamba  # noqa: the reference, executed in place, never copied
    return MedMamba


def np_(t):
    return t.detach().cpu().numpy().copy()   # copy: BN running stats are later updated in place


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez(path, **arrs)
    print(f"  wrote {name}: {os.path.getsize(path) / 1e6:.2f} MB")


def sd_arrays(mod, prefix="sd/"):
    return {prefix + k: np_(v) for k, v in mod.state_dict().items()}


def grad_arrays(mod, prefix="grad/"):
    return {prefix + k: np_(p.grad) for k, p in mod.named_parameters() if p.grad is not None}


def gen_scan():
    """Scan-level vectors at the selective_scan_fn boundary (MedMamba.py:273-279): B/C are the
    NON-contiguous views of x_dbl that the reference really passes."""
    from oracle.scan_ref import selective_scan_ref
    cases = {"scan_small": (2, 4, 8, 37, 3, 0), "scan_mid": (1, 4, 24, 196, 2, 1), "scan_long": (1, 2, 4, 2049, 1, 2)}
    N = 16
    for name, (b, G, H, L, R, seed) in cases.items():
        g = torch.Generator().manual_seed(seed)
        dim = G * H
        u = torch.randn(b, dim, L, generator=g)
        delta = torch.randn(b, dim, L, generator=g)
        A = -torch.exp(torch.randn(dim, N, generator=g) * 0.5)
        x_dbl = torch.randn(b, G, R + 2 * N, L, generator=g)
        D = torch.randn(dim, generator=g)
        dt = torch.exp(torch.rand(dim, generator=g) * (np.log(0.1) - np.log(1e-3)) + np.log(1e-3))
        bias = dt + torch.log(-torch.expm1(-dt))            # softplus^-1, as SS2D.dt_init does
        dout = torch.randn(b, dim, L, generator=g)
        leaves = [t.clone().requires_grad_() for t in (u, delta, A, x_dbl, D, bias)]
        u_, d_, A_, x_, D_, b_ = leaves
        Bs, Cs = x_[:, :, R:R + N], x_[:, :, R + N:]
        out = selective_scan_ref(u_, d_, A_, Bs, Cs, D_, z=None, delta_bias=b_, delta_softplus=True)
        out.backward(dout)
        save(name + ".npz", u=np_(u), delta=np_(delta), A=np_(A), x_dbl=np_(x_dbl), D=np_(D), delta_bias=np_(bias),
             R=np.int64(R), dout=np_(dout), out=np_(out), du=np_(u_.grad), ddelta=np_(d_.grad), dA=np_(A_.grad),
             dx_dbl=np_(x_.grad), dD=np_(D_.grad), ddelta_bias=np_(b_.grad))


def gen_ss2d(M):
    for name, (dm, shape, seed) in {"ss2d_d8": (8, (2, 5, 7, 8), 10), "ss2d_d48": (48, (1, 6, 4, 48), 11)}.items():
        torch.manual_seed(seed)
        m = M.SS2D(d_model=dm)
        with torch.no_grad():  # de-trivialise the constant inits so the fixtures exercise them
            m.Ds.add_(0.1 * torch.randn_like(m.Ds))
            m.A_logs.add_(0.1 * torch.randn_like(m.A_logs))
            m.out_norm.weight.add_(0.1 * torch.randn_like(m.out_norm.weight))
            m.out_norm.bias.add_(0.1 * torch.randn_like(m.out_norm.bias))
        x = torch.randn(*shape, requires_grad=True)
        y = m(x)
        dy = torch.randn_like(y)
        y.backward(dy)
        # intermediate: the conv+SiLU output and the 4-direction core output, for kernel-level tests
        with torch.no_grad():
            xz = m.in_proj(x)
            xc = m.act(m.conv2d(xz.chunk(2, -1)[0].permute(0, 3, 1, 2).contiguous()))
            y4 = m.forward_core(xc)
            core = y4[0] + y4[1] + y4[2] + y4[3]
        save(name + ".npz", x=np_(x), y=np_(y), dy=np_(dy), dx=np_(x.grad), conv_out=np_(xc), core_out=np_(core),
             **sd_arrays(m), **grad_arrays(m))


def gen_block(M):
    torch.manual_seed(20)
    m = M.SS_Conv_SSM(hidden_dim=16, drop_path=0.0, norm_layer=nn.LayerNorm, attn_drop_rate=0.0, d_state=16)
    with torch.no_grad():
        for k, v in m.state_dict().items():
            if "running_mean" in k: v.copy_(0.1 * torch.randn_like(v))
            if "running_var" in k: v.copy_(1 + 0.2 * torch.rand_like(v))
    x = torch.randn(2, 6, 5, 16)
    sd0 = sd_arrays(m)
    m.eval()
    y_eval = m(x)
    m.train()
    xt = x.clone().requires_grad_()
    y_train = m(xt)
    dy = torch.randn_like(y_train)
    y_train.backward(dy)
    sd1 = {k.replace("sd/", "sd_after/"): v for k, v in sd_arrays(m).items() if "running" in k}
    save("block_c16.npz", x=np_(x), y_eval=np_(y_eval), y_train=np_(y_train), dy=np_(dy), dx=np_(xt.grad),
         **sd0, **sd1, **grad_arrays(m))


def gen_patchmerge(M):
    torch.manual_seed(30)
    m = M.PatchMerging2D(dim=8)
    with torch.no_grad():
        m.norm.weight.add_(0.1 * torch.randn_like(m.norm.weight)); m.norm.bias.add_(0.1 * torch.randn_like(m.norm.bias))
    xe, xo = torch.randn(2, 6, 8, 8), torch.randn(2, 7, 9, 8)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):   # the odd path prints a warning (MedMamba.py:98)
        ye, yo = m(xe), m(xo)
    save("patchmerge_c8.npz", x_even=np_(xe), y_even=np_(ye), x_odd=np_(xo), y_odd=np_(yo), **sd_arrays(m))


def gen_vssm_tiny(M):
    torch.manual_seed(40)
    depths, dims = [1, 1, 1, 1], [16, 32, 64, 128]
    net = M.VSSM(num_classes=3, depths=depths, dims=dims, drop_path_rate=0.0)
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if "running_mean" in k: v.copy_(0.05 * torch.randn_like(v))
            if "running_var" in k: v.copy_(1 + 0.1 * torch.rand_like(v))
    x = torch.randn(2, 3, 32, 32)
    labels = torch.tensor([2, 0])
    sd0 = sd_arrays(net)
    net.eval()
    logits_eval = net(x)
    net.train()
    logits_train = net(x)
    loss = nn.CrossEntropyLoss()(logits_train, labels)
    loss.backward()
    save("vssm_tiny.npz", x=np_(x), labels=np_(labels), logits_eval=np_(logits_eval), logits_train=np_(logits_train),
         loss=np_(loss), depths=np.array(depths), dims=np.array(dims), **sd0, **grad_arrays(net))


def gen_seed_kat(M, sizes=("T", "S", "Te", "B")):
    """Seed KATs (SURVEY §8c iv): same seed -> same init -> same logits, no weights shipped.
    Also per-tensor checksums of the freshly initialised state dict (pins the RNG consumption order
    of the constructors: MedMamba.py:164-184, 398-404, 470-473)."""
    cfg = {"T": ([2, 2, 4, 2], [96, 192, 384, 768], 224), "S": ([2, 2, 8, 2], [96, 192, 384, 768], 224),
           "B": ([2, 2, 12, 2], [128, 256, 512, 1024], 384), "Te": ([2, 3, 3, 2], [96, 192, 384, 768], 224)}
    path = os.path.join(OUT, "kat_seed42.json")
    kat = json.load(open(path)) if os.path.exists(path) else {}
    for s in sizes:
        depths, dims, res = cfg[s]
        torch.manual_seed(42)
        net = M.VSSM(num_classes=6, depths=depths, dims=dims).eval()
        x = torch.randn(1, 3, res, res)
        with torch.no_grad():
            logits = net(x)
        sums = {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in net.state_dict().items()
                if v.dtype.is_floating_point}
        kat[s] = dict(depths=depths, dims=dims, res=res, num_classes=6, seed=42,
                      n_params=sum(p.numel() for p in net.parameters()),
                      logits=[float(v) for v in logits[0].double()], state_checksums=sums,
                      x_checksum=[float(x.double().sum()), float(x.double().abs().sum())])
        print(f"  KAT {s}: logits {kat[s]['logits']}")
        json.dump(kat, open(path, "w"), indent=0)


def main():
    if not os.path.isdir(REF):
        print("tools/gen_golden.py: /root/reference not present — nothing to do (fixtures are committed).")
        return 0
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    M = load_reference()
    which = sys.argv[1:] or ["scan", "ss2d", "block", "patchmerge", "vssm", "kat"]
    if "scan" in which: gen_scan()
    if "ss2d" in which: gen_ss2d(M)
    if "block" in which: gen_block(M)
    if "patchmerge" in which: gen_patchmerge(M)
    if "vssm" in which: gen_vssm_tiny(M)
    if "kat" in which: gen_seed_kat(M)
    if "katB" in which: gen_seed_kat(M, ("B",))
    return 0


if __name__ == "__main__":
    sys.exit(main())
