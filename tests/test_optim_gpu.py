"""GPU: optim.FusedAdamW's one-launch update (mm_adamw_step, csrc/adamw.hip) against torch.optim.AdamW(fused=True)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model():
    torch.manual_seed(0)
    # tensors of many sizes: below / above one 2048-element chunk, odd lengths (scalar tail), a 0-d-like single element
    return torch.nn.Sequential(torch.nn.Linear(37, 129), torch.nn.LayerNorm(129), torch.nn.Linear(129, 4100), torch.nn.Linear(4100, 3),
                               torch.nn.Conv2d(3, 5, 3)).to(DEV)


def _grads(net, g):
    for p in net.parameters():
        p.grad = torch.randn(p.shape, device=DEV, generator=g)


def test_one_launch_adamw_matches_torch_fused_adamw(monkeypatch):
    from medmamba_amd import _lib, optim
    calls = []
    real = _lib.lib().mm_adamw_step
    net, ref = _model(), _model()
    opt = optim.FusedAdamW(net.parameters(), lr=3e-3, weight_decay=1e-2, betas=(0.9, 0.99))
    ropt = torch.optim.AdamW(ref.parameters(), lr=3e-3, weight_decay=1e-2, betas=(0.9, 0.99), fused=True)
    g1, g2 = torch.Generator(device=DEV).manual_seed(1), torch.Generator(device=DEV).manual_seed(1)
    for step in range(6):
        _grads(net, g1)
        _grads(ref, g2)
        if step == 3:
            opt.param_groups[0]["lr"] = ropt.param_groups[0]["lr"] = 1e-3       # a scheduler step in between
        opt.step()
        ropt.step()
    assert opt._plans and opt._plans[0][5] is not None, "the HIP update did not engage"
    assert opt._plans[0][5]["step"] == 6.0
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), float((a - b).abs().max())
    sa, sb = opt.state_dict()["state"], ropt.state_dict()["state"]
    for k in sb:
        assert float(sa[k]["step"]) == float(sb[k]["step"]) == 6.0
        assert torch.allclose(sa[k]["exp_avg"], sb[k]["exp_avg"], rtol=2e-6, atol=1e-7)
        assert torch.allclose(sa[k]["exp_avg_sq"], sb[k]["exp_avg_sq"], rtol=2e-6, atol=1e-7)
    # checkpoint written by the HIP path continues identically in torch's optimizer and vice versa (within the same tolerance)
    net2 = _model()
    net2.load_state_dict(net.state_dict())
    o2 = torch.optim.AdamW(net2.parameters(), lr=1e-3, weight_decay=1e-2, betas=(0.9, 0.99), fused=True)
    o2.load_state_dict(copy.deepcopy(opt.state_dict()))
    _grads(net, g1)
    for p, q in zip(net.parameters(), net2.parameters()):
        q.grad = p.grad.clone()
    opt.step()
    o2.step()
    for a, b in zip(net.parameters(), net2.parameters()):
        assert torch.allclose(a, b, rtol=2e-6, atol=2e-7)
    # a parameter whose storage moved falls back to torch's kernel without losing the step
    w = next(net.parameters())
    w.data = w.data.clone()
    _grads(net, g1)
    before = float(opt.state[w]["step"])
    opt.step()
    assert float(opt.state[w]["step"]) == before + 1 and opt._plans[0][5] is None


def test_full_model_update_in_one_launch():
    """MedMamba-S has 365 parameter tensors: one launch (<= 448 tensors per launch), every chunk of every tensor updated once."""
    from medmamba_amd import optim
    from medmamba_amd.modules import MEDMAMBA_CONFIGS, VSSM
    torch.manual_seed(1)
    net = VSSM(num_classes=6, **MEDMAMBA_CONFIGS["S"]).to(DEV)
    ref = copy.deepcopy(net)
    opt = optim.FusedAdamW(net.parameters(), lr=1e-4, weight_decay=1e-4)
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
    g = torch.Generator(device=DEV).manual_seed(5)
    for _ in range(3):
        for p, q in zip(net.parameters(), ref.parameters()):
            p.grad = torch.randn(p.shape, device=DEV, generator=g) * 1e-2
            q.grad = p.grad.clone()
        opt.step()
        ropt.step()
    h = opt._plans[0][5]
    assert h is not None and len(h["groups"]) == 1 and h["groups"][0][1] == len(list(net.parameters()))
    worst = max(float((a - b).abs().max()) for a, b in zip(net.parameters(), ref.parameters()))
    assert worst <= 1e-6, worst


def test_hip_adamw_step_is_seen_by_the_version_keyed_inference_caches():
    """mm_adamw_step writes parameters through raw pointers; the folded BatchNorm constants of
    SS_Conv_SSM._eval_fold and GraphedInference's capture are keyed on (data_ptr, _version) and must notice the step while the
    net stays in eval mode (frozen-BN fine-tuning; a graph kept across steps)."""
    from medmamba_amd import optim
    from medmamba_amd.graphs import GraphedInference
    from medmamba_amd.modules import SS_Conv_SSM, VSSM
    torch.manual_seed(3)
    net = VSSM(num_classes=3, depths=[1, 1, 1, 1], dims=[16, 32, 64, 128], drop_path_rate=0.0).to(DEV).eval()
    x = torch.randn(2, 3, 64, 64, device=DEV)
    params = list(net.parameters())
    v0 = [p._version for p in params]
    opt = optim.FusedAdamW(params, lr=5e-2, weight_decay=0.0)
    g = torch.Generator(device=DEV).manual_seed(9)
    graphed = GraphedInference(net, x)
    for step in range(3):                    # step 0 is torch's own (creates the state); from step 1 on the HIP launch
        with torch.no_grad():
            before = net(x).clone()          # builds / reuses the fold
        for p in params:
            p.grad = torch.randn(p.shape, device=DEV, generator=g)
        opt.step()
        with torch.no_grad():
            after = net(x).clone()
            for m in net.modules():
                if isinstance(m, SS_Conv_SSM):
                    m._fold_cache = None
            fresh = net(x)
        # a step with lr 5e-2 moves the logits by O(1); a forward on stale folded constants would sit between the two.  (Not
        # bitwise: the eval forward's library GEMMs / convolutions are not guaranteed run-to-run identical.)
        moved = float((before - after).abs().max())
        stale = float((after - fresh).abs().max())
        assert moved > 1e-2 and stale <= 1e-5 * max(1.0, float(fresh.abs().max())), (step, moved, stale)
    assert opt._plans and opt._plans[0][5] is not None, "the HIP update did not engage"
    assert all(p._version >= v + 3 for p, v in zip(params, v0))
    n = graphed.captures
    out = graphed(x)
    assert graphed.captures == n + 1, "GraphedInference replayed a graph captured before the optimizer steps"
    with torch.no_grad():
        assert torch.allclose(out, net(x), rtol=1e-5, atol=1e-6)
