#!/usr/bin/env python3
"""Experiment: hipGraph capture of the training step (whole step / per block) vs eager."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.modules import VSSM, MEDMAMBA_CONFIGS, SS_Conv_SSM
mode = sys.argv[1] if len(sys.argv) > 1 else "whole"
dev = torch.device("cuda:0")
torch.manual_seed(42)
net = VSSM(num_classes=6, **MEDMAMBA_CONFIGS["S"]).to(dev).train()
x = torch.randn(64, 3, 224, 224, device=dev); y = torch.randint(0, 6, (64,), device=dev)
lossf = torch.nn.CrossEntropyLoss()

def bench(step, n=10):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

if mode == "whole":
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-4, capturable=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            loss = lossf(net(x), y); loss.backward(); opt.step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        sloss = lossf(net(x), y); sloss.backward(); opt.step()
    ms = bench(g.replay)
    print(f"whole-step graph: {ms:.2f} ms/step  {64e3/ms:.1f} img/s  loss {float(sloss):.4f}")
elif mode == "blocks":
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-4)
    blocks, samples = [], []
    res, = [56]
    for li, layer in enumerate(net.layers):
        for b in layer.blocks:
            blocks.append(b)
            hw = 56 >> li
            samples.append((torch.randn(64, hw, hw, layer.dim, device=dev, requires_grad=True),))
    graphed = torch.cuda.make_graphed_callables(tuple(blocks), tuple(samples))
    k = 0
    for layer in net.layers:
        for i in range(len(layer.blocks)):
            layer.blocks[i] = graphed[k]; k += 1
    def step():
        opt.zero_grad(set_to_none=True)
        loss = lossf(net(x), y); loss.backward(); opt.step()
        return loss
    ms = bench(step)
    print(f"per-block graphs: {ms:.2f} ms/step  {64e3/ms:.1f} img/s  loss {float(step()):.4f}")
elif mode == "prio":
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-4)
    print("priority range", torch.cuda.Stream.priority_range())
    hi = torch.cuda.Stream(priority=-1)
    hi.wait_stream(torch.cuda.current_stream())
    def step():
        opt.zero_grad(set_to_none=True)
        loss = lossf(net(x), y); loss.backward(); opt.step()
        return loss
    with torch.cuda.stream(hi):
        ms = bench(step)
        print(f"eager on a high-priority main stream: {ms:.2f} ms/step  {64e3/ms:.1f} img/s  loss {float(step()):.4f}")
else:
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-4)
    def step():
        opt.zero_grad(set_to_none=True)
        loss = lossf(net(x), y); loss.backward(); opt.step()
        return loss
    ms = bench(step)
    print(f"eager: {ms:.2f} ms/step  {64e3/ms:.1f} img/s  loss {float(step()):.4f}")
