#!/usr/bin/env python3
"""How much does a hipEvent bracket add to a kernel's duration?  Run under `rocprofv3 --kernel-trace`: the forward scan of the S
stages is launched back to back, each launch inside an event pair (as bench.py's KERNEL_TIMER does); prints the event medians, and
tools/event_vs_trace.py --report DIR compares them with the trace's durations of the same launches."""
import csv, glob, json, os, statistics, sys
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    ev = json.load(open(os.path.join(sys.argv[2], "events.json")))
    rows = [r for r in csv.DictReader(open(f)) if "scan_fwd" in r["Kernel_Name"]]
    by = {}
    for r in rows:
        by.setdefault((r["Kernel_Name"][:60], r["Grid_Size_X"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("# events (median us per launch, bracket of two events on the launch stream):")
    for k, v in ev.items():
        print(f"  {k}: {v:.1f}")
    print("# kernel trace (median us of the same launches):")
    for (k, g), v in sorted(by.items(), key=lambda kv: -statistics.median(kv[1])):
        print(f"  {k} grid {g}: n={len(v)} median {statistics.median(v):.1f}")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmamba_amd.selective_scan_interface import SelectiveScanFn
from tools.bench_scan import STAGES
dev = torch.device("cuda:0"); K, N, Bz = 4, 16, 64
out = {}
for D, L, _ in STAGES["S"]:
    g = torch.Generator(device=dev).manual_seed(0)
    R = max(1, (D // 2 + 15) // 16)
    u = torch.randn(Bz, K * D, L, device=dev, generator=g); delta = torch.randn(Bz, K * D, L, device=dev, generator=g)
    A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(K * D, 1)
    x_dbl = torch.randn(Bz, K, R + 2 * N, L, device=dev, generator=g)
    Bs, Cs = x_dbl[:, :, R:R + N], x_dbl[:, :, R + N:]
    Dp = torch.ones(K * D, device=dev); bias = torch.randn(K * D, device=dev, generator=g) - 4.0
    fn = lambda: SelectiveScanFn.apply(u, delta, A, Bs, Cs, Dp, bias, True, 0)
    for _ in range(10): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    out[f"L={L}"] = statistics.median(s.elapsed_time(e) for s, e in evs) * 1e3
    # the empty bracket in the same queue state
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for s, e in evs:
        fn(); s.record(); e.record()
    torch.cuda.synchronize()
    out[f"L={L} empty bracket"] = statistics.median(s.elapsed_time(e) for s, e in evs) * 1e3
print(json.dumps(out))
os.makedirs(sys.argv[1], exist_ok=True)
json.dump(out, open(os.path.join(sys.argv[1], "events.json"), "w"))
