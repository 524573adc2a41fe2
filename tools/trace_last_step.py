#!/usr/bin/env python3
"""Per-kernel time of the LAST training step in a rocprofv3 --kernel-trace csv (warm-up steps contain
MIOpen's solver search and must not be counted).  A step starts at the first scan_fwd launch of a group of
`nscan` (= number of blocks: 14 for MedMamba-S).  usage: tools/trace_last_step.py DIR [nscan] [top]"""
import collections, csv, glob, os, re, sys

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n[:100]

def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    d = args[0]; nscan = int(args[1]) if len(args) > 1 else 14; top = int(args[2]) if len(args) > 2 else 45
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if re.search(r"scan_fwd(_wg|_rows)?_kernel", r["Kernel_Name"])]      # all forward kernels
    start = idx[-nscan]
    last = rows[start:]
    t0, t1 = int(last[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in last)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in last:
        a = agg[short(r["Kernel_Name"])]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy = sum(v[1] for v in agg.values())
    print(f"# source: {f}\n# last step: {len(last)} launches, wall {(t1-t0)/1e6:.3f} ms, sum of kernel time {busy/1e3:.3f} ms")
    qs = collections.defaultdict(lambda: [0, 0.0])
    for r in last:
        q = qs[r.get("Queue_Id", "?")]; q[0] += 1; q[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ev = sorted([(int(r["Start_Timestamp"]), 1) for r in last] + [(int(r["End_Timestamp"]), -1) for r in last])
    depth, prev, union, both = 0, ev[0][0], 0, 0
    for t, d in ev:
        if depth > 0: union += t - prev
        if depth > 1: both += t - prev
        depth += d; prev = t
    print("# per HSA queue: " + ", ".join(f"q{k}: {n} launches {us/1e3:.2f} ms" for k, (n, us) in sorted(qs.items())))
    print(f"# GPU busy (union of kernel intervals) {union/1e6:.3f} ms, >=2 kernels in flight {both/1e6:.3f} ms")
    print(f"{'kernel':<102} {'calls':>6} {'total_us':>10} {'avg_us':>9} {'pct':>6}")
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k:<102} {n:>6} {us:>10.1f} {us/n:>9.2f} {100*us/busy:>6.2f}")
    if "--per-queue" in sys.argv:
        for qid in sorted(qs):
            sub = collections.defaultdict(lambda: [0, 0.0])
            for r in last:
                if r.get("Queue_Id", "?") == qid:
                    a = sub[short(r["Kernel_Name"])]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            tot = sum(v[1] for v in sub.values())
            print(f"\n# queue {qid}: {sum(v[0] for v in sub.values())} launches, {tot/1e3:.3f} ms of kernel time")
            for k, (n, us) in sorted(sub.items(), key=lambda kv: -kv[1][1])[:30]:
                print(f"{k:<102} {n:>6} {us:>10.1f} {us/n:>9.2f} {100*us/tot:>6.2f}")

if __name__ == "__main__":
    main()
