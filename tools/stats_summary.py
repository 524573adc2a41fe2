#!/usr/bin/env python3
"""Condense rocprofv3 --kernel-trace --stats output (kernel_stats.csv) into a short, committable table.
usage: tools/stats_summary.py DIR [top_n] > profiles/NAME.txt"""
import csv, glob, os, re, sys

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([\w:]+(?:<[^(]{0,60})?)", n)
    s = m.group(1) if m else n
    return s[:110]

def main():
    d = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        print("no kernel_stats.csv under", d); return
    rows = list(csv.DictReader(open(files[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# source: {files[0]}\n# total kernel time {tot/1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} launches")
    print(f"{'kernel':<112} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'pct':>6}")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
        print(f"{short(r['Name']):<112} {r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:>10.3f} {float(r['AverageNs'])/1e3:>10.2f} {float(r['Percentage']):>6.2f}")

if __name__ == "__main__":
    main()
