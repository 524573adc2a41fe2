#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters over EVERY kernel of a run (tools/pmc_summary.py keeps the scan kernels only):
usage: tools/pmc_by_kernel.py DIR [DIR...] [--min-us 3]   -> one line per kernel name: calls, mean duration, mean of each counter,
sorted by total time.  Counters of several passes (several DIRs) are merged by kernel name."""
import collections, csv, glob, os, re, sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:64]


def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    min_us = float(sys.argv[sys.argv.index("--min-us") + 1]) if "--min-us" in sys.argv else 3.0
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                did = r.get("Dispatch_Id")
                if (f, did) not in seen:
                    seen.add((f, did)); acc[k]["_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    names = sorted({c for v in acc.values() for c in v if c != "_us"})
    print(f"{'kernel':<66} {'calls':>6} {'us':>8} " + " ".join(f"{c[:22]:>22}" for c in names))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]["_us"])):
        us = sum(v["_us"]) / max(1, len(v["_us"]))
        if us < min_us: continue
        print(f"{k:<66} {len(v['_us']):>6} {us:>8.1f} " + " ".join(f"{(sum(v[c]) / len(v[c]) if v.get(c) else float('nan')):>22.4g}" for c in names))


if __name__ == "__main__":
    main()
