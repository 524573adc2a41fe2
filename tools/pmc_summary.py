#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv / kernel_trace.csv files for OUR kernels only.
usage: tools/pmc_summary.py DIR [DIR...] [--match scan_]   -> per kernel: mean duration, mean counters."""
import collections, csv, glob, os, re, sys

def short(name):
    m = re.search(r"((?:scan|ss2d|mm)_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None

def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not k: continue
                key = (k, r["Grid_Size"], r["Workgroup_Size"])
                out[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                out[key]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                out[key]["_vgpr"] = [float(r["VGPR_Count"])]; out[key]["_lds"] = [float(r["LDS_Block_Size"])]
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not k: continue
                key = (k, r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")))
                out[key]["_trace_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for key in sorted(out):
        v = out[key]
        print(f"== {key[0]} grid={key[1]} wg={key[2]}")
        for c in sorted(v):
            xs = v[c]
            print(f"   {c:<28} n={len(xs):<4} mean={sum(xs)/len(xs):.6g}")

if __name__ == "__main__":
    main()
