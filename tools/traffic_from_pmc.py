#!/usr/bin/env python3
"""profiles/scan_traffic.json from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py.
usage: tools/traffic_from_pmc.py DIR_FETCH DIR_WRITE OUT.json   (bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, averaged over
all launches of scan_fwd_kernel / scan_bwd_kernel in the traces; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes)"""
import collections, csv, glob, json, os, sys

def per_kernel(d, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            nm = r["Kernel_Name"]
            k = "scan_fwd" if ("scan_fwd_kernel" in nm or "scan_fwd_wg_kernel" in nm or "scan_fwd_rows_kernel" in nm) else \
                "scan_bwd" if "scan_bwd_kernel" in nm else None
            if k:
                tot[k] += float(r["Counter_Value"]); n[k] += 1
    return {k: tot[k] / n[k] for k in tot}, dict(n)

fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"note": "HBM traffic of the scan kernels inside bench.py (MedMamba-S, 64 x 224^2, training step), from two separate "
               "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE). bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE is "
               "doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests of wide coalesced streams at 64 B); WRITE_SIZE "
               "is exact for 16-B streaming stores. Mean over all launches in the trace (2 x stage1, 2 x stage2, 8 x stage3, "
               "2 x stage4 per step).", "launches_counted": {"fetch": nf, "write": nw}}
for k in ("scan_fwd", "scan_bwd"):
    out[k] = {"bytes_per_launch": round((2 * fetch[k] + write[k]) * 1024), "fetch_KB": round(fetch[k], 1), "write_KB": round(write[k], 1)}
if len(sys.argv) > 4:
    out["measured_at_commit"] = sys.argv[4]           # HEAD of the tree the passes ran on (the GPU box has no .git: passed in)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
