#!/usr/bin/env python3
"""Every dispatch of the kernels whose name contains PATTERN in the LAST training step of a rocprofv3 --kernel-trace csv: start
offset inside the step, duration, grid / workgroup size, HSA queue, and the kernels launched around it on the same queue — enough
to tell which call site issued it (a GEMM kernel that is 10x slower in two of its four calls).
usage: tools/trace_kernel_instances.py DIR PATTERN [nscan] [context]"""
import csv, glob, os, re, sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n[:70]


def main():
    d, pat = sys.argv[1], sys.argv[2]
    nscan = int(sys.argv[3]) if len(sys.argv) > 3 else 14
    ctx = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if re.search(r"scan_fwd(_wg|_rows)?_kernel", r["Kernel_Name"])]      # all forward kernels
    last = rows[idx[-nscan]:]
    t0 = int(last[0]["Start_Timestamp"])
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    byq = {}
    for r in last:
        byq.setdefault(r.get("Queue_Id", "?"), []).append(r)
    print(f"# source {f}; last step: {len(last)} launches; pattern {pat!r}")
    for q, lst in sorted(byq.items()):
        for i, r in enumerate(lst):
            if pat not in r["Kernel_Name"]:
                continue
            grid = "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
            wg = "x".join(r.get(k, "?") for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
            print(f"\nq{q} +{(int(r['Start_Timestamp']) - t0) / 1e6:8.3f} ms  {dur(r):9.1f} us  grid {grid} wg {wg} lds {r.get('LDS_Block_Size', '?')} "
                  f"vgpr {r.get('VGPR_Count', '?')}  {short(r['Kernel_Name'])}")
            # kernels of the OTHER queues that were running during this dispatch (overlap in us): a small kernel that shares the chip
            # with a long one waits for wave slots, and its "duration" is mostly the other kernel's
            s0, e0 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            for q2, lst2 in sorted(byq.items()):
                if q2 == q:
                    continue
                for o in lst2:
                    ov = min(e0, int(o["End_Timestamp"])) - max(s0, int(o["Start_Timestamp"]))
                    if ov > 0.2 * (e0 - s0):
                        print(f"      beside q{q2} {ov / 1e3:9.1f} us of {dur(o):9.1f} us  grid {o.get('Grid_Size_X', '?')} wg {o.get('Workgroup_Size_X', '?')} "
                              f"{short(o['Kernel_Name'])}")
            for j in range(max(0, i - ctx), min(len(lst), i + ctx + 1)):
                if j != i:
                    print(f"      {'before' if j < i else 'after '} {dur(lst[j]):9.1f} us  {short(lst[j]['Kernel_Name'])}")


if __name__ == "__main__":
    main()
