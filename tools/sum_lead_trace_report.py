#!/usr/bin/env python3
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
shapes = [(64, 2, 70, 96), (64, 768, 384), (1024, 96), (1024, 384), (1024, 768), (4096, 768), (3136, 384), (448, 768)]
ours = [r for r in rows if "sum_lead_kernel" in r["Kernel_Name"]]; aten = [r for r in rows if "reduce_kernel" in r["Kernel_Name"]]
d = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("sum_lead launches", len(ours), "ATen reduce launches", len(aten))
per = len(aten) // len(shapes)
for i, s in enumerate(shapes):
    a = sorted(d(r) for r in ours[i * 20:(i + 1) * 20]); b = sorted(d(r) for r in aten[i * per:(i + 1) * per])
    print(f"{str(s):<20} mm_sum_lead {a[len(a) // 2]:6.1f} us   torch.sum {sum(b[len(b)//4:]) / max(1, len(b) - len(b)//4) * (per / 20):6.1f} us ({per // 20} kernel(s) per call)")
