#!/usr/bin/env python3
"""Print a VGPR/SGPR/spill/occupancy table for every kernel in medmamba_amd/csrc/*.hip
(hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles for gfx950, no GPU needed)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medmamba_amd.build import FLAGS, HIPCC, sources

def main():
    only = sys.argv[1:]
    for src in sources():
        if only and not any(o in src for o in only):
            continue
        out = subprocess.run([HIPCC, *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                             capture_output=True, text=True).stderr
        cur = None
        for line in out.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                cur = dict(name=re.sub(r"\(anonymous namespace\)::|\(.*", "", name))
                continue
            if cur is None:
                continue
            for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"),
                             ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                             ("vspill", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
                m = re.search(pat, line)
                if m:
                    cur[key] = int(m.group(1))
            if "lds" in cur:
                print(f"{cur['name']:<58} vgpr {cur.get('vgpr'):>3} sgpr {cur.get('sgpr'):>3} spill {cur.get('vspill'):>3} "
                      f"scratch {cur.get('scratch'):>4} occ {cur.get('occ')}")
                cur = None

if __name__ == "__main__":
    main()
