#!/usr/bin/env python3
"""Per-(kernel, grid size) launch statistics from a rocprofv3 --kernel-trace CSV: which multigrid LEVEL a stencil kernel's time
goes to (the level fixes the grid size).  python tools/trace_by_grid.py <dir or kernel_trace.csv> [substring ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

src = sys.argv[1]
pats = sys.argv[2:] or [""]
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
acc = defaultdict(list)
for f in files:
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"]
            if not any(p in name for p in pats):
                continue
            g = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
            acc[(name.split("(")[0][-48:], g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values())
print(f"{'kernel':50s} {'grid':>10s} {'calls':>7s} {'avg us':>9s} {'min':>8s} {'max':>8s} {'total ms':>9s} {'share':>6s}")
for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:50s} {g:10d} {len(v):7d} {sum(v) / len(v):9.1f} {min(v):8.1f} {max(v):8.1f} {sum(v) / 1e3:9.2f} {100 * sum(v) / tot:5.1f}%")
