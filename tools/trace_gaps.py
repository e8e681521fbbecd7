#!/usr/bin/env python3
"""Where the GPU idles inside a solve: gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (one stream), grouped by
the kernel that FOLLOWS the gap (the launch that was late), plus wall span / summed kernel time of the region between the first and
the last multigrid launch.   python tools/trace_gaps.py <dir or kernel_trace.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

src = sys.argv[1]
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]))
rows.sort()
mg = [i for i, r in enumerate(rows) if "k_f_smooth" in r[2] or "k_st_smooth" in r[2]]
rows = rows[mg[0]:mg[-1] + 1]
span = (rows[-1][1] - rows[0][0]) / 1e3
busy = sum(e - s for s, e, _ in rows) / 1e3
gaps = defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gaps[n1].append(max(0, s1 - e0) / 1e3)
tot_gap = sum(sum(v) for v in gaps.values())
print(f"region: {len(rows)} launches, span {span / 1e3:.2f} ms, summed kernel time {busy / 1e3:.2f} ms, gaps {tot_gap / 1e3:.2f} ms "
      f"({100 * tot_gap / span:.1f} % of the span)")
print(f"{'gap BEFORE kernel':46s} {'count':>7s} {'avg us':>8s} {'max us':>8s} {'total ms':>9s}")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"{k:46s} {len(v):7d} {sum(v) / len(v):8.2f} {max(v):8.1f} {sum(v) / 1e3:9.3f}")
