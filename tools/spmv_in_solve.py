#!/usr/bin/env python3
"""Durations of the operator-apply kernel INSIDE the solves vs in pgx_spmv_bench's back-to-back loop, from a rocprofv3 kernel
trace of `bench.py` (the loop's launches are the trailing run of consecutive launches of the kernel):
    python tools/spmv_in_solve.py <dir or kernel_trace.csv> [kernel substring, default k_st_spmv_r]"""
import csv
import glob
import os
import sys

src = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "k_st_spmv_r"
f = src if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f, newline="")))
runs, cur = [], []
for s, e, n in rows:
    if name in n:
        cur.append(e - s)
    elif cur:
        runs.append(cur)
        cur = []
if cur:
    runs.append(cur)
loop = [r for r in runs if len(r) >= 20]
solve = [d for r in runs if len(r) < 20 for d in r]
avg = lambda v: sum(v) / max(len(v), 1) / 1e3  # noqa: E731
print(f"{name}: in-solve launches {len(solve)}  avg {avg(solve):.1f} us  min {min(solve) / 1e3:.1f}  max {max(solve) / 1e3:.1f}")
for r in loop:
    print(f"  back-to-back loop of {len(r)} launches: avg {avg(r):.1f} us")
