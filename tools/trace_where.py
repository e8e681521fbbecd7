"""When do the launches of one kernel happen relative to the Newton solves?  For a rocprofv3 --kernel-trace directory: prints every
launch of kernels matching PATTERN with its start relative to the first k_st_spmv_r launch (the first Krylov iteration) and the
duration - tells setup-time launches (negative offsets) from launches inside the timed solves.
    python tools/trace_where.py <trace dir> <pattern>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = next(int(r["Start_Timestamp"]) for r in rows if "k_st_spmv_r" in r["Kernel_Name"])
t1 = max(int(r["End_Timestamp"]) for r in rows if "k_st_spmv_r" in r["Kernel_Name"])
print(f"first ... last k_st_spmv_r: span {(t1 - t0) / 1e6:.1f} ms")
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e6:10.2f} ms  {(e - s) / 1e3:8.1f} us  {'INSIDE' if t0 <= s <= t1 else 'outside'}  {r['Kernel_Name'][:60]}")
