#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output (the *_counter_collection.csv files under one or more output directories) per kernel:

    python tools/pmc_summary.py --kernel k_bspmv_stream --out profiles/r02_spmv_pmc_traffic.json \\
           --traffic --cells 2048 --algorithmic-bytes 973570116 gpurun_out/pmc_fetch gpurun_out/pmc_write

Without --traffic: mean per-launch value of every counter found for kernels whose name contains --kernel (optionally only
launches with --grid GRID_SIZE), as JSON.  With --traffic: the HBM bytes per launch as MI355X_MICROARCH.md prescribes -
FETCH_SIZE and WRITE_SIZE collected in SEPARATE passes, both in KB, FETCH_SIZE doubled (gfx950 tallies a 128-B request as 64 B).
"""
import argparse
import csv
import datetime
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(dirs, kernel, grid=None):
    acc = defaultdict(list)
    names = set()
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    kn = row.get("Kernel_Name", "")
                    if kernel not in kn:
                        continue
                    if grid is not None and int(row.get("Grid_Size", "0")) != grid:
                        continue
                    names.add(kn)
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc, sorted(names)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--grid", type=int, default=None)
    ap.add_argument("--out", required=True)
    ap.add_argument("--traffic", action="store_true")
    ap.add_argument("--cells", type=int, default=0)
    ap.add_argument("--algorithmic-bytes", type=float, default=0.0)
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    acc, names = collect(a.dirs, a.kernel, a.grid)
    if not acc:
        sys.exit(f"no counter rows for a kernel containing {a.kernel!r} under {a.dirs}")
    so = os.path.join(ROOT, "proximalgalerkin_amd", "libpgx.so")
    out = {"file": os.path.relpath(a.out, ROOT) if os.path.isabs(a.out) else a.out, "kernel": a.kernel,
           "kernel_names": [n[:160] for n in names], "date": datetime.date.today().isoformat(),
           "libpgx_sha256_16": hashlib.sha256(open(so, "rb").read()).hexdigest()[:16] if os.path.exists(so) else None,
           "counters": {k: {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "launches": len(v)} for k, v in sorted(acc.items())}}
    if a.grid is not None:
        out["grid_size"] = a.grid
    if a.note:
        out["note"] = a.note
    if a.traffic:
        fetch = out["counters"]["FETCH_SIZE"]["mean"] * 1024.0 * 2.0
        write = out["counters"]["WRITE_SIZE"]["mean"] * 1024.0
        out.update({"cells": a.cells, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
                    "hbm_traffic_bytes_per_launch": fetch + write,
                    "algorithmic_bytes_per_launch": a.algorithmic_bytes or None,
                    "traffic_over_algorithmic": (fetch + write) / a.algorithmic_bytes if a.algorithmic_bytes else None,
                    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (TCC slots), KB per dispatch, "
                              "mean over the launches of this kernel; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports "
                              "half of a coalesced streaming read)"})
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "kernel_names"}, indent=1))


if __name__ == "__main__":
    main()
