#!/usr/bin/env python3
"""HBM traffic of the sparse-LU kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE - separate passes, both in KB,
FETCH_SIZE doubled: gfx950 tallies a 128-B request as 64 B; MI355X_MICROARCH.md) over the same program, with the kernel durations of
the first pass's kernel trace:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d F -o f -- python3 tools/gc_scaling.py 1024 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d W -o w -- python3 tools/gc_scaling.py 1024 1
    python tools/nd_traffic.py F W > profiles/rNN_nd_traffic.json
Per kernel name: launches, summed bytes, summed duration (counter collection serialises the kernels, so durations are per-kernel
device times without overlap), bytes / duration."""
import csv
import glob
import json
import os
os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")  # PGX_* switches reach the library through the loader's opt-in bridge
import sys
from collections import defaultdict


def counters(d, name):
    acc = defaultdict(float)
    cnt = defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if row["Counter_Name"] == name:
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[k] += float(row["Counter_Value"])
                cnt[k] += 1
    return acc, cnt


def durations(d):
    acc = defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
    return acc


import hashlib
import re

fetch, cnt = counters(sys.argv[1], "FETCH_SIZE")
write, _ = counters(sys.argv[2], "WRITE_SIZE")
dur = durations(sys.argv[1])
out = {}
SOLVE = ("k_nd_fwd", "k_nd_bwd", "k_nd_trsv", "k_nd_gemv", "k_nd_write_x")
tot = {"factor": [0.0, 0.0, 0.0], "solve": [0.0, 0.0, 0.0]}
for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0.0))):
    if not k.startswith("k_nd"):
        continue
    rd, wr = 2.0 * fetch[k] * 1024.0, write.get(k, 0.0) * 1024.0
    out[k] = {"launches": cnt[k], "read_GB": round(rd / 1e9, 3), "written_GB": round(wr / 1e9, 3), "device_ms": round(dur.get(k, 0.0) * 1e3, 2),
              "TB_per_s": round((rd + wr) / dur[k] / 1e12, 2) if dur.get(k) else None}
    t = tot["solve" if k.startswith(SOLVE) else "factor"]
    t[0] += rd
    t[1] += wr
    t[2] += dur.get(k, 0.0)
# how many factorisations / solves the program made, and the arena: from its log (tools/gc_scaling.py / sg_scaling.py print them)
log = ""
for d in sys.argv[1:3]:
    f = d.rstrip("/") + ".log"
    if os.path.exists(f):
        log = open(f).read()
        break
nfac = nsol = None
m = re.search(r"(\d+) factorisations, (\d+) solves", log)
if m:
    nfac, nsol = int(m.group(1)), int(m.group(2))
arena = None
m = re.search(r"sparse LU storage ([0-9.]+) GB", log)
if m:
    arena = float(m.group(1))
so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "proximalgalerkin_amd", "libpgx.so")
doc = {"note": "sums over ONE run of the program named in `program` (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE "
               "doubled per MI355X_MICROARCH.md); counter collection serialises the kernels, so device_ms are per-kernel times without overlap",
       "program": os.environ.get("ND_TRAFFIC_PROGRAM", ""),
       "libpgx_sha256_16": hashlib.sha256(open(so, "rb").read()).hexdigest()[:16] if os.path.exists(so) else None,
       "factorisations": nfac, "solves": nsol, "arena_GB": arena}
if nfac:
    rd, wr, t = tot["factor"]
    doc["per_factorisation"] = {"read_GB": round(rd / nfac / 1e9, 2), "written_GB": round(wr / nfac / 1e9, 2),
                                "traffic_GB": round((rd + wr) / nfac / 1e9, 2), "kernel_ms_serialised": round(t / nfac * 1e3, 2),
                                "traffic_over_arena": round((rd + wr) / nfac / 1e9 / arena, 2) if arena else None}
if nsol:
    rd, wr, t = tot["solve"]
    doc["per_solve"] = {"read_GB": round(rd / nsol / 1e9, 2), "written_GB": round(wr / nsol / 1e9, 2), "traffic_GB": round((rd + wr) / nsol / 1e9, 2),
                        "kernel_ms_serialised": round(t / nsol * 1e3, 2), "TB_per_s": round((rd + wr) / t / 1e12, 2) if t else None}
doc["kernels"] = out
json.dump(doc, sys.stdout, indent=1)
print()
