#!/usr/bin/env python3
"""Per-tree-depth wall time of ONE sparse-LU factorisation from a rocprofv3 kernel trace: pgx_nd_factor processes the dissection
tree depth by depth (deepest first), each depth = prelude on the main stream (memset, scatter, pad, extend-add of the children) +
elimination of the depth's batches on forked streams.  Depths are delimited by the first parent-centric gather of a depth (default) or by the working-buffer memsets (PGX_ND_GATHER=0);
the xadd column is the extend-add / gather kernel time.
    PGX_ND_PREP_AHEAD=0 rocprofv3 --kernel-trace ... (the depth-ahead buffer preparation would blur the boundaries)
    python tools/nd_depth_timeline.py <dir or kernel_trace.csv> [index of the factorisation, default: the last complete one]"""
import csv
import glob
import os
os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")  # PGX_* switches reach the library through the loader's opt-in bridge
import sys

src = sys.argv[1]
f = src if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
with open(f, newline="") as fh:
    for x in csv.DictReader(fh):
        n = x["Kernel_Name"].split("(")[0].replace("void ", "")
        if n.startswith("k_nd") or "fillBuffer" in n:
            def dim(ax):
                g = x.get("Grid_Size_" + ax) or (x.get("Grid_Size") if ax == "X" else 1) or 1
                w = x.get("Workgroup_Size_" + ax) or (x.get("Workgroup_Size") if ax == "X" else 1) or 1
                return int(g) // max(1, int(w))
            rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), n, (dim("X"), dim("Y"), dim("Z"))))
rows.sort()
solve_k = ("k_nd_fwd_assemble", "k_nd_trsv", "k_nd_gemv", "k_nd_bwd_gather", "k_nd_write_x")
segs, cur = [], []
for r in rows:
    if r[2] in solve_k:
        if len(cur) > 50:
            segs.append(cur)
        cur = []
    else:
        cur.append(r)
if len(cur) > 50:
    segs.append(cur)
seg = segs[int(sys.argv[2]) if len(sys.argv) > 2 else -1]
t0 = seg[0][0]
# depth boundaries: a fillBuffer that follows elimination kernels starts a new depth
depths, cur, seen_elim = [], [], False
for r in seg:
    if ("fillBuffer" in r[2] or r[2].startswith("k_nd_gather")) and seen_elim:  # zero fill (push mode) or the first gather (default)
        depths.append(cur)
        cur, seen_elim = [], False
    cur.append(r)
    if r[2] == "k_nd_diag" or r[2].startswith("k_nd_panel") or r[2].startswith("k_nd_gemm") or r[2].startswith("k_nd_leaf"):
        seen_elim = True
depths.append(cur)
print(f"factorisation span {(seg[-1][1] - t0) / 1e6:.1f} ms, {len(depths)} depth groups (deepest first)")
print(f"{'grp':>3s} {'start ms':>9s} {'wall ms':>8s} {'prelude':>8s} {'elim':>8s} {'diag':>5s} {'panel ms':>9s} {'gemm ms':>8s} {'diag ms':>8s} {'xadd ms':>8s}")
for i, d in enumerate(depths):
    s, e = d[0][0], max(r[1] for r in d)
    first = next((r[0] for r in d if r[2] == "k_nd_diag" or r[2].startswith("k_nd_leaf")), e)
    ks = lambda pre: sum(r[1] - r[0] for r in d if r[2].startswith(pre)) / 1e6  # noqa: E731
    print(f"{i:3d} {(s - t0) / 1e6:9.2f} {(e - s) / 1e6:8.2f} {(first - s) / 1e6:8.2f} {(e - first) / 1e6:8.2f} {sum(r[2] == 'k_nd_diag' for r in d):5d} "
          f"{ks('k_nd_panel'):9.2f} {ks('k_nd_gemm'):8.2f} {ks('k_nd_diag'):8.2f} {ks('k_nd_extend') + ks('k_nd_gather'):8.2f}")

# PGX_ND_LAUNCHES=g0,g1,...: every elimination launch of those depth groups (workgroup grid, duration)
for gi in [int(t) for t in os.environ.get("PGX_ND_LAUNCHES", "").split(",") if t]:
    print(f"--- launches of depth group {gi}")
    for r in depths[gi]:
        if r[2].startswith("k_nd"):
            print(f"  {(r[0] - t0) / 1e6:9.3f} ms  {r[2]:24s} grid {r[3]}  {(r[1] - r[0]) / 1e3:9.1f} us")
