"""Overlap analysis of a rocprofv3 kernel trace (CSV, optionally .gz) of a run that uses the sparse direct solver:
per factorisation / solve the wall span, the busy time (union of kernel intervals), the summed kernel time and the
stream -> hardware queue mapping.  python tools/trace_overlap.py trace.csv[.gz]"""
import collections
import csv
import gzip
import sys


def load(f):
    op = gzip.open if f.endswith(".gz") else open
    rows = []
    with op(f, "rt") as fh:
        for x in csv.DictReader(fh):
            rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"].split("(")[0].replace("void ", ""),
                         x["Queue_Id"], x["Stream_Id"]))
    rows.sort()
    return rows


def stats(seg):
    busy, cs, ce = 0, seg[0][0], seg[0][1]
    for s, e, *_ in seg[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    q = collections.Counter((r[4], r[3]) for r in seg)
    return (seg[-1][1] - seg[0][0]) / 1e6, busy / 1e6, sum(e - s for s, e, *_ in seg) / 1e6, dict(q)


if __name__ == "__main__":
    rows = [r for r in load(sys.argv[1]) if r[2].startswith("k_nd") or "fillBuffer" in r[2]]
    solve_k = ("k_nd_fwd_assemble", "k_nd_trsv", "k_nd_gemv", "k_nd_bwd_gather", "k_nd_write_x")
    segs, cur, kind = [], [], None
    for r in rows:
        k = "solve" if r[2] in solve_k else "factor"
        if k != kind and cur:
            segs.append((kind, cur))
            cur = []
        kind = k
        cur.append(r)
        if r[2] == "k_nd_write_x":
            segs.append((kind, cur))
            cur, kind = [], None
    if cur:
        segs.append((kind, cur))
    for kind in ("factor", "solve"):
        ss = [stats(s) for k, s in segs if k == kind and len(s) > 10]
        if not ss:
            continue
        n = len(ss)
        print(f"{kind}: {n} calls, mean span {sum(x[0] for x in ss) / n:.2f} ms, busy {sum(x[1] for x in ss) / n:.2f} ms, "
              f"kernel sum {sum(x[2] for x in ss) / n:.2f} ms; (stream, queue) -> kernels of the last call: {ss[-1][3]}")
