#!/usr/bin/env python3
"""CPU ladders of the example 06 / example 02 oracles (numpy assembly + SuperLU exact Newton, 1 thread): seconds per Newton step
at a series of mesh sizes and the exponent of t = c n^p, written to profiles/r02_cpu_ladder_ex06.json / _ex02.json.  bench.py's
cpu_baseline leg extrapolates its live small-mesh measurement to the benchmarked mesh with that exponent (and says so).
    python tools/cpu_ladder.py ex06 32 64 96 128        python tools/cpu_ladder.py ex02 8 12 16 20"""
import json
import pathlib
import platform
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import pg_oracle as O  # noqa: E402


def run(kind, n):
    t0 = time.perf_counter()
    if kind == "ex06":
        from oracle import gc_oracle as G

        c, e = O.create_rectangle(n, n, (0.0, 0.0), (1.0, 1.0))
        prob = G.GradientConstraintP2(c, e)
        _, its, _ = G.solve_problem(prob)
        newton, unknowns = int(np.sum(its)), int(prob.ntot)
    else:
        from oracle import sg_oracle as S

        c, t = S.create_unit_cube_tets(n, n, n)
        prob = S.SignoriniP1(c, t, S.boundary_facets_where(c, t, lambda x: np.isclose(x[:, 2], 0.0)), np.flatnonzero(np.isclose(c[:, 2], 1.0)))
        _, _, its = S.solve_contact_problem(prob)
        newton, unknowns = int(sum(its)), int(prob.ntot)
    wall = time.perf_counter() - t0
    rec = {"N": n, "unknowns": unknowns, "newton_steps": newton, "wall_s": wall, "s_per_newton_step": wall / newton}
    print(json.dumps(rec), flush=True)
    return rec


def main():
    kind = sys.argv[1]
    out = ROOT / "profiles" / f"r02_cpu_ladder_{kind}.json"
    recs = json.loads(out.read_text())["points"] if out.exists() else []
    for a in sys.argv[2:]:
        r = run(kind, int(a))
        recs = sorted([p for p in recs if p["N"] != r["N"]] + [r], key=lambda p: p["N"])
        d = {"what": f"CPU oracle of {kind} (numpy assembly + SuperLU exact Newton, 1 thread), the reference's default settings, full LVPP run "
                     "(setup included)", "host": platform.processor() or platform.machine(), "points": recs}
        if len(recs) >= 2:
            p, c = np.polyfit(np.log([q["N"] for q in recs]), np.log([q["s_per_newton_step"] for q in recs]), 1)
            d["fit"] = {"model": "s_per_newton_step = c * N^p", "p": float(p), "c": float(np.exp(c))}
        out.write_text(json.dumps(d, indent=1))


if __name__ == "__main__":
    main()
