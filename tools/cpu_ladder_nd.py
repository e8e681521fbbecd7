#!/usr/bin/env python3
"""CPU ladder of the oracle with its nested-dissection LU (oracle/nd_lu.py), ONE BLAS thread - the configuration bench.py's
cpu_baseline leg measures live on the GPU box: seconds per Newton step (numpy assembly + numeric factorisation + solves with
refinement; symbolic analysis and mesh setup untimed) for the first Newton steps of settings B at a series of mesh sizes, and the
exponent of t = c N^p.  bench.py carries its live 1024^2 measurement to the benchmarked 2048^2 mesh with that exponent (2-D nested
dissection: about N^3 in flops, N^2 log N in storage), and says so.   python tools/cpu_ladder_nd.py 256 512 1024  (build container)
`--ex06 N...` does the same for example 06 (oracle/gc_oracle.py: P2 / vector-P1, alpha = 1 first proximal step; bench.py
--workload ex06 measures the 256^2 point live) -> profiles/r03_cpu_ladder_ex06_nd.json."""
import json
import pathlib
import platform
import sys
import time

import numpy as np
from threadpoolctl import threadpool_limits

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import nd_lu as ND  # noqa: E402
from oracle import pg_oracle as O  # noqa: E402

OUT = ROOT / "profiles" / "r03_cpu_ladder_nd.json"
STEPS = 3


def one(N):
    coords, cells = O.create_rectangle(N, N)
    prob = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    x = np.zeros(2 * prob.n)
    ls = ND.NDLinearSolve(*ND.nodes_of_problem(prob))
    ls.nd = ND.NDLU(prob.jacobian(x, 1.0), ls.node_of_dof, ls.node_coords, ls.leaf_nodes)
    ls.nd.aoff = np.concatenate(([0], np.cumsum(ls.nd.p * ls.nd.p + 2 * ls.nd.p * ls.nd.b)))
    ls.nd.arena = np.ones(int(ls.nd.aoff[-1]))
    ND.MAX_THREADS = 1
    with threadpool_limits(1):
        t0 = time.perf_counter()
        F = prob.residual(x, x * 0, 1.0)
        xk = x.copy()
        for _ in range(STEPS):
            x = x + ls(prob.jacobian(x, 1.0), -F)
            F = prob.residual(x, xk, 1.0)
        dt = time.perf_counter() - t0
    ND.MAX_THREADS = 0
    rec = {"N": N, "unknowns": 2 * prob.n, "newton_steps": STEPS, "s_per_newton_step": dt / STEPS, "factor_s_per_step": ls.t_factor / STEPS,
           "solve_refine_s_per_step": ls.t_solve / STEPS, "factor_gflop": ls.nd.flops / 1e9, "symbolic_s_untimed": ls.nd.symbolic_s,
           "factor_storage_GB": 8e-9 * ls.nd.factor_entries}
    print(json.dumps(rec), flush=True)
    return rec


def one_ex06(N):
    from oracle import gc_oracle as G

    c, e = O.create_rectangle(N, N, (0.0, 0.0), (1.0, 1.0))
    prob = G.GradientConstraintP2(c, e)
    ls = ND.NDLinearSolve(*ND.nodes_of_problem(prob))
    ND.MAX_THREADS = 1
    st = {"n": 0, "t0": None}

    class Done(Exception):
        pass

    def solve(J, rhs):
        dx = ls(J, rhs)
        if st["t0"] is None:  # first call: symbolic analysis + first touch of the arena, untimed
            st["t0"] = time.perf_counter()
            ls.t_factor = ls.t_solve = 0.0
            return dx
        st["n"] += 1
        st["dt"] = time.perf_counter() - st["t0"]
        if st["n"] >= STEPS + 1:
            raise Done
        return dx

    with threadpool_limits(1):
        try:
            G.solve_problem(prob, linear_solve=solve)
        except Done:
            pass
    ND.MAX_THREADS = 0
    n = st["n"]
    rec = {"N": N, "unknowns": prob.ntot, "newton_steps": n, "s_per_newton_step": st["dt"] / n, "factor_s_per_step": ls.t_factor / n,
           "solve_refine_s_per_step": ls.t_solve / n, "factor_gflop": ls.nd.flops / 1e9, "symbolic_s_untimed": ls.nd.symbolic_s,
           "factor_storage_GB": 8e-9 * ls.nd.factor_entries}
    print(json.dumps(rec), flush=True)
    return rec


def one_ex02(m):
    """example 02 (oracle/sg_oracle.py, unit cube of m^3 x 6 P1 tetrahedra): the full LVPP run (6 Newton steps), first linear solve
    (symbolic analysis) untimed."""
    from oracle import sg_oracle as S

    c, t = S.create_unit_cube_tets(m, m, m)
    prob = S.SignoriniP1(c, t, S.boundary_facets_where(c, t, lambda x: np.isclose(x[:, 2], 0.0)), np.flatnonzero(np.isclose(c[:, 2], 1.0)))
    ls = ND.NDLinearSolve(*ND.nodes_of_problem(prob))
    ND.MAX_THREADS = 1
    st = {"n": 0, "t0": None, "dt": 0.0}

    def solve(J, rhs):
        dx = ls(J, rhs)
        if st["t0"] is None:
            st["t0"] = time.perf_counter()
            ls.t_factor = ls.t_solve = 0.0
            return dx
        st["n"] += 1
        st["dt"] = time.perf_counter() - st["t0"]
        return dx

    with threadpool_limits(1):
        S.solve_contact_problem(prob, linear_solve=solve)
    ND.MAX_THREADS = 0
    n = st["n"]
    rec = {"N": m, "unknowns": prob.ntot, "newton_steps": n, "s_per_newton_step": st["dt"] / n, "factor_s_per_step": ls.t_factor / n,
           "solve_refine_s_per_step": ls.t_solve / n, "factor_gflop": ls.nd.flops / 1e9, "symbolic_s_untimed": ls.nd.symbolic_s,
           "factor_storage_GB": 8e-9 * ls.nd.factor_entries}
    print(json.dumps(rec), flush=True)
    return rec


def main():
    global OUT
    argv = sys.argv[1:]
    ex06 = "--ex06" in argv
    ex02 = "--ex02" in argv
    if ex06:
        argv.remove("--ex06")
        OUT = ROOT / "profiles" / "r03_cpu_ladder_ex06_nd.json"
    if ex02:
        argv.remove("--ex02")
        OUT = ROOT / "profiles" / "r03_cpu_ladder_ex02_nd.json"
    doc = json.loads(OUT.read_text()) if OUT.exists() else {}
    pts = doc.get("points", [])
    for a in argv:
        r = one_ex02(int(a)) if ex02 else one_ex06(int(a)) if ex06 else one(int(a))
        pts = sorted([p for p in pts if p["N"] != r["N"]] + [r], key=lambda p: p["N"])
    doc["what"] = ("CPU oracle of example 02 (oracle/sg_oracle.py, Signorini contact on N^3 x 6 P1 tetrahedra) with the nested-dissection multifrontal "
                   "LU (oracle/nd_lu.py), 1 BLAS thread: seconds per Newton step of the full LVPP run after the first linear solve (assembly + "
                   "numeric factorisation + solves with refinement; symbolic analysis untimed)") if ex02 else ("CPU oracle of example 06 (oracle/gc_oracle.py, P2 / vector-P1 on the unit square) with the nested-dissection multifrontal LU "
                   "(oracle/nd_lu.py), 1 BLAS thread: seconds per Newton step over the first Newton steps after the first (assembly + numeric "
                   "factorisation + solves with refinement; symbolic analysis untimed)") if ex06 else ("CPU oracle with the nested-dissection multifrontal LU (oracle/nd_lu.py), 1 BLAS thread, P1 settings B on [-1,1]^2: "
                   "seconds per Newton step over the first Newton steps (assembly + numeric factorisation + solves; symbolic analysis untimed)")
    doc["host"] = platform.processor() or platform.machine()
    doc["points"] = pts
    if len(pts) >= 2:
        p, c = np.polyfit(np.log([q["N"] for q in pts]), np.log([q["s_per_newton_step"] for q in pts]), 1)
        doc["fit"] = {"model": "s_per_newton_step = c * N^p", "p": float(p), "c": float(np.exp(c))}
    OUT.write_text(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
