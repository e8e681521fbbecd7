#!/usr/bin/env python3
"""Goldens of example 01 at the sizes of the BASELINE configs, from the CPU oracle with its nested-dissection LU.

    python tools/make_golden_nd.py p1:1024 p1:2048 p2:512        (build container; 10 min / 1-2 h / 20 min)

The exact-Newton oracle (oracle/pg_oracle.py: the loop of /root/reference/examples/01_obstacle_problem/obstacle_pg.py:173-227,
SNES newtonls + `pc_type lu`) runs settings B (the reference's CI settings, compare_all.py:80-87) with oracle/nd_lu.py as its
direct solver and writes tests/golden/obstacle_p{k}_n{N}_settingsB_nd.npz:

* per-proximal-step Newton counts, the observable columns, alpha values;
* the final primal field on a sub-lattice (`u_sample`, every `stride`-th vertex in x and y: 257^2 values) AND the sums of u over
  the stride x stride vertex blocks (`u_blocksum`: every vertex contributes, so an error anywhere in the field shows up), its
  2-norm and maximum - a 4.2 M-vertex field itself would be 30 MB per fixture;
* the largest relative residual the direct solver left after iterative refinement (`lin_relres_max`).

tests/test_gpu_golden.py compares the HIP path with these at identical Newton counts and u <= 1e-10.  Also appends the timing of the
run to profiles/r03_cpu_ladder_nd.json (seconds per Newton step of the oracle: assembly + factorisation + solves).
These fixtures are generated from the ORACLE (parity unpinned w.r.t. FEniCSx itself, oracle/pg_oracle.py header).
"""
import json
import pathlib
import platform
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import nd_lu as ND  # noqa: E402
from oracle import pg_oracle as O  # noqa: E402

GOLD = ROOT / "tests" / "golden"
LADDER = ROOT / "profiles" / "r03_cpu_ladder_nd.json"


def lattice_fingerprint(u_grid, stride):
    """u_grid: (M, M) vertex values, M = k*stride + 1.  Returns (sample, blocksum)."""
    M = u_grid.shape[0]
    nb = (M - 1) // stride
    sample = u_grid[::stride, ::stride].copy()
    core = u_grid[: nb * stride, : nb * stride].reshape(nb, stride, nb, stride).sum(axis=(1, 3))
    return sample, core


def one(degree, N):
    coords, cells = O.create_rectangle(N, N)
    t0 = time.perf_counter()
    if degree == 1:
        prob = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    else:
        prob = O.ObstacleLagrange(coords, cells, degree=degree)
    print(f"P{degree} {N}^2: problem built in {time.perf_counter() - t0:.1f} s, {2 * prob.n} unknowns", flush=True)
    ls = ND.NDLinearSolve(*ND.nodes_of_problem(prob), verbose=True)
    relres = []
    ls_call = ls.__call__

    def solve(J, rhs):
        x = ls_call(J, rhs)
        relres.append(ls.last_relres)
        return x

    log = O.NewtonLog()
    t0 = time.perf_counter()
    x, hist = O.solve_problem(prob, 100, "double_exponential", 1e2, 1e-4, linear_solve=solve, log=log, verbose=True)
    wall = time.perf_counter() - t0
    newton = int(sum(hist["Newton steps"]))
    n = prob.n
    nv = (N + 1) ** 2
    M = N + 1
    stride = max(1, N // 256)
    ug = x[:nv].reshape(M, M)  # vertex dofs come first for both degrees; row-major lattice (create_rectangle)
    sample, blocksum = lattice_fingerprint(ug, stride)
    out = GOLD / f"obstacle_p{degree}_n{N}_settingsB_nd.npz"
    np.savez_compressed(out, N=N, degree=degree, stride=stride, u_sample=sample, u_blocksum=blocksum,
                        u_norm2=float(np.linalg.norm(x[:n])), u_vertex_norm2=float(np.linalg.norm(x[:nv])),
                        u_max=float(x[:n].max()), psi_min=float(x[n:].min()), lin_relres_max=float(max(relres)),
                        **{("hist_" + k.replace(" ", "_")): np.asarray(v) for k, v in hist.items()})
    sym = ls.nd.symbolic_s
    rec = {"degree": degree, "N": N, "unknowns": 2 * n, "newton_steps": newton, "wall_s": wall,
           "s_per_newton_step": (wall - sym) / newton, "symbolic_s": sym, "t_factor": ls.t_factor, "t_solve_refine": ls.t_solve,
           "t_jacobian": log.t_jacobian, "t_residual": log.t_residual, "factor_flops": ls.nd.flops,
           "factor_entries": ls.nd.factor_entries, "blas_threads": "default (all cores of the build container: 8)",
           "host": platform.processor() or platform.machine()}
    print(json.dumps(rec), flush=True)
    recs = json.loads(LADDER.read_text())["golden_runs"] if LADDER.exists() else []
    recs = [r for r in recs if (r["degree"], r["N"]) != (degree, N)] + [rec]
    doc = json.loads(LADDER.read_text()) if LADDER.exists() else {}
    doc["golden_runs"] = sorted(recs, key=lambda r: (r["degree"], r["N"]))
    LADDER.write_text(json.dumps(doc, indent=1))
    print(f"wrote {out} ({out.stat().st_size} bytes)", flush=True)


if __name__ == "__main__":
    for a in sys.argv[1:]:
        d, _, N = a.partition(":")
        one(int(d[1:]), int(N))
