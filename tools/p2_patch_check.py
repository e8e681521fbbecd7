#!/usr/bin/env python3
"""P2 example 01 through FGMRES + the two-level cycle with the vertex-star patch smoother (pc_type pgx_mg) against the sparse-LU
path: Newton counts per proximal step, Krylov iterations, primal field, time.   python tools/p2_patch_check.py 64 128 [A|B]"""
import os
import sys
import time

os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")  # PGX_* switches reach the library through the loader's opt-in bridge
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from proximalgalerkin_amd import fem  # noqa: E402
from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem  # noqa: E402

BASE = {"ksp_type": "preonly", "ksp_error_if_not_converged": True, "snes_error_if_not_converged": False,
        "snes_linesearch_type": "none", "snes_rtol": 1e-6, "snes_max_it": 100}
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()]
settings = "A" if "A" in sys.argv else "B"
S = {"A": ("constant", 1e5, 1e-6), "B": ("double_exponential", 1e2, 1e-4)}[settings]


def run(N, pc):
    msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
    t0 = time.perf_counter()
    problem, sol, sol_k, alpha = setup_problem(msh, 2, petsc_options=dict(BASE, pc_type=pc) if pc else dict(BASE))
    ts = time.perf_counter() - t0
    lin = []
    orig = problem.solve

    def solve():
        r = orig()
        lin.append(problem.solver.getLinearSolveIterations())
        return r

    problem.solve = solve
    t0 = time.perf_counter()
    try:
        hist = run_outer_loop(problem, sol, sol_k, alpha, 100, S[0], S[1], S[2])
        ok = True
    except Exception as e:  # noqa: BLE001
        hist, ok = {"Newton steps": [str(e)[:60]]}, False
    dt = time.perf_counter() - t0
    x = sol.x.array.copy()
    problem.close()
    return x, hist, lin, dt, ts, ok


for N in sizes:
    mode = None if "--auto" in sys.argv else "pgx_mg"  # auto: patch multigrid first, sparse LU for a solve in which it stagnates
    xm, hm, lm, tm, sm, okm = run(N, mode)
    print(f"N={N} settings {settings} {mode or 'auto'} (patch): {tm:7.2f} s (setup {sm:.1f}) newton {hm['Newton steps']} krylov(last per solve) {lm}", flush=True)
    if "--no-lu" in sys.argv:
        continue
    xl, hl, ll, tl, sl, okl = run(N, "pgx_lu")
    n = len(xl) // 2
    print(f"N={N} settings {settings} pgx_lu        : {tl:7.2f} s (setup {sl:.1f}) newton {hl['Newton steps']}  rel u diff "
          f"{np.linalg.norm(xm[:n] - xl[:n]) / np.linalg.norm(xl[:n]):.2e}", flush=True)
