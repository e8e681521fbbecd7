#!/usr/bin/env python3
"""Sharded vs single-handle solve at full size on ONE GPU: R strips driven by R threads through the in-process
transport (the algorithm the RCCL launch runs; timings here are NOT multi-GPU timings).
usage: sharded_check.py N R [settings] [dist_levels]"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from bench import SETTINGS  # noqa: E402
from proximalgalerkin_amd import comm as pcomm  # noqa: E402
from proximalgalerkin_amd import fem  # noqa: E402
from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem  # noqa: E402

N, R = int(sys.argv[1]), int(sys.argv[2])
S = SETTINGS[sys.argv[3] if len(sys.argv) > 3 else "B"]
LEV = int(sys.argv[4]) if len(sys.argv) > 4 else 0
DOM = ((-1.0, -1.0), (1.0, 1.0))


def solve(comm):
    msh = fem.create_rectangle(DOM, (N, N), comm=comm, dist_levels=LEV)
    problem, sol, sol_k, alpha = setup_problem(msh, 1)
    t0 = time.perf_counter()
    hist = run_outer_loop(problem, sol, sol_k, alpha, S["max_outer"], S["alpha_scheme"], S["alpha_max"], S["tol_exit"])
    dt = time.perf_counter() - t0
    x = sol.x.array.copy()
    rng = problem.owned_range()
    lin = problem.solver.getLinearSolveIterations()
    problem.close()
    return x, hist, dt, msh.partition, rng, lin


xg, hg, dtg, _, _, ling = solve(None)
print(f"single : {dtg*1e3:8.1f} ms  Newton {hg['Newton steps']}  last KSP its {ling}")
out = [None] * R
comms = pcomm.local_group(R)
th = [threading.Thread(target=lambda r=r: out.__setitem__(r, solve(comms[r]))) for r in range(R)]
[t.start() for t in th]
[t.join() for t in th]
sx, ng = N + 1, (N + 1) ** 2
u = np.full(ng, np.nan)
for x, h, dt, part, (off, cnt), lin in out:
    assert h["Newton steps"] == hg["Newton steps"], (h["Newton steps"], hg["Newton steps"])
    u[part.own0 * sx:part.own0 * sx + cnt] = x[off:off + cnt]
print(f"sharded: R={R} dist_levels={out[0][3].dist_levels}  Newton counts identical, last KSP its {out[0][5]}, "
      f"|u - u_single|/|u| = {np.linalg.norm(u - xg[:ng]) / np.linalg.norm(xg[:ng]):.2e}  "
      f"(threads on one GPU: {max(o[2] for o in out)*1e3:.0f} ms, not a multi-GPU timing)")
