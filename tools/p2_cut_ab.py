"""Example 01 P2 on an N x N mesh through the sparse LU, settings A, six proximal steps, with per-phase device times - run under
PGX_ND_CUT_GB=96 / -1 to compare the subtree-sequenced schedule of large factorisations with the plain one:
    PGX_ND_CUT_GB=-1 python tools/p2_cut_ab.py 2048"""
import os
os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")  # PGX_* switches reach the library through the loader's opt-in bridge
import sys, time
sys.path.insert(0, ".")
from proximalgalerkin_amd import fem
from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem
N = int(sys.argv[1])
msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
t = time.perf_counter()
problem, sol, sol_k, alpha = setup_problem(msh, 2, petsc_options={"snes_linesearch_type": "none", "snes_rtol": 1e-6, "snes_max_it": 100})
print(f"setup {time.perf_counter() - t:.1f} s", flush=True)
problem.profile(enable=True, reset=True)
t = time.perf_counter()
h = run_outer_loop(problem, sol, sol_k, alpha, 6, "constant", 1e5, 1e-6)
dt = time.perf_counter() - t
print(f"P2 {N}: Newton {sum(h['Newton steps'])} in {dt:.2f} s; phases ms:", {k: round(v, 1) for k, v in problem.profile().items()}, flush=True)
