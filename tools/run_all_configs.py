"""BASELINE.json configs on ONE MI355X, one line each (configs 3 and 5 name multi-GPU launches - the single-GPU numbers are
the N = 1 point of those series):
python tools/run_all_configs.py [--full] > profiles/rNN_all_configs.txt
--full adds config 3 at its full size (2048^2 P2, 33.6 M unknowns, ~4 minutes incl. the symbolic phase)."""
import sys
import time

import numpy as np

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from oracle import pg_oracle as O  # noqa: E402  (config 1 is the reference-side CPU case)
from proximalgalerkin_amd import fem  # noqa: E402
from proximalgalerkin_amd import signorini as sg  # noqa: E402
from proximalgalerkin_amd.gradient_constraint import solve_problem as gc_solve  # noqa: E402
from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem  # noqa: E402


def obstacle(N, degree, settings):
    scheme, amax, tol = settings
    msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
    problem, sol, sol_k, alpha = setup_problem(msh, degree, petsc_options={"snes_linesearch_type": "none", "snes_rtol": 1e-6,
                                                                          "snes_max_it": 100})
    run_outer_loop(problem, sol, sol_k, alpha, 500, scheme, amax, tol)  # warm-up
    t = time.perf_counter()
    h = run_outer_loop(problem, sol, sol_k, alpha, 500, scheme, amax, tol)
    dt = time.perf_counter() - t
    reason = problem.solver.getConvergedReason()
    x = sol.x.array.copy()
    problem.close()
    return h, dt, reason, x


B = ("double_exponential", 1e2, 1e-4)
A = ("constant", 1e5, 1e-6)
# config 1: examples/01 on 64x64 P1 - CPU oracle beside the HIP path
coords, cells = O.create_rectangle(64, 64)
prob = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(64, 64))
t = time.perf_counter()
xr, hr = O.solve_problem(prob, 500, *B)
tc = time.perf_counter() - t
h, dt, _, x = obstacle(64, 1, B)
print(f"config 1  ex01 64^2 P1: oracle (CPU, SuperLU) {sum(hr['Newton steps'])} Newton its in {tc:.2f} s; HIP {sum(h['Newton steps'])} its in "
      f"{dt * 1e3:.0f} ms; rel L2(u) HIP vs oracle {np.linalg.norm(x[:prob.n] - xr[:prob.n]) / np.linalg.norm(xr[:prob.n]):.1e}", flush=True)
h, dt, _, _ = obstacle(2048, 1, B)
print(f"config 2  ex01 2048^2 P1 (settings B): {sum(h['Newton steps'])} Newton its, {h['outer_iterations']} proximal its in {dt * 1e3:.0f} ms "
      f"= {sum(h['Newton steps']) / dt:.1f} Newton it/s", flush=True)
sizes = [(512, B, "B"), (1024, A, "A: constant alpha")] + ([(2048, A, "A: constant alpha")] if "--full" in sys.argv else [])
for N, S, tag in sizes:
    h, dt, reason, _ = obstacle(N, 2, S)
    print(f"config 3' ex01 {N}^2 P2 (settings {tag}; single GPU, patch-smoother multigrid with sparse-LU fallback; the 8-GPU case is the sharded path, bench.py --degree 2 --gpus 8): "
          f"{sum(h['Newton steps'])} Newton its in {dt:.2f} s, last SNES reason {reason}", flush=True)
t = time.perf_counter()
its, diffs = gc_solve(1024, 1024, verbose=False)
dt = time.perf_counter() - t
print(f"config 4  ex06 1024^2 P2/vector-P1: {len(its)} LVPP its, {int(its.sum())} Newton its in {dt:.1f} s incl. setup", flush=True)
mesh = sg.create_unit_cube(70, 70, 70)
mt, bcs = sg.native_tags(mesh)
t = time.perf_counter()
it, iters = sg.solve_contact_problem(mesh, mt, bcs, verbose=False)
dt = time.perf_counter() - t
print(f"config 5  ex02 70^3 x 6 = {mesh.cells.shape[0]} tets P1 (single GPU; 4-GPU launch = pgx_sg_create_dist): {it} LVPP its, {sum(iters)} Newton "
      f"its in {dt:.1f} s incl. setup", flush=True)
