"""Sensitivity of the final primal field to the Newton linear-solve tolerance (GPU). The LU oracle cannot
reach 2048^2; a ksp_rtol=1e-13 run is the stand-in reference."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proximalgalerkin_amd import fem
from proximalgalerkin_amd.obstacle import setup_problem, run_outer_loop
N = int(sys.argv[1])
msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
n = msh.num_vertices
ref = None
for spec in sys.argv[2:]:
    opts = {"snes_linesearch_type": "none", "snes_rtol": 1e-6, "snes_max_it": 100, "snes_error_if_not_converged": True}
    for kv in spec.split(","):
        k, v = kv.split("="); opts[k] = float(v) if any(c in v for c in ".e") else int(v)
    problem, sol, sol_k, alpha = setup_problem(msh, 1, petsc_options=opts)
    run_outer_loop(problem, sol, sol_k, alpha, 500, "double_exponential", 1e2, 1e-4)  # warm-up
    t = time.perf_counter()
    h = run_outer_loop(problem, sol, sol_k, alpha, 500, "double_exponential", 1e2, 1e-4)
    dt = time.perf_counter() - t
    x = sol.x.array.copy(); problem.close()
    if ref is None: ref = x
    du = np.linalg.norm(x[:n] - ref[:n]) / np.linalg.norm(ref[:n]); dp = np.linalg.norm(x[n:] - ref[n:]) / np.linalg.norm(ref[n:])
    print(f"{spec:45s} newton {h['Newton steps']} time {dt*1e3:7.1f} ms  rel-L2 vs first: u {du:.2e} psi {dp:.2e}", flush=True)
