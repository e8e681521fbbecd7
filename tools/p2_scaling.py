"""One full LVPP solve (settings B) of the ex 01 obstacle problem at degree `p` on an N x N mesh with a chosen
preconditioner: python tools/p2_scaling.py N [p] [pgx_lu|pgx_mg|auto] [double_exponential|constant] [reps]
(constant = settings A: alpha = 1, tol 1e-6)"""
import sys
import time

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from proximalgalerkin_amd import fem  # noqa: E402
from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem  # noqa: E402

N = int(sys.argv[1])
p = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pc = sys.argv[3] if len(sys.argv) > 3 else "auto"
scheme = sys.argv[4] if len(sys.argv) > 4 else "double_exponential"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
opts = {"ksp_type": "preonly", "pc_type": pc, "snes_error_if_not_converged": False, "snes_linesearch_type": "none",
        "snes_rtol": 1e-6, "snes_max_it": 100}
msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
t = time.perf_counter()
problem, sol, sol_k, alpha = setup_problem(msh, p, petsc_options=opts)
print(f"N={N} p={p} pc={pc} dofs={problem.ndofs} setup {time.perf_counter() - t:.2f}s", flush=True)
problem.profile(True)
for rep in range(reps):
    t = time.perf_counter()
    hist = (run_outer_loop(problem, sol, sol_k, alpha, 100, "double_exponential", 1e2, 1e-4, verbose=N >= 1024) if scheme == "double_exponential"
            else run_outer_loop(problem, sol, sol_k, alpha, 100, "constant", 1e5, 1e-6, verbose=N >= 1024))
    dt = time.perf_counter() - t
    print(f"  run {rep}: {dt * 1e3:.0f} ms  Newton {hist['Newton steps']} (sum {sum(hist['Newton steps'])}) "
          f"reason {problem.solver.getConvergedReason()}  lin its last {problem.solver.ksp._its}", flush=True)
print("  phases ms [resid, jac, setup/factor, spmv, pc, orth, obs, total]:", problem.profile(reset=True), flush=True)
