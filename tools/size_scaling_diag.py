"""Why do the reference's CI settings (settings B: double-exponential alpha, alpha_max 1e2, tol 1e-4) cost so much more per Newton
step above 2048^2 - Newton overshoot of the undamped iteration or collapse of the multigrid preconditioner?  
Per proximal step of ONE LVPP solve: alpha, SNES reason, Newton steps, Krylov iterations (total and per Newton step), wall time -
once with the default preconditioner (FGMRES + single-precision geometric multigrid) and, with `--lu`, once more with the sparse LU
(pc_type pgx_lu: every linear solve exact) on the same mesh.  Same Newton counts with both => the extra cost is Newton's (the
linear solves are fine); fewer Newton steps / no failure with LU => the cycle is what gives way.
    python tools/size_scaling_diag.py 3072 [--lu] [--settings A|B]"""
import os
os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proximalgalerkin_amd import fem  # noqa: E402
from proximalgalerkin_amd.obstacle import COLUMNS, alpha_update, setup_problem  # noqa: E402,F401
from proximalgalerkin_amd.problem import ConvergenceError  # noqa: E402

SET = {"A": ("constant", 1e5, 1e-6, 100), "B": ("double_exponential", 1e2, 1e-4, 100)}


def run(N, settings, pc):
    rule, amax, tol, kmax = SET[settings]
    msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
    opts = {"ksp_type": "preonly", "pc_type": pc, "snes_linesearch_type": "none", "snes_rtol": 1e-6, "snes_max_it": 100,
            "snes_error_if_not_converged": False, "ksp_error_if_not_converged": False}
    problem, sol, sol_k, alpha = setup_problem(msh, 1, petsc_options=opts)
    problem.zero_state()
    alpha_k, tot_n, tot_k = 1, 0, 0
    print(f"{N}^2 settings {settings}, preconditioner {pc}", flush=True)
    print(f"  {'step':>4s} {'alpha':>10s} {'reason':>6s} {'Newton':>6s} {'Krylov':>6s} {'per Newton':>10s} {'ms':>8s} {'increment':>11s}", flush=True)
    for k in range(kmax):
        alpha.value, alpha_k = alpha_update(rule, k, alpha_k, amax, current=alpha.value)
        t = time.perf_counter()
        try:
            problem.solve()
        except ConvergenceError as e:
            print(f"  {k + 1:4d} {alpha.value:10.3e}  {e}", flush=True)
            break
        ms = 1e3 * (time.perf_counter() - t)
        rsn, n, lin = problem.solver.getConvergedReason(), problem.solver.getIterationNumber(), problem.solver.ksp.getIterationNumber()
        inc = problem.observables()[4]
        tot_n += n
        tot_k += lin
        print(f"  {k + 1:4d} {alpha.value:10.3e} {rsn:6d} {n:6d} {lin:6d} {lin / max(n, 1):10.1f} {ms:8.1f} {inc:11.3e}", flush=True)
        if rsn <= 0:
            print("  -> SNES diverged: stop", flush=True)
            break
        if inc < tol:
            break
        sol_k.x.assign_from(sol.x)
    print(f"  total: {tot_n} Newton steps, {tot_k} Krylov iterations ({tot_k / max(tot_n, 1):.1f} per Newton step)", flush=True)
    problem.close()


if __name__ == "__main__":
    a = sys.argv[1:]
    st = a[a.index("--settings") + 1] if "--settings" in a else "B"
    N = int(a[0])
    run(N, st, "pgx_mg")
    if "--lu" in a:
        run(N, st, "pgx_lu")
