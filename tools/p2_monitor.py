import os, sys
os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")
sys.path.insert(0, "/root/repo")
from proximalgalerkin_amd import fem
from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem
N = int(sys.argv[1])
opts = {"ksp_type": "preonly", "pc_type": "pgx_mg", "ksp_error_if_not_converged": True, "snes_error_if_not_converged": False,
        "snes_linesearch_type": "none", "snes_rtol": 1e-6, "snes_max_it": 100, "snes_monitor": True, "ksp_max_it": 400}
msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
problem, sol, sol_k, alpha = setup_problem(msh, 2, petsc_options=opts)
try:
    hist = run_outer_loop(problem, sol, sol_k, alpha, 100, "double_exponential", 1e2, 1e-4)
    print(hist["Newton steps"])
except Exception as e:
    print("EXC", e)
