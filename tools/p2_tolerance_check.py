import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import pg_oracle as O
from proximalgalerkin_amd import fem
from proximalgalerkin_amd.obstacle import setup_problem, run_outer_loop
N=32
coords, cells = O.create_rectangle(N, N); prob = O.ObstacleLagrange(coords, cells, 2)
x_ref, h_ref = O.solve_problem(prob, 100, "double_exponential", 1e2, 1e-4); n=prob.n
msh = fem.create_rectangle(((-1.0,-1.0),(1.0,1.0)), (N, N))
xs={}
for rt in (1e-9,1e-10,1e-11,1e-12):
    opts={"snes_linesearch_type":"none","snes_rtol":1e-6,"snes_max_it":100,"snes_error_if_not_converged":True,"ksp_rtol":rt}
    problem, sol, sol_k, alpha = setup_problem(msh, 2, petsc_options=opts)
    h = run_outer_loop(problem, sol, sol_k, alpha, 100, "double_exponential", 1e2, 1e-4)
    x=sol.x.array.copy(); xs[rt]=x; problem.close()
    print(rt, h["Newton steps"]==h_ref["Newton steps"], "u vs LU", np.linalg.norm(x[:n]-x_ref[:n])/np.linalg.norm(x_ref[:n]), "u vs rt=1e-12" if rt!=1e-12 else "")
for rt in (1e-9,1e-10,1e-11): print(rt, "u vs 1e-12 run", np.linalg.norm(xs[rt][:n]-xs[1e-12][:n])/np.linalg.norm(xs[1e-12][:n]))
