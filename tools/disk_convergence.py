"""Mesh-convergence of example 01 on the reference's own domain (unit disk) against the closed-form solution of SURVEY.md App. A.6:\nu = phi for r <= a, -c ln r beyond, a^2 (1 - ln a) = r0^2.  python tools/disk_convergence.py 0.1 0.05 0.025"""
import sys, time
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))
import numpy as np
from scipy.optimize import brentq
from proximalgalerkin_amd import fem
from proximalgalerkin_amd.obstacle import setup_problem, run_outer_loop
r0 = 0.5
a = brentq(lambda a: a*a*(1-np.log(a)) - r0*r0, 0.1, 0.45)
c = a*a/np.sqrt(r0*r0 - a*a)
def exact(x):
    r = np.hypot(x[:,0], x[:,1])
    return np.where(r <= a, np.sqrt(np.maximum(r0*r0 - r*r, 0)), -c*np.log(np.maximum(r, 1e-300)))
print("a", a, "c", c)
for h in [float(v) for v in sys.argv[1:]]:
    msh = fem.create_disk(h)
    problem, sol, sol_k, alpha = setup_problem(msh, 1, petsc_options={"snes_linesearch_type": "none", "snes_rtol": 1e-6, "snes_max_it": 100})
    t = time.perf_counter()
    hist = run_outer_loop(problem, sol, sol_k, alpha, 500, "double_exponential", 1e2, 1e-7)
    u = sol.x.array[:msh.num_vertices]
    err = np.abs(u - exact(msh.geometry)).max()
    print(f"h={h} nv={msh.num_vertices} newton={sum(hist['Newton steps'])} outer={hist['outer_iterations']} max err {err:.3e} time {time.perf_counter()-t:.2f}", flush=True)
    problem.close()
