"""Who is off when HIP and the LU oracle disagree at the 1e-10 level? Runs the HIP path at several ksp_rtol and
compares (a) HIP runs among themselves, (b) each with the SuperLU oracle, (c) the oracle with iterative refinement."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scipy.sparse.linalg as spla
from scipy.spatial import Delaunay
from oracle import pg_oracle as O
from proximalgalerkin_amd import fem
from proximalgalerkin_amd.obstacle import phi_set, run_outer_loop
from proximalgalerkin_amd.problem import NonlinearProblem, ObstacleResidual, derivative

def make(case):
    if case == "disk":
        pts = [(0.0, 0.0)]; nr = 12
        for k in range(1, nr + 1):
            m = max(6, int(round(2 * np.pi * k))); th = 2 * np.pi * (np.arange(m) + 0.5 * (k % 2)) / m
            pts += list(zip(k / nr * np.cos(th), k / nr * np.sin(th)))
        pts = np.array(pts); msh = fem.Mesh(pts, Delaunay(pts).simplices.astype(np.int32))
        return msh, 1, 0.0, O.ObstacleP1(msh.geometry, msh.cells, msh.exterior_vertices())
    msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (12, 12)); c, ce = O.create_rectangle(12, 12)
    return msh, 2, -1.0, O.ObstacleLagrange(c, ce, 2, f=-1.0)

def refined_solve(J, b):  # LU + 3 steps of iterative refinement in fp64
    lu = spla.splu(J.tocsc()); x = lu.solve(b)
    for _ in range(3): x = x + lu.solve(b - J @ x)
    return x

for case in sys.argv[1:]:
    msh, degree, f, prob = make(case); n = prob.n
    x_lu, h = O.solve_problem(prob, 100, "double_exponential", 1e2, 1e-4)
    x_ir, h2 = O.solve_problem(prob, 100, "double_exponential", 1e2, 1e-4, linear_solve=refined_solve)
    print(case, "LU vs LU+refinement: u", np.linalg.norm(x_lu[:n] - x_ir[:n]) / np.linalg.norm(x_ir[:n]), h["Newton steps"] == h2["Newton steps"])
    for rt in (1e-9, 1e-10, 1e-12):
        V = fem.functionspace(msh, ("Lagrange", degree)); sol, sol_k = fem.Function(V), fem.Function(V)
        alpha, fc = fem.Constant(msh, 1.0), fem.Constant(msh, f); phi = fem.QuadratureFunction(msh, 6); phi.interpolate(phi_set)
        bc = fem.dirichletbc(0.0, msh.exterior_dofs(degree), V.sub(0)); F = ObstacleResidual(sol, sol_k, alpha, fc, phi, 6)
        p = NonlinearProblem(F, sol, bcs=[bc], J=derivative(F, sol), petsc_options={"snes_rtol": 1e-6, "snes_max_it": 100, "snes_linesearch_type": "none", "ksp_rtol": rt, "ksp_max_it": 2000})
        hh = run_outer_loop(p, sol, sol_k, alpha, 100, "double_exponential", 1e2, 1e-4); x = sol.x.array.copy(); p.close()
        print(f"  ksp_rtol={rt:g}: vs LU {np.linalg.norm(x[:n]-x_lu[:n])/np.linalg.norm(x_lu[:n]):.2e}   vs LU+refinement {np.linalg.norm(x[:n]-x_ir[:n])/np.linalg.norm(x_ir[:n]):.2e}  counts equal {hh['Newton steps']==h['Newton steps']}")
