"""General-mesh path check (GPU): a Delaunay triangulation of the unit disk (the reference's ex01 domain,
generate_mesh_gmsh.py:23) with ~npts vertices; reports Newton / Krylov counts of the single-level path."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.spatial import Delaunay
from proximalgalerkin_amd import fem
from proximalgalerkin_amd.obstacle import setup_problem, run_outer_loop

def disk_mesh(h):
    pts = [(0.0, 0.0)]
    nr = int(round(1.0 / h))
    for k in range(1, nr + 1):
        r = k / nr
        m = max(6, int(round(2 * np.pi * r / h)))
        th = 2 * np.pi * (np.arange(m) + 0.5 * (k % 2)) / m
        pts += list(zip(r * np.cos(th), r * np.sin(th)))
    pts = np.array(pts)
    tri = Delaunay(pts)
    return fem.Mesh(pts, tri.simplices.astype(np.int32))

for h in [float(a) for a in sys.argv[1:]]:
    msh = disk_mesh(h)
    opts = {"snes_linesearch_type": "none", "snes_rtol": 1e-6, "snes_max_it": 100, "snes_error_if_not_converged": True,
            "ksp_max_it": 2000, "ksp_gmres_restart": 50}
    problem, sol, sol_k, alpha = setup_problem(msh, 1, petsc_options=opts)
    t = time.perf_counter()
    try:
        hist = run_outer_loop(problem, sol, sol_k, alpha, 500, "double_exponential", 1e2, 1e-4)
        print(f"h={h}: {msh.num_vertices} vertices, newton {hist['Newton steps']}, last-step Krylov its {problem.solver.getLinearSolveIterations()}, "
              f"{time.perf_counter()-t:.2f} s, u_max {sol.x.array[:msh.num_vertices].max():.4f}", flush=True)
    except Exception as e:
        print(f"h={h}: {msh.num_vertices} vertices FAILED: {e}", flush=True)
    problem.close()
