#!/usr/bin/env python3
"""Iteration-count comparison of example 01 - the table of
/root/reference/examples/01_obstacle_problem/compare_all.py:170-182 - on one mesh, every solver on the GPU:

    bound-constrained trust-region slot (reference: Galahad TRB)   proximalgalerkin_amd.optimization.galahad_solver (projected Newton)
    proximal Galerkin P1 / P2 (reference: obstacle_pg.solve_problem, `double_exponential`, alpha_max 1e2, tol)   the HIP LVPP path
    first-order bound-constrained method (reference: IPOPT without Hessian)   projected gradient (use_hessian=False)
    VI Newton slot (reference: PETSc vinewtonssls)                 primal-dual active set / semismooth Newton

    python compare_all.py -N 64            (structured [-1,1]^2 mesh)      python compare_all.py --disk 0.05 | -f mesh.msh
"""
import argparse
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from proximalgalerkin_amd import fem, io  # noqa: E402
from proximalgalerkin_amd.obstacle import solve_problem  # noqa: E402
from proximalgalerkin_amd.optimization import ObstacleProblem, galahad_solver, setup_problem, vi_newton_solver  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-N", type=int, default=64)
    ap.add_argument("--disk", type=float, default=0.0, help="mesh size of a unit-disk Delaunay mesh (the reference's domain)")
    ap.add_argument("-f", "--infile", "-P", "--path", dest="infile", type=pathlib.Path, default=None,
                    help="gmsh .msh or inline-data .xdmf file (-P / --path: the reference's flag, compare_all.py:24-31)")
    ap.add_argument("-O", "--results", dest="result_dir", type=pathlib.Path, default=None,
                    help="directory for the solutions of the five solvers as VTU files (the reference writes them with VTXWriter, :32-39)")
    ap.add_argument("--max_iter", type=int, default=500)  # compare_all.py:32
    ap.add_argument("--tol", type=float, default=1e-4)  # compare_all.py:31
    ap.add_argument("--first-order-max-iter", type=int, default=20000)
    a = ap.parse_args()
    if a.infile:
        mesh = io.read_mesh(a.infile)
    elif a.disk > 0:
        mesh = fem.create_disk(a.disk)
    else:
        mesh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (a.N, a.N))
    S, M, f, bounds, coords = setup_problem(mesh)
    problem = ObstacleProblem(S, M, f)
    x_g, it_g = galahad_solver(problem, np.zeros(len(f)), bounds, log_level=0, use_hessian=True, max_iter=a.max_iter, tol=a.tol,
                               coords=coords)
    u1, it_p1 = solve_problem(mesh, 1, a.max_iter, "double_exponential", 1e2, a.tol, verbose=False)
    u2, it_p2 = solve_problem(mesh, 2, a.max_iter, "double_exponential", 1e2, a.tol, verbose=False)
    x_f, it_f = galahad_solver(problem, np.zeros(len(f)), bounds, log_level=0, use_hessian=False, max_iter=a.first_order_max_iter,
                               tol=a.tol, coords=coords)
    u_vi, it_vi = vi_newton_solver(S, M @ f, bounds[0], bounds[1], coords=coords)
    n = mesh.num_vertices
    print(f"vertices {n}, smallest |u_PG(P1) - u_VI|_inf = {np.abs(u1.x.array[:n] - u_vi).max():.2e}, |u_TR - u_VI|_inf = {np.abs(x_g - u_vi).max():.2e}")
    if a.result_dir is not None:
        a.result_dir.mkdir(parents=True, exist_ok=True)
        fields = {"galahad": x_g, "llvp_first_order": u1.x.array[:n], "ipopt_slot_first_order_method": x_f, "snes": u_vi}
        for nm, v in fields.items():
            io.write_vtu(a.result_dir / f"{nm}.vtu", mesh.geometry, mesh.cells, {nm: np.asarray(v)[:n]})
        io.write_vtu(a.result_dir / "llvp_second_order_at_vertices.vtu", mesh.geometry, mesh.cells, {"llvp_second_order": u2.x.array[:n]})
    name = a.infile or (f"disk h={a.disk}" if a.disk > 0 else f"square N={a.N}")
    print(f"{name} trust-region (projected Newton, GPU LU) iterations: {it_g}")
    print(f"{name} llvp iterations: (P=1) {it_p1}")
    print(f"{name} llvp iterations: (P=2) {it_p2}")
    print(f"{name} first-order (projected gradient) iterations: {it_f}")
    print(f"{name} VI semismooth Newton (primal-dual active set) iterations: {it_vi}")


if __name__ == "__main__":
    main()
