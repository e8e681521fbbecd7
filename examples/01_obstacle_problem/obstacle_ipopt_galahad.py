"""The obstacle problem as a bound-constrained minimisation, solved in the Galahad and IPOPT slots on the GPU - counterpart of the
reference's obstacle_ipopt_galahad.py with its flags (:19-41): -P/--path mesh, --ipopt, --galahad, --max-iter, --tol, --hessian,
-o/--output.  `ObstacleProblem` and `setup_problem` (the reference defines them in this file, :46-127) live in
proximalgalerkin_amd.optimization and are re-exported here, so `from obstacle_ipopt_galahad import ObstacleProblem, setup_problem`
(compare_all.py:14) keeps working.

    python generate_mesh_gmsh.py && python obstacle_ipopt_galahad.py -P meshes/disk_1.xdmf --galahad --ipopt --hessian
"""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from proximalgalerkin_amd import fem, io  # noqa: E402
from proximalgalerkin_amd.optimization import ObstacleProblem, galahad_solver, ipopt_solver  # noqa: E402
from proximalgalerkin_amd.optimization import setup_problem as _setup_on_mesh  # noqa: E402

__all__ = ["ObstacleProblem", "setup_problem"]


def setup_problem(filename: Path):
    """(S, M, f, (lower, upper), coords, mesh) of the mesh file - the reference's `setup_problem(filename)` (:46-91) returns the
    first four as PETSc / DOLFINx objects; here they are scipy matrices and numpy arrays from the HIP assembly."""
    mesh = io.read_mesh(filename)
    S, M, f, bounds, coords = _setup_on_mesh(mesh)
    return S, M, f, bounds, coords, mesh


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Solve the obstacle problem on a general mesh using the Galahad or IPOPT slot",
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--path", "-P", dest="infile", type=Path, default=Path("./meshes/disk_3.xdmf"), help="Path to infile")
    parser.add_argument("--ipopt", action="store_true", default=False, help="Use the IPOPT slot")
    parser.add_argument("--galahad", action="store_true", default=False, help="Use the Galahad slot")
    parser.add_argument("--max-iter", type=int, default=200, help="Maximum number of iterations")
    parser.add_argument("--tol", type=float, default=1e-6, help="Convergence tolerance")
    parser.add_argument("--hessian", dest="use_hessian", action="store_true", default=False, help="Use exact hessian")
    parser.add_argument("--output", "-o", dest="outdir", type=Path, default=Path("results"), help="Output directory")
    args = parser.parse_args()
    S, M, f, bounds, coords, mesh = setup_problem(args.infile)
    problem = ObstacleProblem(S, M, f)
    args.outdir.mkdir(parents=True, exist_ok=True)
    n = mesh.num_vertices
    if args.galahad:
        x, iterations = galahad_solver(problem, np.zeros(len(f)), bounds, max_iter=args.max_iter, use_hessian=args.use_hessian,
                                       tol=args.tol, coords=coords)
        print(f"galahad slot: {iterations} iterations, objective {problem.objective(x):.12e}")
        io.write_vtu(args.outdir / "galahad_obstacle.vtu", mesh.geometry, mesh.cells, {"galahad": x[:n]})
    if args.ipopt:
        x = ipopt_solver(problem, np.zeros(len(f)), bounds, max_iter=args.max_iter, tol=args.tol, activate_hessian=args.use_hessian,
                         coords=coords if args.use_hessian else None)
        print(f"ipopt slot: {problem.total_iteration_count} iterations, objective {problem.objective(x):.12e}")
        io.write_vtu(args.outdir / "ipopt_obstacle.vtu", mesh.geometry, mesh.cells, {"ipopt": x[:n]})
