"""Gradient-constraint problem |grad u| <= phi with the proximal Galerkin method on the HIP backend.

Counterpart of /root/reference/examples/06_gradient_constraints/gradient_constraint_dolfinx.py with the same CLI flags
(:208-320) where they apply: -N -M --primal_degree {2..8} --alpha_scheme --alpha_0 --alpha_c --max_iterations -s --warm_start
--result_dir --cell_type {triangle,quadrilateral}.
"""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from proximalgalerkin_amd.gradient_constraint import solve_problem  # noqa: E402

if __name__ == "__main__":
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("-N", type=int, default=200, help="Number of elements in x-direction")
    parser.add_argument("-M", type=int, default=200, help="Number of elements in y-direction")
    parser.add_argument("--cell_type", "-c", type=str, default="triangle", choices=["triangle", "quadrilateral"], help="Cell type")
    parser.add_argument("--primal_space", type=str, default="Lagrange", choices=["Lagrange", "P", "CG"])
    parser.add_argument("--primal_degree", type=int, default=2, choices=[2, 3, 4, 5, 6, 7, 8], help="Polynomial degree for primal variable")
    parser.add_argument("--alpha_scheme", type=str, default="doubling", choices=["constant", "linear", "doubling"])
    parser.add_argument("--alpha_0", type=float, default=1.0, help="Initial value of alpha")
    parser.add_argument("--alpha_c", type=float, default=1.0, help="Increment of alpha in linear scheme")
    parser.add_argument("--max_iterations", type=int, default=25, help="Maximum number of iterations")
    parser.add_argument("-s", "--stopping_tol", type=float, default=1e-8,
                        help="Stopping tolerance between two successive PG iterations (L2-difference)")
    parser.add_argument("--warm_start", action="store_true", help="Use warm start (solve Poisson problem to get initial guess)")
    parser.add_argument("--result_dir", type=Path, default=Path("results"), help="Directory to store results")
    a = parser.parse_args()
    iteration_counts, L2_diffs = solve_problem(N=a.N, M=a.M, primal_space=a.primal_space, primal_degree=a.primal_degree, cell_type=a.cell_type, alpha_scheme=a.alpha_scheme, alpha_0=a.alpha_0,
                                               alpha_c=a.alpha_c, max_iterations=a.max_iterations,
                                               stopping_tol=a.stopping_tol, result_dir=a.result_dir,
                                               warm_start=a.warm_start)
    print(f"Number of LVPP iterations {len(iteration_counts)}")
    print(f"Minimum number of solves {np.min(iteration_counts)}")
    print(f"Maximum number of solves {np.max(iteration_counts)}")
    print(f"Total number of Newton iterations: {np.sum(iteration_counts)}")
    print(iteration_counts)
    print(L2_diffs)
