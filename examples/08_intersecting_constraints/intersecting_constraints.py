"""Intersecting constraints (an obstacle and a gradient bound on one membrane) with the latent variable proximal point method on
the HIP backend.  Counterpart of /root/reference/examples/08_intersecting_constraints/intersecting_constraints_dolfinx.py (a script
without flags; -N sets the mesh size the reference hard-codes as 1001, --forms states the problem as UFL forms like the script)."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from proximalgalerkin_amd.intersecting import solve_problem, solve_problem_forms  # noqa: E402

if __name__ == "__main__":
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("-N", type=int, default=1001, help="cells of the unit interval")
    parser.add_argument("--forms", action="store_true", help="state the problem as forms (through the UFL-subset front end)")
    a = parser.parse_args()
    if a.forms:
        num_lvpp_iterations, num_newton_iterations, _ = solve_problem_forms(a.N, verbose=True)
    else:
        num_lvpp_iterations, num_newton_iterations, _, _ = solve_problem(a.N)
    print(f"{num_lvpp_iterations=}")  # :189-190
    print(f"{num_newton_iterations=}")
