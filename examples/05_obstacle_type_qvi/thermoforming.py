"""Thermoforming quasi-variational inequality with the latent variable proximal point method on the HIP backend.
Counterpart of /root/reference/examples/05_obstacle_type_qvi/thermoforming_dolfinx.py (a script without flags; -M sets
the mesh size the reference hard-codes as 150)."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from proximalgalerkin_amd.thermoforming import solve_problem  # noqa: E402

if __name__ == "__main__":
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("-M", type=int, default=150, help="cells per side of the unit square")
    a = parser.parse_args()
    solve_problem(a.M)
