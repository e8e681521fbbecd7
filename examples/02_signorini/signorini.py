"""Signorini contact problem with the latent variable proximal point algorithm on the HIP backend.

Counterpart of /root/reference/examples/02_signorini/signorini_dolfinx.py (`native` mesh branch, :361-386) with the
reference's flags where they apply: --E --nu --disp --gap --n-tol --max-iterations --tol --alpha_scheme --alpha_0
--alpha_c --nx --ny --nz.  Degree 1 on a tetrahedral unit cube (BASELINE.json config 5).
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from proximalgalerkin_amd.signorini import create_unit_cube, native_tags, solve_contact_problem  # noqa: E402

if __name__ == "__main__":
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--output", "-o", type=Path, default=Path("output"))
    parser.add_argument("--E", type=float, default=2.0e4, help="Young's modulus")
    parser.add_argument("--nu", type=float, default=0.3, help="Poisson's ratio")
    parser.add_argument("--disp", type=float, default=-0.25, help="Displacement in the z direction")
    parser.add_argument("--gap", type=float, default=-0.00, help="z coordinate of rigid surface")
    parser.add_argument("--n-tol", dest="newton_tol", type=float, default=1e-6, help="Tolerance for Newton iteration")
    parser.add_argument("--max-iterations", dest="max_iterations", type=int, default=25)
    parser.add_argument("--tol", type=float, default=1e-6, help="Tolerance for the LVPP algorithm")
    parser.add_argument("--alpha_scheme", type=str, default="doubling", choices=["constant", "linear", "doubling"])
    parser.add_argument("--alpha_0", type=float, default=1.0)
    parser.add_argument("--alpha_c", type=float, default=1.0)
    parser.add_argument("--nx", type=int, default=16)
    parser.add_argument("--ny", type=int, default=7)
    parser.add_argument("--nz", type=int, default=5)
    a = parser.parse_args()
    mesh = create_unit_cube(a.nx, a.ny, a.nz)
    mt, bcs = native_tags(mesh)
    it, iterations = solve_contact_problem(mesh=mesh, facet_tag=mt, boundary_conditions=bcs, degree=1, E=a.E, nu=a.nu,
                                           gap=a.gap, disp=a.disp, newton_max_its=250, newton_tol=a.newton_tol,
                                           max_iterations=a.max_iterations, alpha_scheme=a.alpha_scheme, alpha_0=a.alpha_0,
                                           alpha_c=a.alpha_c, tol=a.tol, output=a.output)
    print(it, iterations, sum(iterations), min(iterations), max(iterations))
    assert it == len(iterations)
