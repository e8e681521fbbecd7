"""Obstacle problem (experiment 4 of Keith & Surowiec, Proximal Galerkin, FoCM 2024) on the HIP backend.

Counterpart of /root/reference/examples/01_obstacle_problem/obstacle_pg.py: same CLI flags, same proximal
loop, same CSV columns and return value.  The XDMF mesh argument (-f) is replaced by -N cells per side of a
right-diagonal triangulation of [-1,1]^2 (the reference's own square-domain variant,
obstacle_finite_difference.jl:46); mesh file I/O is outside the hot path.
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))

from proximalgalerkin_amd import fem  # noqa: E402
from proximalgalerkin_amd.obstacle import solve_problem  # noqa: E402

if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Run examples from paper",
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("-N", dest="N", type=int, default=64, help="cells per side of the square mesh")
    parser.add_argument("--file-path", "--path", "-f", dest="filename", type=Path, default=None,
                        help="gmsh MSH (2.2 / 4.1 ASCII) file with a triangle mesh - the reference's -f takes the XDMF file its "
                             "gmsh script writes (obstacle_pg.py:276-280); HDF5 is not available offline")
    parser.add_argument("--disk", dest="disk_h", type=float, default=0.0,
                        help="mesh size of a unit-DISK mesh (the reference's own domain, generate_mesh_gmsh.py:23) instead of "
                             "the square; general mesh -> sparse-LU preconditioner")
    parser.add_argument("--polynomial_order", "-p", dest="polynomial_order", type=int, default=1, choices=[1, 2],
                        help="Polynomial order of primal space")
    parser.add_argument("--alpha-scheme", dest="alpha_scheme", type=str, default="constant",
                        choices=["constant", "double_exponential", "geometric"], help="Step size rule")
    parser.add_argument("--max-iter", "-i", dest="maximum_number_of_outer_loop_iterations", type=int, default=100,
                        help="Maximum number of outer loop iterations")
    parser.add_argument("--alpha-max", "-a", dest="alpha_max", type=float, default=1e5, help="Maximum alpha")
    parser.add_argument("--tol", "-t", dest="tol_exit", type=float, default=1e-6,
                        help="Tolerance for exiting Newton iteration")
    args = parser.parse_args()
    if args.filename is not None:
        from proximalgalerkin_amd.io import read_mesh

        msh = read_mesh(args.filename)  # .xdmf / gmsh .msh; order-2 triangles keep their mid-side nodes (curved cells)
    else:
        msh = (fem.create_disk(args.disk_h) if args.disk_h > 0.0 else
               fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (args.N, args.N)))
    sol, newton_steps = solve_problem(msh, args.polynomial_order, args.maximum_number_of_outer_loop_iterations,
                                      args.alpha_scheme, args.alpha_max, args.tol_exit,
                                      output_dir=Path.cwd() / "output")
    print(f"total Newton steps: {newton_steps}")
    # fields for ParaView (the reference writes u and psi with VTXWriter, obstacle_pg.py:239-243)
    from proximalgalerkin_amd.io import write_vtu

    V = sol.function_space
    n = V.block_size
    write_vtu(Path.cwd() / "output" / "obstacle.vtu", V.dof_coordinates(), V.cell_dofs(),
              {"u": sol.x.array[:n], "psi": sol.x.array[n:]})
