#!/usr/bin/env python3
"""VI Newton solve of the obstacle problem - the script of /root/reference/examples/01_obstacle_problem/obstacle_snes.py on the GPU:
P1 space, F = (grad u, grad v) - (f, v), u = 0 on the boundary, lower bound = interpolated obstacle (setVariableBounds, :83-87),
`snes_type vinewtonssls`.  The semismooth Newton method here is the primal-dual active-set method with the GPU sparse LU as
linear solver (proximalgalerkin_amd/optimization.py::vi_newton_solver).

    python obstacle_snes.py -N 128 | --disk 0.05 | -f mesh.msh
"""
import argparse
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from proximalgalerkin_amd import fem, io  # noqa: E402
from proximalgalerkin_amd.optimization import setup_problem, vi_newton_solver  # noqa: E402


def snes_solve(mesh, snes_options=None):
    """(u, num_iterations) like the reference's snes_solve(filename, snes_options) (:36-102)."""
    o = dict(snes_options or {})
    if o.get("snes_type", "vinewtonssls") not in ("vinewtonssls", "vinewtonrsls"):
        raise NotImplementedError(f"snes_type {o['snes_type']}: a VI Newton type is required (variable bounds are set)")
    S, M, f, (lower, upper), coords = setup_problem(mesh)
    return vi_newton_solver(S, M @ f, lower, upper, max_it=int(o.get("snes_max_it", 1000)), coords=coords,
                            monitor="snes_monitor" in o)


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-N", type=int, default=64)
    ap.add_argument("--disk", type=float, default=0.0)
    ap.add_argument("-f", "--infile", "-P", "--path", dest="infile", type=pathlib.Path, default=None,
                    help="mesh file (-P / --path: the reference's flag, obstacle_snes.py:26-33)")
    a = ap.parse_args()
    mesh = io.read_mesh(a.infile) if a.infile else fem.create_disk(a.disk) if a.disk > 0 else fem.create_rectangle(
        ((-1.0, -1.0), (1.0, 1.0)), (a.N, a.N))
    u, its = snes_solve(mesh, snes_options={"snes_type": "vinewtonssls", "snes_monitor": None, "ksp_type": "preonly", "pc_type": "lu",
                                            "snes_max_it": 1000, "snes_atol": 1e-8, "snes_rtol": 1e-8, "snes_stol": 1e-8})  # :103-115
    print(f"VI Newton iterations: {its}   max u = {u.max():.6f}")
