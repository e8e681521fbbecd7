"""Signorini contact problem with the latent variable proximal point algorithm on the HIP backend.

Counterpart of /root/reference/examples/02_signorini/signorini_dolfinx.py with the reference's flags where they apply: --E --nu
--disp --gap --n-tol --n-max-iterations --quadrature-degree --max-iterations --tol --alpha_scheme --alpha_0 --alpha_c and the two mesh
branches (the reference's sub-command words `native` / `file` are accepted): `--nx --ny --nz` (native,
:361-386: a tetrahedral unit cube, BASELINE.json config 5) or `--filename mesh.msh|mesh.xdmf --contact-tag --displacement-tag`
(file, :406-409: tetrahedra + tagged boundary triangles, e.g. the half sphere of generate_mesh.py - an ORDER-2 mesh like the
reference's: `--degree 2` is isoparametric on its 10-node tetrahedra, `--degree 1` uses the vertices; XDMF must carry inline data).  `--degree {1,2}`, default 2 as in the reference (:68-73); the native mesh is hexahedral
as in the reference (`--cell-type hexahedron`, Q1 / Q2 elements); BASELINE.json config 5 is
`--cell-type tetrahedron --degree 1 --nx 70 --ny 70 --nz 70`.
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from proximalgalerkin_amd.signorini import create_unit_cube, create_unit_cube_hex, native_tags, solve_contact_problem  # noqa: E402

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
    parser.add_argument("--degree", type=int, default=2, choices=[1, 2], help="Degree of primal and latent space")
    parser.add_argument("--cell-type", dest="cell_type", default="hexahedron", choices=["hexahedron", "tetrahedron"],
                        help="native mesh: hexahedra as in the reference (:381-386), or the same vertex grid split into tetrahedra")
    # the reference selects the mesh with a sub-command, `native [--dim --nx --ny --nz]` or `file --filename ...` (:120-140): both words
    # are accepted (and implied by --filename); 2-D meshes are not built
    parser.add_argument("mesh_mode", nargs="?", choices=["native", "file"], default=None, help="mesh branch (optional)")
    parser.add_argument("--dim", type=int, default=3, choices=[3], help="Geometrical dimension of the native mesh (3 only)")
    parser.add_argument("--quadrature-degree", dest="quadrature_degree", type=int, default=4, help="Quadrature degree for integration")
    parser.add_argument("--n-max-iterations", dest="newton_max_iterations", type=int, default=250,
                        help="Maximum number of iterations of Newton iteration")
    parser.add_argument("--nx", type=int, default=16)
    parser.add_argument("--ny", type=int, default=7)
    parser.add_argument("--nz", type=int, default=5)
    parser.add_argument("--filename", type=Path, default=None, help="mesh file (the reference's `file` sub-command)")
    parser.add_argument("--contact-tag", dest="ct", type=int, default=2, help="Tag of contact surface")
    parser.add_argument("--displacement-tag", dest="dt", type=int, default=1, help="Tag of displacement surface")
    a = parser.parse_args()
    if a.mesh_mode == "file" and a.filename is None:
        parser.error("file: --filename is required")
    if a.filename is not None:
        from proximalgalerkin_amd.io import read_tet_mesh

        mesh, mt = read_tet_mesh(a.filename)
        bcs = {"contact": (a.ct,), "displacement": (a.dt,)}
    else:
        mesh = create_unit_cube_hex(a.nx, a.ny, a.nz) if a.cell_type == "hexahedron" else create_unit_cube(a.nx, a.ny, a.nz)
        mt, bcs = native_tags(mesh)
    it, iterations = solve_contact_problem(mesh=mesh, facet_tag=mt, boundary_conditions=bcs, degree=a.degree, E=a.E, nu=a.nu,
                                           gap=a.gap, disp=a.disp, newton_max_its=a.newton_max_iterations, newton_tol=a.newton_tol,
                                           max_iterations=a.max_iterations, alpha_scheme=a.alpha_scheme, alpha_0=a.alpha_0,
                                           alpha_c=a.alpha_c, tol=a.tol, output=a.output, quadrature_degree=a.quadrature_degree)
    print(it, iterations, sum(iterations), min(iterations), max(iterations))
    assert it == len(iterations)
