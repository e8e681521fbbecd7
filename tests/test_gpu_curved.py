"""ORDER-2 GEOMETRY (round 5): the reference's own meshes are gmsh meshes of element order 2
(examples/01_obstacle_problem/generate_mesh_gmsh.py:30-33, src/lvpp/mesh_generation.py:88,158), and DOLFINx integrates `-p 1` (its
default) and `-p 2` on the curved cells.  The HIP path (pgx_create_curved: weights and inverse Jacobians of the quadratic cell map per
quadrature point - in every P2 kernel, csrc/pgx_p2.hip, and in the k_*_c twins of the P1 kernels, csrc/pgx_kernels.hip) against the
CPU oracle with the same map (oracle/pg_oracle.py ObstacleLagrange(midside=...)) on the
committed order-2 disk mesh: kernels to 1e-12, the full LVPP run with identical Newton counts and the primal field to 1e-10; the
affine fast path when every mid-side node is its edge's midpoint; and the closed-form solution on the disk, which the curved cells
approach with a far smaller error constant than the polygon."""
import pathlib

import numpy as np
import pytest

from oracle import pg_oracle as O

pytestmark = pytest.mark.gpu
GOLD = pathlib.Path(__file__).resolve().parent / "golden"


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _disk(h, curved=True):
    """fem.create_disk's polygonal mesh with (curved) the mid-side nodes of the boundary edges pushed onto the unit circle - what
    gmsh writes for `Mesh.ElementOrder 2` on a disk (tests/golden/disk_h0.2_order2.msh is exactly this at h = 0.2)."""
    from proximalgalerkin_amd import fem

    m = fem.create_disk(h)
    e, ce = m.edges()
    mid = 0.5 * (m.geometry[e[:, 0]] + m.geometry[e[:, 1]])
    if curved:
        b = np.flatnonzero(np.bincount(ce.ravel(), minlength=len(e)) == 1)
        mid[b] /= np.linalg.norm(mid[b], axis=1)[:, None]
    return fem.Mesh(m.geometry, m.cells, midside=mid)


@pytest.mark.parametrize("degree", [1, 2])
def test_curved_kernels_match_the_oracle(require_gpu, degree):
    from proximalgalerkin_amd import io
    from proximalgalerkin_amd.obstacle import setup_problem

    mesh = io.read_mesh(GOLD / "disk_h0.2_order2.msh")
    assert mesh.curved and abs(np.linalg.norm(mesh.midside, axis=1).max() - 1.0) < 1e-12
    problem, sol, sol_k, alpha = setup_problem(mesh, degree)
    prob = O.ObstacleLagrange(mesh.geometry, mesh.cells, degree, midside=mesh.midside)
    flat = O.ObstacleLagrange(mesh.geometry, mesh.cells, degree)
    assert sol.function_space.block_size == prob.n
    rng = np.random.default_rng(3)
    x, xk = rng.standard_normal(2 * prob.n) * 0.1, rng.standard_normal(2 * prob.n) * 0.1
    x[prob.n:] = -np.abs(rng.standard_normal(prob.n)) * np.where(rng.random(prob.n) < 0.3, 200.0, 2.0)
    alpha.value = 1.75
    sol_k.x.array[:] = xk
    F, fn = problem.residual(x)
    Fr = prob.residual(x, xk, 1.75)
    assert _rel(F, Fr) < 1e-12 and abs(fn - np.linalg.norm(Fr)) < 1e-12 * np.linalg.norm(Fr)
    assert _rel(F, flat.residual(x, xk, 1.75)) > 1e-4  # ... and it is NOT the polygon's residual
    problem.assemble_jacobian(x)
    rowptr, col, K, Mv, D = problem.export_blocks()
    assert np.array_equal(rowptr, prob.indptr_s.astype(np.int32)) and np.array_equal(col, prob.indices_s)
    assert _rel(K, prob.K.data) < 1e-12 and _rel(Mv, prob.M.data) < 1e-12 and _rel(D, prob.jacobian_blocks(x)) < 1e-12
    assert abs(Mv.sum() - np.pi) < 2e-5 and abs(flat.M.data.sum() - np.pi) > 1e-2  # the mass matrix sums to the AREA: disk vs polygon
    v = rng.standard_normal(2 * prob.n)
    assert _rel(problem.spmv(v), prob.jacobian(x, 1.75) @ v) < 1e-12
    xc = np.clip(x, -40, None)
    sol.x.array[:] = xc
    assert np.allclose(problem.observables(), prob.observables(xc, xk, 1.75), rtol=1e-12, atol=1e-14)
    problem.close()


@pytest.mark.parametrize("degree", [1, 2])
def test_curved_full_run_matches_the_oracle(require_gpu, degree):
    from proximalgalerkin_amd import io
    from proximalgalerkin_amd.obstacle import COLUMNS, solve_problem

    mesh = io.read_mesh(GOLD / "disk_h0.2_order2.msh")
    sol, newton, hist = solve_problem(mesh, degree, 100, "double_exponential", 1e2, 1e-4, verbose=False, return_history=True)
    prob = O.ObstacleLagrange(mesh.geometry, mesh.cells, degree, midside=mesh.midside)
    x_ref, h_ref = O.solve_problem(prob, 100, "double_exponential", 1e2, 1e-4)
    assert hist["Newton steps"] == h_ref["Newton steps"]
    assert _rel(sol.x.array[:prob.n], x_ref[:prob.n]) < 1e-10
    for c in COLUMNS:
        assert np.allclose(hist[c], h_ref[c], rtol=1e-7, atol=1e-11), c


def test_straight_midside_nodes_take_the_affine_path(require_gpu):
    """A mesh whose mid-side nodes ARE the edge midpoints is affine: `Mesh.curved` is False and the run is bit for bit the run on the
    plain mesh (degree 1 and 2); `flattened()` of a curved mesh is the polygon of rounds 1-4."""
    from proximalgalerkin_amd.obstacle import solve_problem

    straight, plain = _disk(0.3, curved=False), _disk(0.3, curved=False).flattened()
    assert not straight.curved
    for degree in (1, 2):
        a = solve_problem(straight, degree, 100, "double_exponential", 1e2, 1e-4, verbose=False, return_history=True)
        b = solve_problem(plain, degree, 100, "double_exponential", 1e2, 1e-4, verbose=False, return_history=True)
        assert a[2]["Newton steps"] == b[2]["Newton steps"] and np.array_equal(a[0].x.array, b[0].x.array)
    curved = _disk(0.3)
    sol, newton, hist = solve_problem(curved.flattened(), 1, 100, "double_exponential", 1e2, 1e-4, verbose=False, return_history=True)
    prob = O.ObstacleP1(curved.geometry, curved.cells, curved.exterior_vertices())
    x_ref, h_ref = O.solve_problem(prob, 100, "double_exponential", 1e2, 1e-4)
    assert hist["Newton steps"] == h_ref["Newton steps"] and _rel(sol.x.array[:prob.n], x_ref[:prob.n]) < 1e-10


def test_curved_cells_shrink_the_error_against_the_closed_form_on_the_disk(require_gpu):
    """on the unit disk the obstacle problem of phi_set has the closed-form solution u = phi for r <= a,
    -c ln r beyond (a^2 (1 - ln a) = r0^2).  Away from the free boundary (r >= 0.6, where u is smooth) the error of a P2 run on the
    POLYGON is the O(h^2) displacement of the boundary on which u = 0 is imposed; isoparametric cells remove that term."""
    from scipy.optimize import brentq

    from proximalgalerkin_amd.obstacle import solve_problem

    r0 = 0.5
    a = brentq(lambda a: a * a * (1 - np.log(a)) - r0 * r0, 0.1, 0.45)
    c = a * a / np.sqrt(r0 * r0 - a * a)
    err = {}
    for curved in (False, True):
        mesh = _disk(0.1, curved=curved)
        sol, newton, hist = solve_problem(mesh, 2, 100, "constant", 1e5, 1e-8, verbose=False, return_history=True)
        nv = mesh.num_vertices
        r = np.hypot(mesh.geometry[:, 0], mesh.geometry[:, 1])
        outer = r >= 0.6
        err[curved] = np.abs(sol.x.array[:nv][outer] + c * np.log(r[outer])).max()
    assert err[True] < 0.5 * err[False], err
