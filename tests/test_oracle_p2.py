"""Oracle, Lagrange degree 2 (obstacle_pg.py -p 2): pinned by mathematics and by the degree-1 class."""
import numpy as np

from oracle import pg_oracle as O


def test_general_class_at_degree_one_equals_p1_class():
    N = 12
    c, ce = O.create_rectangle(N, N)
    p1 = O.ObstacleP1(c, ce, O.boundary_vertices_rectangle(N, N))
    g1 = O.ObstacleLagrange(c, ce, 1)
    rng = np.random.default_rng(0)
    x, xk = rng.standard_normal(2 * p1.n) * 0.3, rng.standard_normal(2 * p1.n) * 0.3
    assert np.array_equal(np.sort(g1.bc), p1.bc)
    assert np.abs(g1.residual(x, xk, 1.3) - p1.residual(x, xk, 1.3)).max() < 1e-13
    assert abs(g1.jacobian(x, 1.3) - p1.jacobian(x, 1.3)).max() < 1e-13
    assert np.allclose(g1.observables(x, xk, 1.3), p1.observables(x, xk, 1.3), rtol=1e-13)


def test_p2_space_counts_and_patch_tests():
    N = 10
    c, ce = O.create_rectangle(N, N)
    p = O.ObstacleLagrange(c, ce, 2)
    nv, ne, nc = (N + 1) ** 2, 3 * N * N + 2 * N, 2 * N * N
    assert p.n == nv + ne == (2 * N + 1) ** 2
    assert p.nnz_s == nv + 7 * ne + 12 * nc
    assert len(p.bc) == 4 * 2 * N  # vertices + edge midpoints on the boundary
    # P2 reproduces quadratics: K q = int (-lap q) N  at interior dofs; M is exact for products of quadratics
    X, Y = p.dof_coords[:, 0], p.dof_coords[:, 1]
    q = 0.3 + 0.5 * X - 0.2 * Y + 0.7 * X * Y + 0.1 * Y**2 - 0.4 * X**2
    minus_lap = -(0.2 - 0.8)
    r = p.K @ q - minus_lap * p.m_l
    assert np.abs(r[~p.isbc]).max() < 1e-13
    assert abs(p.M.sum() - 4.0) < 1e-13
    assert abs(q @ (p.M @ q) - _int_q2()) < 1e-12
    assert abs(p.K - p.K.T).max() < 1e-13 and abs(p.M - p.M.T).max() < 1e-15


def _int_q2():
    # int_{[-1,1]^2} q^2 with q above, by tensor Gauss quadrature (exact)
    from numpy.polynomial.legendre import leggauss

    g, w = leggauss(6)
    X, Y = np.meshgrid(g, g, indexing="ij")
    q = 0.3 + 0.5 * X - 0.2 * Y + 0.7 * X * Y + 0.1 * Y**2 - 0.4 * X**2
    return float(np.sum(np.outer(w, w) * q * q))


def test_p2_jacobian_is_derivative_of_residual_and_run_converges():
    N = 8
    c, ce = O.create_rectangle(N, N)
    p = O.ObstacleLagrange(c, ce, 2)
    rng = np.random.default_rng(1)
    x = rng.standard_normal(2 * p.n) * 0.2
    x[p.bc] = 0.0
    xk = rng.standard_normal(2 * p.n) * 0.2
    d = rng.standard_normal(2 * p.n)
    d[p.bc] = 0.0
    eps = 1e-6
    fd = (p.residual(x + eps * d, xk, 1.7) - p.residual(x - eps * d, xk, 1.7)) / (2 * eps)
    assert np.linalg.norm(p.jacobian(x, 1.7) @ d - fd) < 1e-8 * np.linalg.norm(fd)
    xs, h = O.solve_problem(p, 500, "double_exponential", 1e2, 1e-4)
    assert h["Newton steps"][:3] == [5, 4, 3] and h["Primal increments"][-1] < 1e-4
    # P1 and P2 solutions of the same mesh agree to discretisation accuracy
    p1 = O.ObstacleLagrange(c, ce, 1)
    x1, _ = O.solve_problem(p1, 500, "double_exponential", 1e2, 1e-4)
    assert np.abs(xs[:p.nv] - x1[:p1.n]).max() < 0.05


def test_p2_golden_fixture():
    import pathlib

    g = np.load(pathlib.Path(__file__).resolve().parent / "golden" / "obstacle_p2_n16_settingsB.npz")
    c, ce = O.create_rectangle(16, 16)
    p = O.ObstacleLagrange(c, ce, 2)
    assert np.allclose(p.residual(g["x_iter"], g["xk_iter"], 2.5), g["F_iter"], rtol=1e-12, atol=1e-14)
    x, h = O.solve_problem(p, 100, "double_exponential", 1e2, 1e-4)
    assert h["Newton steps"] == g["hist_Newton_steps"].tolist()
    assert np.linalg.norm(x[:p.n] - g["x_final"][:p.n]) <= 1e-11 * np.linalg.norm(g["x_final"][:p.n])


def _curved_disk(h=0.3):
    """the polygonal disk mesh of fem.create_disk with the mid-side nodes of its boundary edges on the unit circle (what gmsh writes
    for Mesh.ElementOrder 2: generate_mesh_gmsh.py:30-33)"""
    from proximalgalerkin_amd import fem

    m = fem.create_disk(h)
    e, ce = O.build_edges(m.cells, m.num_vertices)
    mid = 0.5 * (m.geometry[e[:, 0]] + m.geometry[e[:, 1]])
    straight = mid.copy()
    b = np.flatnonzero(np.bincount(ce.ravel(), minlength=len(e)) == 1)
    mid[b] /= np.linalg.norm(mid[b], axis=1)[:, None]
    return m, straight, mid


def test_order2_geometry_in_the_oracle():
    """ObstacleLagrange(midside=...): the isoparametric cell map of degree 2 (round 5).  Straight mid-side nodes reproduce the affine
    element matrices to rounding; curved ones integrate over the DISK (area -> pi at O(h^4) instead of the polygon's O(h^2)); the
    Jacobian is still the derivative of the residual (finite differences), and constants are in the kernel of the stiffness matrix
    on the interior (the gradient of the mapped basis functions sums to zero at every quadrature point)."""
    m, straight, mid = _curved_disk(0.3)
    affine = O.ObstacleLagrange(m.geometry, m.cells, 2)
    same = O.ObstacleLagrange(m.geometry, m.cells, 2, midside=straight)
    assert abs(affine.K - same.K).max() < 1e-13 and abs(affine.M - same.M).max() < 1e-15 and abs(affine.b_phi - same.b_phi).max() < 1e-15
    errs = []
    for h in (0.3, 0.15):
        mh, _, midh = _curved_disk(h)
        p = O.ObstacleLagrange(mh.geometry, mh.cells, 2, midside=midh)
        errs.append((abs(p.wdet.sum() - np.pi), abs(O.ObstacleLagrange(mh.geometry, mh.cells, 2).wdet.sum() - np.pi)))
        assert abs(p.M.sum() - p.wdet.sum()) < 1e-12  # the basis is a partition of unity on the curved cells too
        assert np.abs(p.Gq.sum(axis=2)).max() < 1e-11  # ... and its gradients sum to zero
    assert errs[0][0] < 2e-3 * errs[0][1] and errs[1][0] < errs[0][0] / 10  # 16x per halving for the curved cells, 4x for the polygon
    p = O.ObstacleLagrange(m.geometry, m.cells, 2, midside=mid)
    rng = np.random.default_rng(2)
    x, xk = 0.1 * rng.standard_normal(2 * p.n), 0.1 * rng.standard_normal(2 * p.n)
    J = p.jacobian(x, 2.0)
    v = rng.standard_normal(2 * p.n)
    v[p.bc] = 0.0
    eps = 1e-6
    fd = (p.residual(x + eps * v, xk, 2.0) - p.residual(x - eps * v, xk, 2.0)) / (2 * eps)
    assert np.linalg.norm(J @ v - fd) < 1e-7 * np.linalg.norm(fd)
    x_ref, h_ref = O.solve_problem(p, 100, "double_exponential", 1e2, 1e-4)
    assert all(r > 0 for r in h_ref["Newton steps"]) and np.all(np.isfinite(x_ref))
