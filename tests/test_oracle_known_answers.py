"""Pin the CPU oracle by mathematics (the reference has no tests or golden data for this path:
SURVEY.md section 4 and 8c - "parity unpinned").  Every check here is independent of FEniCSx."""
import math

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import krylov_proto as KP
from oracle import pg_oracle as O


@pytest.fixture(scope="module")
def prob16():
    coords, cells = O.create_rectangle(16, 16)
    return O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(16, 16))


def test_quadrature_table_is_exact_to_degree_6():
    pts, w = O.load_quadrature("tri_deg6_12")
    assert len(w) == 12 and abs(w.sum() - 0.5) < 1e-15 and np.all(w > 0)
    assert np.all(pts >= 0) and np.all(pts.sum(axis=1) <= 1)
    for p in range(7):
        for q in range(7 - p):
            exact = math.factorial(p) * math.factorial(q) / math.factorial(p + q + 2)
            assert abs(np.sum(w * pts[:, 0] ** p * pts[:, 1] ** q) - exact) < 2e-16 + 1e-14 * exact
    # and NOT exact at degree 7 (it is a degree-6 rule, not something stronger by accident)
    assert abs(np.sum(w * pts[:, 0] ** 7) - 1.0 / 72.0) > 1e-9


def test_mesh_is_the_right_diagonal_triangulation():
    coords, cells = O.create_rectangle(3, 2)
    assert coords.shape == (12, 2) and cells.shape == (12, 3)
    # vertex v = j*(nx+1)+i ; first square -> [v0,v1,v3],[v0,v2,v3]
    assert cells[0].tolist() == [0, 1, 5] and cells[1].tolist() == [0, 4, 5]
    assert np.allclose(coords[5], [-1 + 2 / 3, 0.0])
    # positive orientation not required (abs det), but areas must tile the domain
    x = coords[cells]
    e1, e2 = x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]
    det = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
    assert abs(0.5 * np.abs(det).sum() - 4.0) < 1e-14


def test_patch_tests_stiffness_and_mass(prob16):
    p = prob16
    interior = ~p.isbc
    lin = 0.3 + 1.7 * p.coords[:, 0] - 0.9 * p.coords[:, 1]
    assert np.abs((p.K @ lin)[interior]).max() < 1e-13  # K annihilates linears at interior vertices
    assert np.abs(p.K @ np.ones(p.n)).max() < 1e-13
    assert abs(p.M.sum() - 4.0) < 1e-13  # 1^T M 1 = |Omega|
    # M row sums = (area of the vertex patch)/3 = lumped mass
    assert np.allclose(np.asarray(p.M.sum(axis=1)).ravel(), p.m_l, rtol=1e-13)
    # right-diagonal uniform mesh: P1 stiffness is the 5-point Laplacian (diagonal edges decouple)
    i = 8 * 17 + 8
    row = p.K[i].toarray().ravel()
    assert abs(row[i] - 4.0) < 1e-13 and abs(row[i + 1] + 1.0) < 1e-13 and abs(row[i + 17] + 1.0) < 1e-13
    assert abs(row[i + 18]) < 1e-13
    assert (p.K - p.K.T).nnz == 0 or abs(p.K - p.K.T).max() < 1e-14


def test_scalar_pattern_counts(prob16):
    # nnz per scalar block = Nv + 2 Ne
    N = 16
    nv, ne = (N + 1) ** 2, 3 * N * N + 2 * N
    assert prob16.nnz_s == nv + 2 * ne
    J = prob16.jacobian(np.zeros(2 * prob16.n), 1.0)
    assert J.nnz == 4 * prob16.nnz_s  # explicit zeros kept in BC rows/cols, like PETSc


def test_bphi_against_higher_order_quadrature(prob16):
    # phi is only C^1 across r=b and non-polynomial: compare the degree-6 rule with a 400-point tensor
    # Gauss rule collapsed to the triangle; agreement ~1e-6 relative is the quadrature error itself
    from numpy.polynomial.legendre import leggauss

    g, gw = leggauss(20)
    s, t = (g + 1) / 2, (g + 1) / 2
    S, T = np.meshgrid(s, t, indexing="ij")
    X, Y = S, T * (1 - S)
    W = np.outer(gw, gw) / 4 * (1 - S)
    p = prob16
    x = p.coords[p.cells]
    tot_hi = 0.0
    for c in range(p.nc):
        xq = x[c, 0][None, None] * (1 - X - Y)[..., None] + x[c, 1] * X[..., None] + x[c, 2] * Y[..., None]
        tot_hi += p.detJ[c] * np.sum(W * O.phi_set(xq.reshape(-1, 2).T).reshape(X.shape))
    assert abs(p.b_phi.sum() - tot_hi) < 2e-4 * abs(tot_hi)


def test_obstacle_function_constants():
    # B ~ 1.1470787, C ~ -2.0647416, zero at r = 5/9
    r = np.array([[0.0, 0.45 - 1e-12, 0.45 + 1e-12, 5.0 / 9.0], [0.0, 0.0, 0.0, 0.0]])
    v = O.phi_set(r)
    assert abs(v[0] - 0.5) < 1e-15
    assert abs(v[1] - v[2]) < 1e-10  # continuous at r=b
    assert abs(v[3]) < 1e-12


def test_jacobian_is_the_derivative_of_the_residual(prob16):
    p = prob16
    rng = np.random.default_rng(0)
    x = rng.standard_normal(2 * p.n) * 0.2
    x[p.bc] = 0.0
    xk = rng.standard_normal(2 * p.n) * 0.2
    alpha = 1.7
    J = p.jacobian(x, alpha)
    d = rng.standard_normal(2 * p.n)
    d[p.bc] = 0.0
    eps = 1e-6
    fd = (p.residual(x + eps * d, xk, alpha) - p.residual(x - eps * d, xk, alpha)) / (2 * eps)
    assert np.linalg.norm(J @ d - fd) < 1e-8 * np.linalg.norm(fd)
    assert abs(J - J.T).max() < 1e-14  # symmetric saddle point [[aK, M],[M, -D]]


def test_boundary_condition_contract(prob16):
    """lvpp/problem.py:54-77: F[bc] = x[bc]-g; lifting == raw residual with BC values imposed;
    Jacobian BC rows/cols zero with unit diagonal."""
    p = prob16
    rng = np.random.default_rng(1)
    x = rng.standard_normal(2 * p.n)
    xk = rng.standard_normal(2 * p.n)
    F = p.residual(x, xk, 2.0)
    assert np.array_equal(F[p.bc], x[p.bc])
    x2 = x.copy()
    x2[p.bc] = 0.0
    F2 = p.residual(x2, xk, 2.0)
    free = np.setdiff1d(np.arange(2 * p.n), p.bc)
    assert np.allclose(F[free], F2[free], rtol=0, atol=1e-13)
    J = p.jacobian(x, 2.0).tocsr()
    for b in p.bc[:5]:
        r = J[b].toarray().ravel()
        assert r[b] == 1.0 and np.count_nonzero(r) == 1
        c = J[:, b].toarray().ravel()
        assert c[b] == 1.0 and np.count_nonzero(c) == 1


def test_exp_underflow_is_harmless(prob16):
    p = prob16
    x = np.zeros(2 * p.n)
    x[p.n:] = -5000.0  # deeper than anything in SURVEY.md H4
    F = p.residual(x, x, 1.0)
    assert np.all(np.isfinite(F))
    assert np.all(p.jacobian_blocks(x) == 0.0)


def test_newton_converges_quadratically(prob16):
    p = prob16
    log = O.NewtonLog()
    z = np.zeros(2 * p.n)
    x, reason, its = O.newton_solve(p, z, z, 1.0, O.SnesOptions(rtol=1e-13, max_it=20), log=log)
    assert reason in (O.SNES_CONVERGED_FNORM_RELATIVE, O.SNES_CONVERGED_SNORM_RELATIVE)
    f = np.array(log.fnorms)
    # away from the round-off floor the contraction is (at least nearly) quadratic: f[k+1] <= C f[k]^1.7
    ks = [k for k in range(len(f) - 1) if 1e-7 < f[k] < 1e-2 and f[k + 1] > 1e-13]
    assert ks, f
    for k in ks:
        assert f[k + 1] < 1e3 * f[k] ** 1.7, (k, f)


def test_snes_reason_codes(prob16):
    p = prob16
    z = np.zeros(2 * p.n)
    _, reason, its = O.newton_solve(p, z, z, 1.0, O.SnesOptions(rtol=1e-30, stol=0.0, max_it=2))
    assert (reason, its) == (O.SNES_DIVERGED_MAX_IT, 2)
    x, reason, its = O.newton_solve(p, z, z, 1.0, O.SnesOptions(rtol=1e-6, max_it=100))
    assert reason == O.SNES_CONVERGED_FNORM_RELATIVE and its == 5
    _, reason, its = O.newton_solve(p, x, z, 1.0, O.SnesOptions(rtol=1e-6, atol=1.0, max_it=100))
    assert (reason, its) == (O.SNES_CONVERGED_FNORM_ABS, 0)


def test_alpha_schedules():
    s = O.AlphaSchedule("double_exponential", 1e2)
    vals = [s.update(k) for k in range(9)]
    # 1, 1, 1.49, 2.44, 5.35, 16.4, 85.0, 100, 100
    assert np.allclose(vals, [1.0, 1.0, 1.4900343193257237, 2.439200639104808, 5.349445965312164,
                              16.387223352344883, 84.95478289516922, 100.0, 100.0], rtol=1e-14)
    # overflow branch of obstacle_pg.py:179-182 keeps the capped value
    for k in range(9, 40):
        assert s.update(k) == 100.0
    g = O.AlphaSchedule("geometric", 1e5)
    assert [g.update(k) for k in range(3)] == [1.0, 1.5, 2.25]
    c = O.AlphaSchedule("constant", 1e5)
    assert [c.update(k) for k in range(3)] == [1.0, 1.0, 1.0]


def test_full_run_iteration_counts_match_survey_probe():
    """SURVEY.md App. D (surveyor's independent prototype): N=64 settings B -> 9 outer / 21 Newton with
    steps 5,4,3,3,2,1,1,1,1.  A different quadrature rule was used there, so only counts are compared."""
    coords, cells = O.create_rectangle(64, 64)
    p = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(64, 64))
    x, h = O.solve_problem(p, 500, "double_exponential", 1e2, 1e-4)
    assert h["Newton steps"] == [5, 4, 3, 3, 2, 1, 1, 1, 1]
    u = x[:p.n]
    phi_v = O.phi_set(p.coords.T.copy())
    assert (u - phi_v).min() > -5e-3  # discrete feasibility up to O(h^2)
    assert h["Feasibility"][-1] < 1e-12  # u >= 0 at quadrature points is not required; u<0 never happens here
    assert h["Energy"][-1] == pytest.approx(h["Energy"][-2], rel=1e-3)


def test_oracle_krylov_agrees_with_oracle_lu():
    """Cross-implementation pin: FGMRES + collective-smoother multigrid vs SuperLU."""
    N = 32
    coords, cells = O.create_rectangle(N, N)
    p = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    x0, h0 = O.solve_problem(p, 500, "double_exponential", 1e2, 1e-4)
    stats = []
    x1, h1 = O.solve_problem(p, 500, "double_exponential", 1e2, 1e-4, linear_solve=KP.make_linear_solve(p, N, stats=stats))
    assert h0["Newton steps"] == h1["Newton steps"]
    assert max(stats) <= 25  # mesh-independent Krylov counts (10-17 observed for N=32..2048)
    n = p.n
    assert np.linalg.norm(x1[:n] - x0[:n]) <= 1e-10 * np.linalg.norm(x0[:n])


def test_galerkin_coarse_operators_stay_seven_point():
    N = 8
    coords, cells = O.create_rectangle(N, N)
    p = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    P = KP.interp_matrix(N)
    Kc = (P.T @ p.K @ P).tocsr()
    cc, cells_c = O.create_rectangle(N // 2, N // 2)
    pc = O.ObstacleP1(cc, cells_c, O.boundary_vertices_rectangle(N // 2, N // 2))
    # nested P1 spaces + exact integration: Galerkin coarse K and M ARE the coarse-mesh K and M
    assert abs(Kc - pc.K).max() < 1e-13
    assert abs((P.T @ p.M @ P) - pc.M).max() < 1e-13


def test_both_admissible_degree6_rules_are_exact_and_are_the_only_ones_the_search_found():
    """SURVEY.md H3 / VERDICT r03 item 8a: the moment system of a fully symmetric 12-point degree-6 rule with orbit structure
    [3, 3, 6], interior points and positive weights has exactly TWO admissible roots in a global search of 30 000 starts
    (tools/quadrature_uniqueness.py -> profiles/r04_quadrature_uniqueness.json): Dunavant's - the default table - and a second
    one, shipped as "tri_deg6_12_b".  Both integrate every monomial of degree <= 6 exactly."""
    import json
    import math
    import pathlib

    root = pathlib.Path(__file__).resolve().parents[1]
    rep = json.loads((root / "profiles" / "r04_quadrature_uniqueness.json").read_text())
    assert rep["n_starts"] >= 30000 and rep["distinct_admissible_rules"] == 2
    found = [np.array(r) for r in rep["rules"]]
    for name in ("tri_deg6_12", "tri_deg6_12_b"):
        pts, w = O.load_quadrature(name)
        assert len(w) == 12 and w.min() > 0 and pts.min() > 0 and (pts.sum(axis=1) < 1).all()
        for p in range(7):
            for q in range(7 - p):
                exact = math.factorial(p) * math.factorial(q) / math.factorial(p + q + 2)
                assert abs(np.sum(w * pts[:, 0] ** p * pts[:, 1] ** q) - exact) < 1e-15
        # canonical form of tools/quadrature_uniqueness.py: the two 3-orbits (a, w) sorted by a, the sorted 6-orbit point, its weight
        bary = np.sort(np.column_stack([pts, 1 - pts.sum(axis=1)]), axis=1)
        groups = {}
        for b, wi in zip(bary, w):
            groups.setdefault(round(float(wi), 13), []).append(b)
        o3 = sorted((float(g[0][0]) if abs(g[0][0] - g[0][1]) < 1e-12 else float(g[0][2]), wi) for wi, g in groups.items() if len(g) == 3)
        # a 3-orbit point sorted is (a, a, 1-2a) or (1-2a, a, a): its repeated coordinate is a
        (w6, g6), = [(wi, g) for wi, g in groups.items() if len(g) == 6]
        canon = np.array([o3[0][0], o3[0][1], o3[1][0], o3[1][1], *g6[0], w6])
        assert min(np.max(np.abs(canon - f)) for f in found) < 1e-9, name
