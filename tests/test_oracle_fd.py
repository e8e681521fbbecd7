"""The one reference-held, self-contained statement of the LVPP algorithm - obstacle_finite_difference.jl, transcribed
in oracle/fd_oracle.py - against the finite-element oracle (oracle/pg_oracle.py) run with the vertex quadrature rule,
for which the two discretisations coincide row for row (fd_oracle.py header).  SURVEY.md section 8(c) item 4."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from oracle import fd_oracle as FD
from oracle import pg_oracle as O


def fe_problem(N):
    coords, cells = O.create_rectangle(N - 1, N - 1)
    return O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N - 1, N - 1), quadrature="tri_vertex_3")


def test_phi_is_the_same_profile_as_obstacle_pg():
    x = np.random.default_rng(0).uniform(-1, 1, (2, 1000))
    np.testing.assert_allclose(FD.phi(x[0], x[1]), O.phi_set(x), rtol=0, atol=1e-15)


def test_alpha_rule_carries_the_capped_value():
    a, seq = 1.0, []
    for k in range(12):
        a = FD.alpha_rule(k, a)
        seq.append(a)
    # k = 0, 1: r^(q^k) - alpha < C -> C; then 1.5^(1.5^k) - previous alpha; capped at 1e2 from k = 7 on
    assert seq[0] == 1.0 and seq[1] == 1.0
    assert seq[2] == pytest.approx(1.5**2.25 - 1.0)
    assert seq[3] == pytest.approx(1.5 ** (1.5**3) - seq[2])
    assert all(s == 100.0 for s in seq[7:])
    assert FD.alpha_rule(100, 100.0) == 100.0  # Julia: Inf - 100 -> min(Inf, 1e2); Python's ** overflows instead


@pytest.mark.parametrize("N", [9, 17])
def test_fe_rows_with_the_vertex_rule_are_the_fd_rows_scaled(N):
    """K = (h^2/4) A_fd, M = D(0) = diag(m), b_phi = m * phi(x_i); FE residual = diag(s) x FD residual, FE Jacobian =
    diag(s) x FD Jacobian with s = h^2 on interior u rows, 1 on boundary u rows, m_i on psi rows."""
    P, Q = FD.FDProblem(N), fe_problem(N)
    h = 2.0 / (N - 1)
    interior = np.ones(P.n, bool)
    interior[P.bcs] = False
    assert np.array_equal(np.sort(Q.bc), P.bcs)
    Kfe = Q.K.toarray()
    np.testing.assert_allclose(Kfe[interior], (h * h / 4.0) * P.A.toarray()[interior], atol=1e-13)
    m = Q.m_l
    np.testing.assert_allclose(Q.M.toarray(), np.diag(m), atol=1e-15)
    np.testing.assert_allclose(m[interior], h * h, rtol=1e-13)
    np.testing.assert_allclose(Q.b_phi, m * P.phiv, rtol=1e-13)
    rng = np.random.default_rng(1)
    u, psi, w = rng.normal(size=P.n), rng.normal(size=P.n), rng.normal(size=P.n)
    u[P.bcs] = 0.0
    a_fd = 2.7
    s = np.concatenate([np.where(interior, h * h, 1.0), m])
    x, xk = np.concatenate([u, psi]), np.concatenate([np.zeros(P.n), w])
    np.testing.assert_allclose(Q.residual(x, xk, 4.0 * a_fd), s * P.residual(u, psi, a_fd, w), rtol=1e-12, atol=1e-13)
    Jfe, Jfd = Q.jacobian(x, 4.0 * a_fd).toarray(), P.jacobian(a_fd, psi).toarray()
    np.testing.assert_allclose(Jfe, s[:, None] * Jfd, rtol=1e-12, atol=1e-13)


def fe_newton_step(Q):
    n = Q.n

    def step(u, psi, alpha_fd, w):
        x, xk = np.concatenate([u, psi]), np.concatenate([np.zeros(n), w])
        a = 4.0 * alpha_fd
        dx = spla.splu(Q.jacobian(x, a).tocsc()).solve(-Q.residual(x, xk, a))
        return u + dx[:n], psi + dx[n:]

    return step


@pytest.mark.parametrize("N", [9, 17, 33])
def test_fe_oracle_reproduces_the_fd_run_iterate_for_iterate(N):
    rec_fd, rec_fe = [], []
    _, U, its, per = FD.fd_lvpp_solve(N, record=rec_fd)
    _, U2, its2, per2 = FD.fd_lvpp_solve(N, newton_step=fe_newton_step(fe_problem(N)), record=rec_fe)
    assert per == per2 and its == its2
    for (k, i, u, p), (k2, i2, u2, p2) in zip(rec_fd, rec_fe):
        assert (k, i) == (k2, i2)
        assert np.linalg.norm(u - u2) <= 1e-11 * max(np.linalg.norm(u), 1e-300)
    assert np.linalg.norm(U - U2) <= 1e-11 * np.linalg.norm(U)


def test_fd_run_properties():
    """What the Julia script's run must look like: it terminates by its own 1e-9 test well before 101 proximal steps, the
    solution is feasible to the size of the last latent update (u - phi = e^psi > 0 at convergence), zero on the boundary, and the
    Newton counts are mesh-independent (its = [...] over N = 2^j + 1 at :114-121 is the script's own output)."""
    counts = {}
    for j in (3, 4, 5, 6):
        N = 2**j + 1
        P = FD.FDProblem(N)
        _, U, its, per = FD.fd_lvpp_solve(N)
        counts[N] = its
        assert len(per) < 60
        u = U.ravel(order="F")
        assert np.all(u[P.bcs] == 0.0)
        assert (u - P.phiv).min() > -1e-6
        assert per[0] >= per[-1] and per[-1] <= 2
    assert max(counts.values()) <= 2 * min(counts.values()), counts
