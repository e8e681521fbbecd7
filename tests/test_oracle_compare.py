"""CPU twins of the comparison solvers (oracle/compare_oracle.py): both must satisfy the KKT conditions of the discrete obstacle
problem min 1/2 u^T S u s.t. u >= phi_h, u = 0 on the boundary (obstacle_ipopt_galahad.py:44-127, obstacle_snes.py:64-87) and agree."""
import numpy as np
import pytest

from oracle import compare_oracle as C
from oracle import pg_oracle as O


def _problem(N):
    coords, cells = O.create_rectangle(N, N)
    p = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    lower = O.phi_set(coords.T.copy()).copy()
    upper = np.full(p.n, np.inf)
    lower[p.bc] = 0.0
    upper[p.bc] = 0.0
    return p, lower, upper


@pytest.mark.parametrize("N", [12, 32])
def test_kkt_and_agreement(N):
    p, lo, up = _problem(N)
    b = np.zeros(p.n)
    u, it, sets = C.primal_dual_active_set(p.K, b, lo, up)
    free = np.ones(p.n, bool)
    free[p.bc] = False
    lam = p.K @ u - b
    assert (u - lo)[free].min() >= -1e-13                       # feasible
    assert lam[free].min() >= -1e-10                           # multiplier sign
    assert np.abs(lam[free] * (u - lo)[free]).max() <= 1e-10   # complementarity
    assert 2 <= it <= 12 and sets[-1] == sets[-2] or len(sets) == it
    x, it2 = C.projected_newton(p.K, b, lo, up, np.zeros(p.n), tol=1e-12)
    assert np.abs(x - u).max() <= 1e-9 and it2 <= 25
    assert abs(u.max() - 0.5) < 2.0 / N  # contact at the tip of the obstacle


def test_one_dimensional_known_answer():
    """-u'' = -1 on (0,1), u >= -0.05, u(0)=u(1)=0: contact interval around the midpoint, u'' = 0 there."""
    import scipy.sparse as sp

    n = 101
    h = 1.0 / (n - 1)
    S = sp.diags([-np.ones(n - 1), 2.0 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]).tocsr() / h
    b = -h * np.ones(n)
    lo, up = np.full(n, -0.05), np.full(n, np.inf)
    lo[[0, -1]] = up[[0, -1]] = 0.0
    u, it, _ = C.primal_dual_active_set(S, b, lo, up)
    assert np.isclose(u.min(), -0.05) and (u == -0.05).sum() > 10
    a = np.sqrt(2 * 0.05)  # free boundary: u = x^2/2 - a x on [0, a], tangent to the obstacle at x = a
    xs = np.linspace(0, 1, n)
    exact = np.where(xs <= a, xs**2 / 2 - a * xs, np.where(xs >= 1 - a, (1 - xs) ** 2 / 2 - a * (1 - xs), -0.05))
    assert np.abs(u - exact).max() < 2e-3
