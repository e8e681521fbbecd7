"""Known answers pinning the example 06 oracle (oracle/gc_oracle.py) by mathematics - the reference holds no tests or
golden data for this path: quadrature exactness, J = dF/dx, symmetry, the BC contract, exactness
for data the discrete spaces contain, Newton's quadratic convergence, and the committed golden fixture."""
import pathlib

import numpy as np
import pytest

from oracle import gc_oracle as G
from oracle import pg_oracle as O

GOLDEN = pathlib.Path(__file__).resolve().parent / "golden"


@pytest.fixture(scope="module")
def prob():
    coords, cells = O.create_rectangle(6, 5, (0.0, 0.0), (1.0, 1.0))
    return G.GradientConstraintP2(coords, cells)


def test_degree10_rule_is_exact_to_degree_11():
    from math import factorial
    X, w = O.load_quadrature("tri_deg10_gj36")
    assert len(w) == 36 and np.all(w > 0) and np.all(X >= 0) and np.all(X.sum(axis=1) <= 1)
    for p in range(12):
        for q in range(12 - p):
            exact = factorial(p) * factorial(q) / factorial(p + q + 2)
            assert abs(np.sum(w * X[:, 0] ** p * X[:, 1] ** q) - exact) < 2e-16 + 1e-14 * exact


def test_jacobian_is_derivative_of_residual_and_symmetric(prob):
    rng = np.random.default_rng(0)
    x = rng.standard_normal(prob.ntot)
    xk = rng.standard_normal(prob.ntot)
    J = prob.jacobian(x, 3.0)
    assert abs(J - J.T).max() < 1e-14
    d = rng.standard_normal(prob.ntot)
    d[prob.bc] = 0.0
    eps = 1e-6
    fd = (prob.residual(x + eps * d, xk, 3.0) - prob.residual(x - eps * d, xk, 3.0)) / (2 * eps)
    assert np.abs(fd - J @ d).max() < 1e-8 * np.abs(fd).max()


def test_bc_contract(prob):
    """lvpp/problem.py:54-77: F[bc] = x[bc] - g, Jacobian rows/cols of bc dofs = identity"""
    rng = np.random.default_rng(1)
    x = rng.standard_normal(prob.ntot)
    F = prob.residual(x, np.zeros(prob.ntot), 2.0)
    assert np.array_equal(F[prob.bc], x[prob.bc])
    J = prob.jacobian(x, 2.0).tocsr()
    sub = J[prob.bc]
    assert sub.nnz == len(prob.bc) or abs(sub).sum() == len(prob.bc)
    assert np.allclose(J[prob.bc][:, prob.bc].diagonal(), 1.0)
    assert abs(J[:, prob.bc]).sum() == len(prob.bc)


def test_blocks_integrate_polynomials_exactly(prob):
    """K reproduces the Dirichlet energy of a quadratic, G the L2 pairing of its gradient with P1 fields."""
    X = prob.dof_coords
    u = 1.0 + 2 * X[:, 0] - X[:, 1] + 0.5 * X[:, 0] ** 2 + X[:, 0] * X[:, 1]  # in P2
    # int |grad u|^2 over the unit square, grad u = (2 + x + y, -1 + x)
    exact = (4 + 1 / 3 + 1 / 3 + 2 + 2 + 0.5) + (1 - 1 + 1 / 3)
    assert abs(u @ (prob.K @ u) - exact) < 1e-12
    V = prob.coords
    wx = 1.0 + V[:, 0]  # P1 test field (w, 0): int (2 + x + y)(1 + x) = 2 + 1 + .5 + .5 + 1/3 + .25
    assert abs(wx @ (prob.Gx @ u) - (2 + 1 + 0.5 + 0.5 + 1 / 3 + 0.25)) < 1e-12


def test_newton_converges_quadratically_and_counts_match_fixture():
    z = np.load(GOLDEN / "gradient_constraint_p2_n12_defaults.npz")
    N = int(z["N"])
    coords, cells = O.create_rectangle(N, N, (0.0, 0.0), (1.0, 1.0))
    prob = G.GradientConstraintP2(coords, cells)
    x, newton, diffs = G.solve_problem(prob)
    assert list(newton) == list(z["newton"])
    assert np.linalg.norm(x - z["x_final"]) <= 1e-9 * np.linalg.norm(z["x_final"])
    assert np.allclose(diffs, z["L2_diff"], rtol=1e-7, atol=1e-14)
    assert diffs[-1] < 1e-8 and np.all(np.diff(diffs[1:]) < 0)
    # quadratic convergence of the first proximal step
    log = O.NewtonLog()
    O.newton_solve(prob, np.zeros(prob.ntot), np.zeros(prob.ntot), 1.0, O.SnesOptions(rtol=1e-12, atol=1e-13, max_it=20), log=log)
    f = np.array(log.fnorms)
    assert f[-1] < 1e-10 * f[0]
    ks = [k for k in range(1, len(f) - 1) if f[k + 1] > 1e-12 * f[0]]  # above the rounding floor
    assert ks and f[ks[-1] + 1] < 10 * (f[ks[-1]] / f[0]) ** 1.7 * f[0], f
    # the constraint is active somewhere and respected everywhere at the solution: |phi psi / s| <= phi
    u, px, py = prob.split(x)
    s = np.sqrt(1 + px**2 + py**2)
    assert (np.hypot(px, py) / s).max() > 0.999
