"""Known answers pinning the example 05 oracle (oracle/qvi_oracle.py): the MODIFIED Jacobian equals dF/dx minus the
documented eps/alpha stiffness term, the BC contract, g's branches, and a complete LVPP run (line search included)."""
import numpy as np

from oracle import pg_oracle as O
from oracle import qvi_oracle as Q


def _prob(M):
    coords, cells = O.create_rectangle(M, M, (0.0, 0.0), (1.0, 1.0))
    return Q.Thermoforming(coords, cells, O.boundary_vertices_rectangle(M, M))


def test_modified_jacobian_and_bc_contract():
    prob = _prob(7)
    rng = np.random.default_rng(0)
    x = rng.standard_normal(prob.ntot)
    x[2 * prob.nv:] = 3 + 2 * rng.standard_normal(prob.nv)  # exp(-psi) on both sides of the knee 0.01
    xk = rng.standard_normal(prob.ntot)
    alpha = 0.5
    J = prob.jacobian(x, alpha)
    d = rng.standard_normal(prob.ntot)
    d[prob.bc] = 0
    e = 1e-6
    fd = (prob.residual(x + e * d, xk, alpha) - prob.residual(x - e * d, xk, alpha)) / (2 * e)
    mod = np.zeros(prob.ntot)
    mod[2 * prob.nv:] = -(Q.EPS_MOD / alpha) * (prob.K @ d[2 * prob.nv:])  # thermoforming_dolfinx.py:69-71
    assert np.abs(fd - (J @ d - mod)).max() < 1e-8 * np.abs(fd).max()
    F = prob.residual(x, xk, alpha)
    assert np.array_equal(F[prob.bc], x[prob.bc])
    Jc = J.tocsr()
    assert abs(Jc[prob.bc]).sum() == len(prob.bc) and abs(Jc[:, prob.bc]).sum() == len(prob.bc)


def test_full_lvpp_run_with_line_search():
    prob = _prob(12)
    x, its, diffs = Q.solve_problem(prob)
    assert diffs[-1] < 1e-9 and len(its) < 100 and max(its) > 1
    u, T, psi = prob.split(x)
    # the membrane stays below the mould Phi0 + xi T (up to discretisation) and T in (0, 1]
    X = prob.coords
    mould = 1.0 - 2.0 * np.maximum(np.abs(X[:, 0] - 0.5), np.abs(X[:, 1] - 0.5)) + np.sin(np.pi * X[:, 0]) * np.sin(np.pi * X[:, 1]) * T
    assert (u - mould).max() < 5e-2 and T.min() > -1e-6 and T.max() < 1.0 + 1e-6
