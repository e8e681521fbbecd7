"""Known answers pinning the example 02 oracle (oracle/sg_oracle.py) by mathematics (the reference holds no tests or
golden data for this path): mesh volume / facet counts, rigid-body modes and symmetry of the elasticity block, a
homogeneous-strain patch test, J = dF/dx, the BC contract and the LVPP run's physical end state."""
import numpy as np
import pytest

from oracle import sg_oracle as S


@pytest.fixture(scope="module")
def prob():
    coords, cells = S.create_unit_cube_tets(4, 3, 3)
    cf = S.boundary_facets_where(coords, cells, lambda c: np.isclose(c[:, 2], 0.0))
    top = np.flatnonzero(np.isclose(coords[:, 2], 1.0))
    return S.SignoriniP1(coords, cells, cf, top)


def test_mesh(prob):
    x = prob.coords[prob.cells]
    vol = np.abs(np.linalg.det(np.stack([x[:, 1] - x[:, 0], x[:, 2] - x[:, 0], x[:, 3] - x[:, 0]], axis=2))) / 6
    assert abs(vol.sum() - 1.0) < 1e-13 and vol.min() > 0
    assert prob.nc == 6 * 4 * 3 * 3 and prob.nf == 2 * 4 * 3 and prob.npsi == 5 * 4
    assert abs(prob.farea2.sum() / 2 - 1.0) < 1e-13  # the contact facets tile the unit square z = 0


def test_elasticity_block(prob):
    nv, X = prob.nv, prob.coords
    assert abs(prob.A - prob.A.T).max() < 1e-9
    for t in np.eye(3):  # translations
        assert np.abs(prob.A @ np.repeat(t, nv)).max() < 1e-9
    rot = np.concatenate([-X[:, 1], X[:, 0], np.zeros(nv)])  # infinitesimal rotation about z
    assert np.abs(prob.A @ rot).max() < 1e-9
    # homogeneous strain u = (a x, b y, c z): energy = vol * (lambda tr^2 + 2 mu (a^2 + b^2 + c^2))
    a, b, c = 0.3, -0.2, 0.5
    u = np.concatenate([a * X[:, 0], b * X[:, 1], c * X[:, 2]])
    exact = prob.lmbda * (a + b + c) ** 2 + 2 * prob.mu * (a * a + b * b + c * c)
    assert abs(u @ (prob.A @ u) - exact) < 1e-9 * exact
    # interior rows of A u vanish (constant stress is divergence free)
    interior = np.flatnonzero(np.all((X > 1e-12) & (X < 1 - 1e-12), axis=1))
    r = prob.A @ u
    assert np.abs(np.concatenate([r[interior], r[nv + interior], r[2 * nv + interior]])).max() < 1e-9


def test_jacobian_is_derivative_and_bc_contract(prob):
    rng = np.random.default_rng(0)
    x = rng.standard_normal(prob.ntot) * 0.01
    xk = rng.standard_normal(prob.ntot) * 0.01
    J = prob.jacobian(x, 2.0)
    d = rng.standard_normal(prob.ntot)
    d[prob.bc] = 0
    eps = 1e-6
    fd = (prob.residual(x + eps * d, xk, 2.0) - prob.residual(x - eps * d, xk, 2.0)) / (2 * eps)
    assert np.abs(fd - J @ d).max() < 1e-9 * np.abs(fd).max()
    F = prob.residual(x, xk, 2.0)
    assert np.allclose(F[prob.bc], x[prob.bc] - prob.bc_vals)
    Jc = J.tocsr()
    assert abs(Jc[prob.bc]).sum() == len(prob.bc) and abs(Jc[:, prob.bc]).sum() == len(prob.bc)


def test_lvpp_run_ends_in_contact_without_penetration():
    coords, cells = S.create_unit_cube_tets(5, 5, 5)
    cf = S.boundary_facets_where(coords, cells, lambda c: np.isclose(c[:, 2], 0.0))
    top = np.flatnonzero(np.isclose(coords[:, 2], 1.0))
    prob = S.SignoriniP1(coords, cells, cf, top)
    x, it, its = S.solve_contact_problem(prob)
    assert it == len(its) <= 10 and its[0] >= 1
    uz = x[2 * prob.nv:3 * prob.nv]
    # rigid plane at z = gap = 0 below a body pushed down by 0.25: the bottom face rests on the plane
    assert np.abs(uz[prob.cverts]).max() < 1e-6
    assert np.allclose(uz[top], -0.25)
