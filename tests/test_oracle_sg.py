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


def test_order2_tetrahedra_in_the_oracle():
    """SignoriniP2(midside=...): isoparametric P2 on 10-node tetrahedra (round 5).  Straight mid-edge nodes reproduce the affine
    matrices to rounding (a degree-5 rule integrates the affine cells' quadratic integrand exactly); on the order-2 half sphere the
    Jacobian is the derivative of the residual, the rigid translations are in the kernel of the elasticity block, and the facet
    coupling sums to the area of the CURVED contact surface."""
    import sys

    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
    from proximalgalerkin_amd import mesh_generation

    coords, cells = S.create_unit_cube_tets(2, 2, 1)
    cf = S.boundary_facets_where(coords, cells, lambda x: np.isclose(x[:, 2], 0.0))
    bf = S.boundary_facets_where(coords, cells, lambda x: np.isclose(x[:, 2], 1.0))
    a = S.SignoriniP2(coords, cells, cf, bf)
    b = S.SignoriniP2(coords, cells, cf, bf, midside=0.5 * (coords[a.edges[:, 0]] + coords[a.edges[:, 1]]))
    assert abs(a.A - b.A).max() < 1e-13 * abs(a.A).max() and abs(a.MG - b.MG).max() < 1e-15 and abs(a.b_g - b.b_g).max() < 1e-15
    pts, w = S.tet_gauss_jacobi(5)
    assert len(w) == 27 and abs(w.sum() - 1 / 6) < 1e-16 and pts.min() > 0 and pts.sum(axis=1).max() < 1

    mesh, _, ft = mesh_generation.create_half_sphere(res=0.2)
    p = S.SignoriniP2(mesh.geometry, mesh.cells, ft.find(2), ft.find(1), disp=-0.1, midside=mesh.midside)
    flat = S.SignoriniP2(mesh.geometry, mesh.cells, ft.find(2), ft.find(1), disp=-0.1)
    assert np.array_equal(p.edges, mesh.edges())
    area = 2 * np.pi * 0.4**2
    assert abs(p.MG.sum() - area) < 1.2e-2 < 3e-2 < abs(flat.MG.sum() - area)  # (res 0.2: a third of the surface edges were pulled back)
    nn = p.nv
    for i in range(3):  # rigid translations: sigma(const) = 0
        t = np.zeros(3 * nn)
        t[i * nn:(i + 1) * nn] = 1.0
        assert abs(p.A @ t).max() < 1e-9 * abs(p.A).max()
    rng = np.random.default_rng(5)
    x, xk = 0.01 * rng.standard_normal(p.ntot), 0.01 * rng.standard_normal(p.ntot)
    v = rng.standard_normal(p.ntot)
    v[p.bc] = 0.0
    eps = 1e-6
    fd = (p.residual(x + eps * v, xk, 2.0) - p.residual(x - eps * v, xk, 2.0)) / (2 * eps)
    assert np.linalg.norm(p.jacobian(x, 2.0) @ v - fd) < 1e-7 * np.linalg.norm(fd)
