"""Known answers pinning the example 08 oracle (oracle/ic_oracle.py): the Jacobian is the derivative of the residual, the BC
contract, the two latent maps reduce to examples 01 / 06 when the other constraint is switched off, and a complete continuation
(line search, rejected solves and alpha adaptivity included) ends feasible for BOTH constraints."""
import numpy as np

from oracle import ic_oracle as IO


def test_jacobian_is_the_derivative_and_bc_contract():
    rng = np.random.default_rng(0)
    x = np.concatenate([[0.0], np.sort(rng.uniform(0, 1, 20)), [1.0]])
    prob = IO.Intersecting(x=x)
    prob.set_phic(0.5)
    z, zk = rng.standard_normal(prob.ntot), rng.standard_normal(prob.ntot)
    z[2 * prob.nv:] *= 3
    d = rng.standard_normal(prob.ntot)
    d[prob.bc] = 0
    e = 1e-6
    fd = (prob.residual(z + e * d, zk, 0.7) - prob.residual(z - e * d, zk, 0.7)) / (2 * e)
    J = prob.jacobian(z, 0.7)
    assert np.abs(fd - J @ d).max() < 1e-8 * np.abs(fd).max()
    assert np.array_equal(prob.residual(z, zk, 0.7)[prob.bc], z[prob.bc])
    Jc = J.tocsr()
    assert abs(Jc[prob.bc]).sum() == len(prob.bc) and abs(Jc[:, prob.bc]).sum() == len(prob.bc)


def test_quadrature_is_basix_default_for_degree_6():
    t, w = IO.gauss_legendre_unit(6)
    assert len(t) == 4 and abs(w.sum() - 1) < 1e-15
    for k in range(8):  # exact to degree 7
        assert abs(w @ t**k - 1.0 / (k + 1)) < 1e-15


def test_latent_rows_are_those_of_examples_01_and_06():
    """R_psi0 vanishes where u = exp(psi0) + phi0 holds pointwise in the quadrature sense; R_psi where u' = phi psi / sqrt(1+psi^2)."""
    prob = IO.Intersecting(64)
    prob.set_phic(2.0)
    n = prob.nv
    s = 0.3  # constant slope, constant psi = s / sqrt(phi^2 - s^2) between 0.2 and 0.8 (phi = 100 there)
    u = s * prob.x
    psi = np.where((prob.x > 0.25) & (prob.x < 0.75), s / np.sqrt(100.0**2 - s * s), 0.0)
    z = np.concatenate([u, np.zeros(n), psi])
    R = prob.residual(z, np.zeros(prob.ntot), 1.0)
    inner = (prob.x > 0.3) & (prob.x < 0.7)
    assert np.abs(R[2 * n:][inner]).max() < 1e-14


def test_full_continuation_is_feasible_for_both_constraints():
    prob = IO.Intersecting(400)
    z, n_lvpp, n_newton, log = IO.solve_problem(prob)
    assert all(k > 0 for k in n_lvpp) and log[-1][5] is not None and log[-1][5] < 1e-4
    assert any(r[5] is None for r in log)  # at least one solve was rejected and retried with alpha / 2
    u = prob.split(z)[0]
    assert (u - IO.phi0_bump(prob.x)).min() > -1e-3
    slope = np.abs(np.diff(u) / prob.h)
    outer = (prob.x[1:] <= 0.2) | (prob.x[:-1] > 0.8)
    assert slope[outer].max() <= 0.01 * (1 + 1e-2) and slope.max() < 100.0
    assert abs(u.max() - 1.0) < 1e-2  # the membrane touches the bump's top, phi0(0.5) = 1


def test_oracle_reproduces_the_committed_golden_of_the_references_configuration():
    """tests/golden/intersecting_n1001.npz (tools/make_golden_families.py ic 1001): 1001 cells, phic = 3 ... 0.01."""
    import pathlib

    g = np.load(pathlib.Path(__file__).parent / "golden" / "intersecting_n1001.npz")
    prob = IO.Intersecting(int(g["N"]))
    z, n_lvpp, n_newton, log = IO.solve_problem(prob)
    assert list(n_lvpp) == list(g["lvpp"]) and list(n_newton) == list(g["newton"])
    assert np.array_equal(np.asarray([[r[0], r[1], r[2], r[3], r[4]] for r in log], dtype=np.float64), g["attempts"])
    assert np.linalg.norm(z - g["z_final"]) <= 1e-12 * np.linalg.norm(g["z_final"])
