"""Example 08 (intersecting constraints: obstacle AND gradient bound, two latent variables, l2 line search) HIP path vs the CPU
oracle (oracle/ic_oracle.py) through the C ABI of include/pgx_ic.h.  Tolerances: kernels 1e-12 relative; one Newton solve: the same
line-search decisions (iteration count, reason) and the iterate to 1e-10; the full continuation in phic (:112-175): identical
proximal and Newton counts per phic - every rejected solve, alpha halving and doubling included - and u to 1e-8 relative L2
(the reference's SNES tolerance here is 1e-6)."""
import numpy as np
import pytest

from oracle import ic_oracle as IO

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _setup(x, phic=0.5):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd import intersecting as I

    problem = I.IntersectingProblem(fem.IntervalMesh(x), I.phi0_bump, I.phi_bound(phic))
    prob = IO.Intersecting(x=x)
    prob.set_phic(phic)
    assert problem.ndofs == prob.ntot
    return problem, prob


def _meshes():
    rng = np.random.default_rng(3)
    graded = np.concatenate([[0.0], np.sort(rng.uniform(0.0, 1.0, 60)), [1.0]])
    return {"two cells": np.array([0.0, 0.4, 1.0]), "uniform 37": np.linspace(0.0, 1.0, 38), "random 61": graded}


@pytest.mark.parametrize("name", list(_meshes()))
def test_kernels_match_oracle(require_gpu, name):
    x = _meshes()[name]
    problem, prob = _setup(x)
    rng = np.random.default_rng(5)
    z = rng.standard_normal(prob.ntot)
    z[2 * prob.nv:] *= 4.0  # the Hellinger map on both sides of its knee
    zk = rng.standard_normal(prob.ntot)
    for alpha, phic in ((2.0**-4, 0.5), (8.0, 0.01)):
        problem.set_alpha(alpha)
        from proximalgalerkin_amd import intersecting as I

        problem.set_phi(I.phi_bound(phic))
        prob.set_phic(phic)
        problem.set_prev(zk)
        F, fn = problem.residual(z)
        Fr = prob.residual(z, zk, alpha)
        assert _rel(F, Fr) < 1e-12 and abs(fn - np.linalg.norm(Fr)) <= 1e-12 * np.linalg.norm(Fr)
        assert np.array_equal(F[prob.bc], z[prob.bc])
        J = problem.jacobian(z)
        Jr = prob.jacobian(z, alpha).tocsr()
        assert abs(J - Jr).max() <= 1e-12 * abs(Jr).max()
        n = prob.nv
        for blk in ((slice(0, n), slice(2 * n, 3 * n)), (slice(n, 2 * n), slice(n, 2 * n)), (slice(2 * n, 3 * n), slice(0, n)),
                    (slice(2 * n, 3 * n), slice(2 * n, 3 * n))):
            assert abs(J[blk] - Jr[blk]).max() <= 1e-12 * abs(Jr[blk]).max()
        v = rng.standard_normal(prob.ntot)
        assert _rel(problem.spmv(v), Jr @ v) < 1e-12
    problem.set_state(z)
    problem.set_prev(zk)
    assert abs(problem.l2_increment() - prob.l2_increment(z, zk)) <= 1e-12 * prob.l2_increment(z, zk)
    problem.close()


def test_one_newton_solve_takes_the_oracles_line_search_path(require_gpu):
    """from z = 0 at alpha = 1, phic = 3 the l2 search rejects the full step more than once; alpha = 1/2 converges (the first two
    rows of the oracle's log at n = 1001)"""
    x = np.linspace(0.0, 1.0, 202)
    problem, prob = _setup(x, phic=3)
    for alpha in (1.0, 0.5, 0.125):
        z0 = np.zeros(prob.ntot)
        problem.set_state(z0)
        problem.set_prev(z0)
        problem.set_alpha(alpha)
        reason, its = problem.solve()
        zr, reason_r, its_r = prob.newton_l2(z0, z0, alpha)
        assert (reason, its) == (reason_r, its_r)
        if reason > 0:
            assert _rel(problem.get_state(), zr) < 1e-10
        else:
            assert np.array_equal(problem.get_state(), z0)  # a failed solve leaves the state alone
    problem.close()


@pytest.mark.parametrize("n", [200, 1001])
def test_full_continuation_matches_oracle(require_gpu, n):
    from proximalgalerkin_amd.intersecting import solve_problem

    n_lvpp, n_newton, z, log = solve_problem(n, verbose=False)
    prob = IO.Intersecting(n)
    z_ref, n_lvpp_ref, n_newton_ref, log_ref = IO.solve_problem(prob)
    assert list(n_lvpp) == list(n_lvpp_ref) and list(n_newton) == list(n_newton_ref)
    assert [r[:5] for r in log] == [r[:5] for r in log_ref]  # (phic, k, alpha, Newton steps, reason) of every attempt
    if n >= 400:
        assert any(r[5] is None for r in log)  # the run exercises the failure branch (alpha halved, iterate restored)
    assert _rel(z[: prob.nv], z_ref[: prob.nv]) < 1e-8
    u = z[: prob.nv]
    assert (u - IO.phi0_bump(prob.x)).min() > -1e-3  # feasible up to the proximal tolerance
    slope = np.abs(np.diff(u) / prob.h)
    outer = (prob.x[1:] <= 0.2) | (prob.x[:-1] > 0.8)
    assert slope[outer].max() <= 0.01 * (1 + 1e-2)  # the last gradient bound phic = 0.01 holds where it applies


def test_problem_stated_as_forms_runs_the_same_solve(require_gpu):
    """SURVEY section 8(f)3's acceptance test: the reference script's own statement (intersecting_constraints_dolfinx.py:13-63 as
    UFL forms, live Constants and Functions, a NonlinearProblem per attempt as at :124-126) through the front end selects the same
    HIP path as the direct host mirror."""
    from proximalgalerkin_amd import intersecting as I

    la, na, za, _ = I.solve_problem(300, verbose=False)
    lb, nb, zb = I.solve_problem_forms(300)
    assert list(la) == list(lb) and list(na) == list(nb)
    assert np.linalg.norm(za - zb) <= 1e-10 * np.linalg.norm(za)


def test_references_configuration_matches_the_committed_golden(require_gpu):
    """1001 cells, phic = 3, 2, 1, 0.5, 0.1, 0.01 (intersecting_constraints_dolfinx.py:13,114) against tests/golden/intersecting_n1001.npz
    (the oracle's run, committed): every attempt's (phic, k, alpha, Newton steps, reason), the counts per phic, u to 1e-8."""
    import pathlib

    from proximalgalerkin_amd.intersecting import solve_problem

    g = np.load(pathlib.Path(__file__).parent / "golden" / "intersecting_n1001.npz")
    n = int(g["N"])
    n_lvpp, n_newton, z, log = solve_problem(n, verbose=False)
    assert list(n_lvpp) == list(g["lvpp"]) and list(n_newton) == list(g["newton"])
    assert np.array_equal(np.asarray([[r[0], r[1], r[2], r[3], r[4]] for r in log], dtype=np.float64), g["attempts"])
    assert _rel(z[: n + 1], g["z_final"][: n + 1]) < 1e-8
