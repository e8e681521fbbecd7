"""Example 05 (thermoforming QVI: three P1 fields, modified Jacobian, bt line search) HIP path vs the CPU oracle
(oracle/qvi_oracle.py) through the C ABI of include/pgx_qvi.h.  Tolerances: kernels 1e-12 relative; full LVPP run: identical
Newton counts per proximal step (every line-search decision included), membrane u <= 1e-9 relative L2 (the reference's
SNES tolerance here is only 1e-5, so late steps amplify rounding differences more than in examples 01/02/06)."""
import numpy as np
import pytest

from oracle import pg_oracle as O
from oracle import qvi_oracle as Q

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _setup(M):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.thermoforming import ThermoformingProblem

    problem = ThermoformingProblem(fem.create_unit_square(M, M))
    coords, cells = O.create_rectangle(M, M, (0.0, 0.0), (1.0, 1.0))
    prob = Q.Thermoforming(coords, cells, O.boundary_vertices_rectangle(M, M))
    assert problem.ndofs == prob.ntot
    return problem, prob


@pytest.mark.parametrize("M", [3, 10, 33])
def test_kernels_match_oracle(require_gpu, M):
    problem, prob = _setup(M)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(prob.ntot)
    x[2 * prob.nv:] = 3 + 3 * rng.standard_normal(prob.nv)  # exp(-psi) on both sides of the knee
    xk = rng.standard_normal(prob.ntot)
    for alpha in (2.0**-6, 16.0):
        problem.set_alpha(alpha)
        problem.set_prev(xk)
        F, fn = problem.residual(x)
        Fr = prob.residual(x, xk, alpha)
        assert _rel(F, Fr) < 1e-12 and abs(fn - np.linalg.norm(Fr)) <= 1e-12 * np.linalg.norm(Fr)
        J = problem.jacobian(x)
        Jr = prob.jacobian(x, alpha).tocsr()
        assert abs(J - Jr).max() <= 1e-12 * abs(Jr).max()
        n = prob.nv
        for blk in ((slice(n, 2 * n), slice(2 * n, 3 * n)), (slice(2 * n, 3 * n), slice(2 * n, 3 * n)), (slice(2 * n, 3 * n), slice(n, 2 * n))):
            assert abs(J[blk] - Jr[blk]).max() <= 1e-12 * abs(Jr[blk]).max()
        v = rng.standard_normal(prob.ntot)
        assert _rel(problem.spmv(v), Jr @ v) < 1e-12
    problem.set_state(x)
    problem.set_prev(xk)
    assert abs(problem.h1_increment() - prob.h1_increment(x, xk)) <= 1e-12 * prob.h1_increment(x, xk)
    problem.close()


@pytest.mark.parametrize("M", [8, 20])
def test_full_lvpp_run_matches_oracle(require_gpu, M):
    from proximalgalerkin_amd.thermoforming import solve_problem

    its, diffs, x = solve_problem(M, verbose=False, return_solution=True)
    coords, cells = O.create_rectangle(M, M, (0.0, 0.0), (1.0, 1.0))
    prob = Q.Thermoforming(coords, cells, O.boundary_vertices_rectangle(M, M))
    x_ref, its_ref, diffs_ref = Q.solve_problem(prob)
    assert list(its) == list(its_ref)
    assert _rel(x[: prob.nv], x_ref[: prob.nv]) < 1e-9


def test_problem_stated_as_forms_runs_the_same_solve(require_gpu):
    """The reference script's own statement (UFL forms, modified Jacobian form, live Constants/Functions;
    thermoforming_dolfinx.py:23-160) through the front end (proximalgalerkin_amd/ufl.py) selects the same HIP path as the
    direct host mirror: identical Newton counts, same final state."""
    from proximalgalerkin_amd import thermoforming as tf

    its_a, _, xa = tf.solve_problem(16, verbose=False, return_solution=True)
    its_b, xb = tf.solve_problem_forms(16)
    assert list(its_a) == list(its_b)
    assert np.linalg.norm(xa - xb) <= 1e-10 * np.linalg.norm(xa)
