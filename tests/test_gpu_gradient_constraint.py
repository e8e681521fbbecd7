"""Example 06 (gradient constraint, vector latent variable) HIP path vs the CPU oracle (oracle/gc_oracle.py), through
the C ABI of include/pgx_gc.h.  Tolerances: element kernels 1e-12 relative (fp64 atomics reorder sums); full LVPP run:
identical Newton counts per proximal step, final primal field <= 1e-10 relative L2."""
import numpy as np
import pytest

from oracle import gc_oracle as G
from oracle import pg_oracle as O

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _setup(N, M=None):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.gradient_constraint import GradientConstraintProblem, f_default, phi_default

    M = N if M is None else M
    problem = GradientConstraintProblem(fem.create_unit_square(N, M), phi_default, f_default)
    coords, cells = O.create_rectangle(N, M, (0.0, 0.0), (1.0, 1.0))
    prob = G.GradientConstraintP2(coords, cells)
    assert problem.ndofs == prob.ntot
    return problem, prob


@pytest.mark.parametrize("N,M", [(3, 3), (9, 6), (24, 24)])
def test_kernels_match_oracle(require_gpu, N, M):
    problem, prob = _setup(N, M)
    rng = np.random.default_rng(7)
    x = rng.standard_normal(prob.ntot)
    x[prob.n2:] *= np.where(rng.random(2 * prob.nv) < 0.3, 500.0, 2.0)  # |psi| spans what real runs meet
    xk = rng.standard_normal(prob.ntot) * 0.1
    for alpha in (1.0, 64.0):
        problem.set_alpha(alpha)
        problem.set_prev(xk)
        F, fn = problem.residual(x)
        Fr = prob.residual(x, xk, alpha)
        assert _rel(F, Fr) < 1e-12
        assert abs(fn - np.linalg.norm(Fr)) <= 1e-12 * np.linalg.norm(Fr)
        J = problem.jacobian(x)
        Jr = prob.jacobian(x, alpha).tocsr()
        d = (J - Jr)
        assert abs(d).max() <= 1e-12 * abs(Jr).max()
        # entry-wise check of the tiny N block (values down to phi/s^3 ~ 1e-9 relative to the K block)
        n2 = prob.n2
        assert abs(d[n2:, n2:]).max() <= 1e-12 * abs(Jr[n2:, n2:]).max()
        v = rng.standard_normal(prob.ntot)
        assert _rel(problem.spmv(v), Jr @ v) < 1e-12
    problem.set_state(x)
    problem.set_prev(xk)
    assert abs(problem.l2_increment() - prob.l2_increment(x, xk)) <= 1e-12 * prob.l2_increment(x, xk)
    problem.close()


@pytest.mark.parametrize("N", [8, 20])
def test_full_lvpp_run_matches_oracle(require_gpu, N):
    from proximalgalerkin_amd.gradient_constraint import solve_problem

    its, diffs, x = solve_problem(N, N, verbose=False, return_solution=True)
    coords, cells = O.create_rectangle(N, N, (0.0, 0.0), (1.0, 1.0))
    prob = G.GradientConstraintP2(coords, cells)
    x_ref, its_ref, diffs_ref = G.solve_problem(prob)
    assert list(its) == list(its_ref)
    assert _rel(x[: prob.n2], x_ref[: prob.n2]) < 1e-10
    assert np.allclose(diffs, diffs_ref, rtol=1e-6, atol=1e-13)


def test_warm_start_matches_oracle(require_gpu):
    """--warm_start (gradient_constraint_dolfinx.py:72-96): Poisson pre-solve on the GPU (sparse LU of the stiffness block)
    vs the oracle's; same Newton counts from the warm start, and they differ from the cold start's first step."""
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.gradient_constraint import GradientConstraintProblem, f_default, phi_default, solve_problem

    N = 12
    coords, cells = O.create_rectangle(N, N, (0.0, 0.0), (1.0, 1.0))
    prob = G.GradientConstraintP2(coords, cells)
    problem = GradientConstraintProblem(fem.create_unit_square(N, N), phi_default, f_default)
    x0 = problem.warm_start()
    import scipy.sparse.linalg as spla

    z = np.zeros(prob.ntot)
    u_ref = spla.spsolve(prob.jacobian(z, 1.0)[: prob.n2, : prob.n2].tocsc(), -prob.residual(z, z, 1.0)[: prob.n2])
    assert _rel(x0[: prob.n2], u_ref) < 1e-12 and not x0[prob.n2:].any()
    assert np.array_equal(problem.get_state(), x0)
    problem.close()
    its, diffs, x = solve_problem(N, N, warm_start=True, verbose=False, return_solution=True)
    x_ref, its_ref, diffs_ref = G.solve_problem(prob, warm_start=True)
    assert list(its) == list(its_ref)
    assert _rel(x[: prob.n2], x_ref[: prob.n2]) < 1e-10
    assert np.allclose(diffs, diffs_ref, rtol=1e-6, atol=1e-13)


def test_other_alpha_schemes(require_gpu):
    from proximalgalerkin_amd.gradient_constraint import solve_problem

    N = 10
    coords, cells = O.create_rectangle(N, N, (0.0, 0.0), (1.0, 1.0))
    prob = G.GradientConstraintP2(coords, cells)
    for scheme, kw in (("linear", dict(alpha_c=3.0, max_iterations=12)), ("constant", dict(alpha_0=4.0, max_iterations=6))):
        its, diffs, x = solve_problem(N, N, alpha_scheme=scheme, verbose=False, return_solution=True, stopping_tol=1e-7, **kw)
        x_ref, its_ref, _ = G.solve_problem(prob, alpha_scheme=scheme, alpha_0=kw.get("alpha_0", 1.0),
                                            alpha_c=kw.get("alpha_c", 1.0), max_iterations=kw["max_iterations"],
                                            stopping_tol=1e-7)
        assert list(its) == list(its_ref)
        assert _rel(x[: prob.n2], x_ref[: prob.n2]) < 1e-10


def test_hip_reproduces_golden_fixture(require_gpu):
    """tests/golden/gradient_constraint_p2_n12_defaults.npz (tools/make_golden.py): kernels at a fixed iterate, then the
    complete LVPP run."""
    import pathlib

    from proximalgalerkin_amd.gradient_constraint import solve_problem

    z = np.load(pathlib.Path(__file__).resolve().parent / "golden" / "gradient_constraint_p2_n12_defaults.npz")
    N = int(z["N"])
    problem, prob = _setup(N)
    problem.set_alpha(5.0)
    problem.set_prev(z["xk_iter"])
    F, _ = problem.residual(z["x_iter"])
    assert _rel(F, z["F_iter"]) < 1e-12
    problem.jacobian(z["x_iter"])
    assert _rel(problem.spmv(z["v"]), z["Jv_iter"]) < 1e-12
    problem.set_state(z["x_iter"])
    assert abs(problem.l2_increment() - float(z["l2_iter"])) <= 1e-12 * float(z["l2_iter"])
    problem.close()
    its, diffs, x = solve_problem(N, N, verbose=False, return_solution=True)
    assert list(its) == list(z["newton"])
    assert _rel(x[: prob.n2], z["x_final"][: prob.n2]) < 1e-10


def test_gmres_safeguard_of_the_linear_solve(require_gpu, monkeypatch):
    """Where LU + refinement cannot reach a true relative residual of 1e-7 (seen at 2048^2, alpha = 1024) the same LU
    preconditions a GMRES on the exact operator (pgx_mixed.h::mx_gmres_lu).  Forced here: no refinement steps, GMRES polishes
    every solve to 1e-12 - the LVPP run must not change."""
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.gradient_constraint import GradientConstraintProblem, f_default, phi_default

    N = 16
    coords, cells = O.create_rectangle(N, N, (0.0, 0.0), (1.0, 1.0))
    prob = G.GradientConstraintP2(coords, cells)
    x_ref, its_ref, _ = G.solve_problem(prob)
    monkeypatch.setenv("PGX_MX_GMRES_ALWAYS", "1")
    problem = GradientConstraintProblem(fem.create_unit_square(N, N), phi_default, f_default)
    problem._opts.ksp_max_it = 1  # plain LU solve, no refinement
    problem.profile(True)
    its, lin = [], 0
    for i in range(25):
        problem.set_alpha(2.0**i)
        reason, n = problem.solve()
        assert reason > 0
        its.append(n)
        lin += problem.solver.ksp._its  # LU solves of this Newton solve
        d = problem.l2_increment()
        if d < 1e-8:
            break
        problem.advance_prev()
    x = problem.get_state()
    problem.close()
    assert its == list(its_ref)
    assert _rel(x[: prob.n2], x_ref[: prob.n2]) < 1e-10
    assert lin > sum(its)  # more LU solves than Newton steps although refinement is off: GMRES ran


def test_problem_stated_as_forms_runs_the_same_solve(require_gpu):
    """gradient_constraint_dolfinx.py:36-132 stated through the UFL-subset front end (elements, mixed space, collapsed coefficient
    space, residual form, NonlinearProblem) reproduces the declarative driver: Newton counts and iterate."""
    from proximalgalerkin_amd.gradient_constraint import solve_problem, solve_problem_forms

    its, diffs, x = solve_problem(10, 10, verbose=False, return_solution=True)
    its_f, diffs_f, sol = solve_problem_forms(10, 10)
    assert list(its_f) == list(its)
    assert np.allclose(diffs_f, diffs, rtol=1e-9, atol=1e-14)
    assert np.linalg.norm(sol.x.array - x) <= 1e-12 * np.linalg.norm(x)


# ------------------------------------------------------------------------------------------------------------------
# general primal degree (gradient_constraint_dolfinx.py:245-250: 2..8), latent degree k - 1: table-driven kernels
# ------------------------------------------------------------------------------------------------------------------
def _general(N, k):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.gradient_constraint import GradientConstraintProblem, f_default, phi_default

    mesh = fem.create_unit_square(N, N)
    problem = GradientConstraintProblem(mesh, phi_default, f_default, degree=k, general=True)
    c, e = O.create_rectangle(N, N, (0.0, 0.0), (1.0, 1.0))
    prob = G.GradientConstraintPk(c, e, k)
    assert problem.ndofs == prob.ntot and problem.n2 == prob.n2 and problem.nv == prob.nv
    return problem, prob


@pytest.mark.parametrize("N,k", [(4, 2), (5, 3), (4, 4), (3, 6), (2, 8)])
def test_general_degree_kernels_match_oracle(require_gpu, N, k):
    problem, prob = _general(N, k)
    rng = np.random.default_rng(20 + k)
    x = rng.standard_normal(prob.ntot) * 0.3
    x[prob.n2:] *= np.where(rng.random(2 * prob.nv) < 0.3, 300.0, 1.0)  # some large latent values: s = sqrt(1 + |psi|^2) >> 1
    xk = rng.standard_normal(prob.ntot) * 0.3
    for alpha in (1.0, 32.0):
        problem.set_alpha(alpha)
        problem.set_prev(xk)
        F, fn = problem.residual(x)
        Fr = prob.residual(x, xk, alpha)
        assert np.linalg.norm(F - Fr) <= 1e-11 * np.linalg.norm(Fr), np.linalg.norm(F - Fr) / np.linalg.norm(Fr)
        J = problem.jacobian(x)
        Jr = prob.jacobian(x, alpha).tocsr()
        assert abs(J - Jr).max() <= 1e-11 * abs(Jr).max()
        n2 = prob.n2
        assert abs(J[n2:, n2:] - Jr[n2:, n2:]).max() <= 1e-11 * abs(Jr[n2:, n2:]).max()
    assert abs(problem.l2_increment() - 0.0) >= 0.0
    problem.close()


@pytest.mark.parametrize("N,k", [(6, 2), (16, 3), (12, 4), (3, 5), (8, 6), (4, 8)])
def test_general_degree_full_run_matches_oracle(require_gpu, N, k):
    from proximalgalerkin_amd.gradient_constraint import solve_problem

    its, _, x = solve_problem(N, N, primal_degree=k, verbose=False, return_solution=True)
    c, e = O.create_rectangle(N, N, (0.0, 0.0), (1.0, 1.0))
    prob = G.GradientConstraintPk(c, e, k)
    xr, its_r, _ = G.solve_problem(prob)
    assert list(its) == list(its_r), (list(its), list(its_r))
    n2 = prob.n2
    # k <= 6: 1e-9 (SNES rtol = atol = 1e-9).  k = 8 agrees to 8e-8 only: the reference's fixed degree-10 rule (:53, 36 points) cannot
    # resolve the latent block of a (P7)^2 element (72 x 72 per cell from 36 points), the late Newton matrices are singular to working
    # precision along those modes and two direct solvers pick different representatives - same Newton counts, same u to 7 digits
    tol = 1e-9 if k <= 6 else 1e-6
    assert np.linalg.norm(x[:n2] - xr[:n2]) <= tol * np.linalg.norm(xr[:n2])


# ------------------------------------------------------------------------------------------------------------------
# --cell_type quadrilateral (gradient_constraint_dolfinx.py:229-236): Q_k / (Q_(k-1))^2 on the grid of rectangles, same table-driven
# kernels (affine cells), tensor Gauss-Legendre rule of degree 10
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,M,k", [(4, 3, 2), (3, 4, 3), (3, 3, 5), (2, 2, 6), (2, 1, 8)])
def test_quadrilateral_kernels_match_oracle(require_gpu, N, M, k):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.gradient_constraint import GradientConstraintProblem, f_default, phi_default

    problem = GradientConstraintProblem(fem.create_unit_square(N, M, "quadrilateral"), phi_default, f_default, degree=k)
    prob = G.GradientConstraintQk(N, M, k)
    assert problem.ndofs == prob.ntot and problem.n2 == prob.n2 and problem.nv == prob.nv
    rng = np.random.default_rng(40 + k)
    x = rng.standard_normal(prob.ntot) * 0.3
    x[prob.n2:] *= np.where(rng.random(2 * prob.nv) < 0.3, 300.0, 1.0)
    xk = rng.standard_normal(prob.ntot) * 0.3
    for alpha in (1.0, 32.0):
        problem.set_alpha(alpha)
        problem.set_prev(xk)
        F, _ = problem.residual(x)
        Fr = prob.residual(x, xk, alpha)
        assert np.linalg.norm(F - Fr) <= 1e-11 * np.linalg.norm(Fr), np.linalg.norm(F - Fr) / np.linalg.norm(Fr)
        J = problem.jacobian(x)
        Jr = prob.jacobian(x, alpha).tocsr()
        assert abs(J - Jr).max() <= 1e-11 * abs(Jr).max()
    problem.set_state(x)
    d = x[:prob.n2] - xk[:prob.n2]
    assert abs(problem.l2_increment() - np.sqrt(d @ (prob.M2 @ d))) <= 1e-11 * np.sqrt(d @ (prob.M2 @ d))
    problem.close()


@pytest.mark.parametrize("N,k", [(8, 2), (10, 3), (5, 4), (3, 6)])
def test_quadrilateral_full_run_matches_oracle(require_gpu, N, k):
    from proximalgalerkin_amd.gradient_constraint import solve_problem

    its, _, x = solve_problem(N, N, primal_degree=k, cell_type="quadrilateral", verbose=False, return_solution=True)
    prob = G.GradientConstraintQk(N, N, k)
    xr, its_r, _ = G.solve_problem(prob)
    assert list(its) == list(its_r), (list(its), list(its_r))
    n2 = prob.n2
    assert np.linalg.norm(x[:n2] - xr[:n2]) <= 1e-9 * np.linalg.norm(xr[:n2])


def test_symmetric_factorisation_agrees_with_the_general_lu(require_gpu, monkeypatch):
    """Round 5: example 06's Newton matrix [[alpha K, G^T], [G, -N(psi)]] is symmetric and its sparse LU runs in symmetric mode (L D L^T
    in LU clothing, include/pgx_nd.h).  PGX_ND_SYM=0 ignores the request: same Newton counts per proximal step, same primal field."""
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.gradient_constraint import GradientConstraintProblem, f_default, phi_default, solve_problem

    out = {}
    for sym in ("1", "0"):
        monkeypatch.setenv("PGX_ND_SYM", sym)
        p = GradientConstraintProblem(fem.create_unit_square(24, 24), phi_default, f_default)
        assert p.lu_stats()["symmetric"] is (sym == "1")
        p.close()
        out[sym] = solve_problem(24, 24, verbose=False, return_solution=True)
    (ia, _, xa), (ib, _, xb) = out["1"], out["0"]
    assert list(ia) == list(ib)
    assert np.linalg.norm(np.asarray(xa) - np.asarray(xb)) <= 1e-10 * np.linalg.norm(np.asarray(xb))
