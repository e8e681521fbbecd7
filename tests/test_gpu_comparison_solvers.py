"""Comparison solvers of example 01 on the GPU (proximalgalerkin_amd/optimization.py; reference: lvpp/optimization.py,
obstacle_ipopt_galahad.py, obstacle_snes.py, compare_all.py) against their CPU twins (oracle/compare_oracle.py): the same
algorithm with SuperLU instead of the GPU sparse LU - identical iteration counts, solutions <= 1e-10 - and against the
proximal-Galerkin solution of the same mesh (a different discretisation of the constraint: agreement to O(h^2))."""
import numpy as np
import pytest

from oracle import compare_oracle as C

pytestmark = pytest.mark.gpu


def _setup(N):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.optimization import ObstacleProblem, setup_problem

    mesh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
    S, M, f, bounds, coords = setup_problem(mesh)
    return mesh, S, M, f, bounds, coords, ObstacleProblem(S, M, f)


@pytest.mark.parametrize("N", [16, 48])
def test_matrices_from_the_hip_assembly_match_the_oracle(require_gpu, N):
    from oracle import pg_oracle as O

    mesh, S, M, f, bounds, coords, _ = _setup(N)
    c, e = O.create_rectangle(N, N)
    p = O.ObstacleP1(c, e, O.boundary_vertices_rectangle(N, N))
    assert abs(S - p.K).max() <= 1e-13 * abs(p.K).max() and abs(M - p.M).max() <= 1e-13 * abs(p.M).max()
    assert np.array_equal(np.flatnonzero(bounds[0] == bounds[1]), np.sort(p.bc))


@pytest.mark.parametrize("N", [24, 64])
def test_vi_semismooth_newton_matches_the_cpu_twin(require_gpu, N):
    from proximalgalerkin_amd.optimization import vi_newton_solver

    mesh, S, M, f, (lo, up), coords, _ = _setup(N)
    u, it = vi_newton_solver(S, M @ f, lo, up, coords=coords)
    ur, itr, _ = C.primal_dual_active_set(S, M @ f, lo, up)
    assert it == itr
    assert np.linalg.norm(u - ur) <= 1e-10 * np.linalg.norm(ur)


@pytest.mark.parametrize("N", [24, 64])
def test_trust_region_slot_matches_the_cpu_twin_and_the_vi_solution(require_gpu, N):
    from proximalgalerkin_amd.optimization import galahad_solver, vi_newton_solver

    mesh, S, M, f, (lo, up), coords, problem = _setup(N)
    x, it = galahad_solver(problem, np.zeros(len(f)), (lo, up), log_level=0, tol=1e-10, coords=coords)
    xr, itr = C.projected_newton(S, M @ f, lo, up, np.zeros(len(f)), tol=1e-10)
    assert it == itr and problem.total_iteration_count == it
    assert np.linalg.norm(x - xr) <= 1e-10 * np.linalg.norm(xr)
    u, _ = vi_newton_solver(S, M @ f, lo, up, coords=coords)
    assert np.abs(x - u).max() <= 1e-8  # two methods, one discrete problem


def test_ipopt_slot_has_the_reference_call_shape(require_gpu):
    """lvpp.optimization.ipopt_solver(problem, x_init, bounds, log_level, max_iter, tol, activate_hessian) -> x, the iteration count
    read from problem.total_iteration_count (compare_all.py:100-135): same solution as the trust-region slot."""
    from proximalgalerkin_amd.optimization import galahad_solver, ipopt_solver

    mesh, S, M, f, (lo, up), coords, problem = _setup(24)
    x = ipopt_solver(problem, np.zeros(len(f)), (lo, up), log_level=0, max_iter=100, tol=1e-10, activate_hessian=True, coords=coords)
    it = problem.total_iteration_count
    xg, itg = galahad_solver(problem, np.zeros(len(f)), (lo, up), log_level=0, tol=1e-10, coords=coords)
    assert isinstance(x, np.ndarray) and it == itg and np.array_equal(x, xg)
    x1 = ipopt_solver(problem, np.zeros(len(f)), (lo, up), log_level=0, max_iter=20000, tol=1e-5, activate_hessian=False)
    assert problem.total_iteration_count > it and np.abs(x1 - x).max() < 1e-3  # first-order method: many more iterations, same minimiser


def test_iteration_table_and_agreement_with_proximal_galerkin(require_gpu):
    """compare_all.py's experiment on one mesh: every solver converges, the second-order methods in a handful of iterations, and
    the proximal-Galerkin solution (quadrature of exp(psi), a different discrete constraint) agrees with the VI solution to
    the discretisation error."""
    from proximalgalerkin_amd.obstacle import solve_problem
    from proximalgalerkin_amd.optimization import galahad_solver, vi_newton_solver

    N = 64
    mesh, S, M, f, (lo, up), coords, problem = _setup(N)
    x_g, it_g = galahad_solver(problem, np.zeros(len(f)), (lo, up), log_level=0, max_iter=500, tol=1e-4, coords=coords)
    u_vi, it_vi = vi_newton_solver(S, M @ f, lo, up, coords=coords)
    sol, it_pg = solve_problem(mesh, 1, 500, "double_exponential", 1e2, 1e-4, verbose=False)
    x_f, it_f = galahad_solver(problem, np.zeros(len(f)), (lo, up), log_level=0, use_hessian=False, max_iter=20000, tol=1e-4)
    n = mesh.num_vertices
    assert it_g <= 15 and it_vi <= 15 and 15 <= it_pg <= 30 and it_f > 10 * it_g
    assert np.abs(sol.x.array[:n] - u_vi).max() < 5e-3
    assert np.abs(x_f - u_vi).max() < 5e-3
