"""Comparison solvers of example 01 on the GPU.

The reference compares proximal Galerkin with three OTHER solvers of the same discrete obstacle problem
    min  1/2 u^T S u - f^T M u     subject to   lower <= u <= upper          (S, M: P1 stiffness / mass matrices)
- Galahad TRB (trust-region, bound constraints) and IPOPT through /root/reference/src/lvpp/optimization.py:13-179 with the problem
object of examples/01_obstacle_problem/obstacle_ipopt_galahad.py:44-127, and PETSc's semismooth VI Newton
(obstacle_snes.py:36-115, `snes_type vinewtonssls` + setVariableBounds) - and prints an iteration-count table
(compare_all.py:170-182).  Those packages are external optimisers; here their role is played by two solvers written for this
repository whose linear algebra - every Newton system - is the nested-dissection sparse LU on the GPU (include/pgx_nd.h):

    trb_solver(problem, x_init, bounds, ...)     projected Newton with Armijo search along the projection arc (Bertsekas 1982):
                                                 the bound-constrained second-order method in the slot of `galahad_solver`
    vi_newton_solver(S, b, lower, upper, ...)    primal-dual active set = semismooth Newton on the NCP function
                                                 lambda - max(0, lambda - c (u - lower)) (Hintermueller, Ito, Kunisch 2002): the
                                                 VI Newton method in the slot of `vinewtonssls`

`OptimizationProblem` keeps the reference's protocol (objective / gradient / pure_hessian / hessianstructure), so a problem object
written for lvpp.optimization works unchanged; `galahad_solver` is kept as an alias of `trb_solver` with the reference's signature.
Problem callbacks are host numpy/scipy exactly as in the reference; the factorisations and triangular solves run on the GPU and
fail loudly without one.  CPU twins for the parity tests: oracle/compare_oracle.py.
"""
from __future__ import annotations

import typing

import numpy as np
import scipy.sparse as sp

from .direct import DirectSolver

__all__ = ["OptimizationProblem", "ObstacleProblem", "trb_solver", "galahad_solver", "vi_newton_solver", "setup_problem"]


class OptimizationProblem(typing.Protocol):
    """lvpp.optimization.OptimizationProblem (src/lvpp/optimization.py:13-36)."""
    total_iteration_count: int

    def objective(self, x): ...

    def gradient(self, x): ...

    def pure_hessian(self, x): ...

    def hessian(self, x, lagrange, obj_factor):
        return obj_factor * self.pure_hessian(x)

    def hessianstructure(self): ...


class ObstacleProblem:
    """obstacle_ipopt_galahad.py:94-127: quadratic energy 1/2 x^T S x - f^T M x with the lower-triangular Hessian in coordinate form."""
    total_iteration_count: int = 0

    def __init__(self, S, M, f):
        S = sp.csr_matrix(S, copy=True)
        S.eliminate_zeros()
        self._S, self._M = S, sp.csr_matrix(M)
        self._f = np.asarray(f, dtype=np.float64)
        self._Mf = self._M @ self._f
        tri = sp.tril(self._S).tocoo()
        self._sparsity = (tri.row.astype(np.int32), tri.col.astype(np.int32))
        self._H_data = tri.data

    def objective(self, x):
        return 0.5 * x @ (self._S @ x) - self._f @ (self._M @ x)

    def gradient(self, x):
        return self._S @ x - self._Mf

    def pure_hessian(self, x):
        return self._H_data

    def hessian(self, x, lagrange, obj_factor):
        return obj_factor * self.pure_hessian(x)

    def hessianstructure(self):
        return self._sparsity

    def intermediate(self, *args):
        self.total_iteration_count = args[1]


def setup_problem(mesh, phi=None, device: int = 0):
    """obstacle_ipopt_galahad.setup_problem (:44-91) for a Mesh: P1 stiffness and mass matrices assembled by the HIP kernels of
    include/pgx.h (the K and M value streams of the shared CSR pattern), f = 0, bounds = (interpolated obstacle, +inf) with the
    Dirichlet dofs pinned to 0 in both.  Returns (S, M, f, (lower, upper), vertex coordinates)."""
    from .obstacle import phi_set, setup_problem as _setup

    problem, sol, sol_k, alpha = _setup(mesh, 1, device=device)
    rowptr, col, K, M, _ = problem.export_blocks(with_D=False)
    problem.close()
    n = mesh.num_vertices
    S = sp.csr_matrix((K, col.copy(), rowptr.copy()), shape=(n, n))  # separate index arrays: eliminate_zeros() works in place
    Mm = sp.csr_matrix((M, col.copy(), rowptr.copy()), shape=(n, n))
    lower = np.asarray((phi or phi_set)(mesh.geometry.T.copy()), dtype=np.float64).copy()
    upper = np.full(n, np.inf)
    bc = mesh.exterior_vertices()
    lower[bc] = 0.0
    upper[bc] = 0.0
    return S, Mm, np.zeros(n), (lower, upper), mesh.geometry


class _MaskedLU:
    """GPU sparse LU of  P_F A P_F + P_B  (rows / columns of the bound set B replaced by the identity) on a fixed pattern."""

    def __init__(self, A: sp.csr_matrix, coords, device=0):
        A = sp.csr_matrix(A)
        A.sort_indices()
        self.A = A
        n = A.shape[0]
        self.rows = np.repeat(np.arange(n), np.diff(A.indptr))
        self.diag = self.rows == A.indices
        self.lu = DirectSolver(A.indptr, A.indices, np.arange(n, dtype=np.int32), coords, device=device)
        # Hessians / stiffness matrices with rows AND columns of the bound set replaced by the identity: symmetric whenever A is
        if abs(A - A.T).max() <= 1e-12 * abs(A).max():
            self.lu.set_symmetric(True)

    def factor(self, bound: np.ndarray):
        keep = ~(bound[self.rows] | bound[self.A.indices])
        self.lu.factor(np.where(keep, self.A.data, 0.0) + (self.diag & bound[self.rows]))

    def solve(self, b):
        x = self.lu.solve(b)
        return x

    def close(self):
        self.lu.close()


def _coords_for(problem, coords, n):
    if coords is not None:
        return np.asarray(coords, dtype=np.float64)
    # no geometry given: a 1-D embedding by index still yields a valid (if poorer) dissection ordering
    return np.stack([np.arange(n, dtype=np.float64), np.zeros(n)], axis=1)


def _hessian_csr(problem, x, n):
    r, c = problem.hessianstructure()
    v = problem.pure_hessian(x)
    L = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsr()
    return (L + sp.tril(L, -1).T).tocsr()


def trb_solver(problem: OptimizationProblem, x_init, bounds, log_level: int = 0, use_hessian: bool = True, max_iter: int = 100,
               tol: float = 1e-6, coords=None, device: int = 0):
    """Bound-constrained minimisation in the slot of lvpp.optimization.galahad_solver (same arguments, same return value
    (x, iterations)): projected Newton - at x, bound variables whose gradient pushes outward are held, the Newton system of the
    others is solved by the GPU sparse LU, and an Armijo search runs along the PROJECTED path P(x + t d).  Stops when the
    projected gradient |P(x - g) - x| <= tol * max(1, |P(x0 - g0) - x0|) (Galahad's stop_pg_relative).  use_hessian=False:
    projected gradient steps (first-order model)."""
    lo, up = (np.asarray(b, dtype=np.float64) for b in bounds)
    x = np.clip(np.asarray(x_init, dtype=np.float64), lo, up)
    n = len(x)
    assert len(lo) == len(up) == n, "Bounds and x_init must have the same length"
    proj = lambda z: np.minimum(np.maximum(z, lo), up)  # noqa: E731
    H = _hessian_csr(problem, x, n)
    lu = _MaskedLU(H, _coords_for(problem, coords, n), device=device) if use_hessian else None
    g = problem.gradient(x)
    pg0 = np.linalg.norm(proj(x - g) - x)
    it = 0
    try:
        for it in range(1, max_iter + 1):
            pg = np.linalg.norm(proj(x - g) - x)
            if log_level:
                print(f"  trb {it - 1:3d}  f = {problem.objective(x):.12e}  |projected gradient| = {pg:.3e}")
            if pg <= tol * max(1.0, pg0):
                it -= 1
                break
            eps = min(1e-8, pg)
            bound = ((x <= lo + eps) & (g > 0.0)) | ((x >= up - eps) & (g < 0.0)) | (lo == up)
            if use_hessian:
                lu.factor(bound)
                d = lu.solve(np.where(bound, 0.0, -g))
                d[bound] = 0.0
            else:
                d = np.where(bound, 0.0, -g)
            f0 = problem.objective(x)
            t = 1.0
            for _ in range(60):
                xn = proj(x + t * d)
                if problem.objective(xn) <= f0 + 1e-4 * (g @ (xn - x)):
                    break
                t *= 0.5
            x = xn
            g = problem.gradient(x)
    finally:
        if lu is not None:
            lu.close()
    problem.total_iteration_count = it
    return x, it


def galahad_solver(problem, x_init, bounds, log_level: int = 1, use_hessian: bool = True, max_iter: int = 100, tol: float = 1e-6,
                   **kw):
    """Drop-in for lvpp.optimization.galahad_solver (src/lvpp/optimization.py:42-96): same call, solved by `trb_solver`."""
    return trb_solver(problem, x_init, bounds, log_level=max(log_level - 1, 0), use_hessian=use_hessian, max_iter=max_iter, tol=tol,
                      **kw)


def ipopt_solver(problem, x_init, bounds, log_level: int = 5, max_iter: int = 100, tol: float = 1e-6, activate_hessian: bool = True, **kw):
    """The call of lvpp.optimization.ipopt_solver (src/lvpp/optimization.py:115-166: same arguments, returns x only; the iteration
    count is left in `problem.total_iteration_count`, which is where the reference's callers read it, compare_all.py:100-108).  The
    bound-constrained problem is solved by `trb_solver`: projected Newton with `activate_hessian`, projected gradient without - the
    first-order method the reference compares against is IPOPT with its limited-memory Hessian approximation."""
    x, _ = trb_solver(problem, x_init, bounds, log_level=1 if log_level > 5 else 0, use_hessian=activate_hessian, max_iter=max_iter,
                      tol=tol, **kw)
    return x


def vi_newton_solver(S, b, lower, upper=None, x_init=None, max_it: int = 1000, c: float = 1.0, coords=None, device: int = 0,
                     monitor: bool = False):
    """Variational inequality  u >= lower,  S u - b >= 0,  (u - lower)^T (S u - b) = 0  (+ fixed dofs where lower == upper) by the
    primal-dual active-set method = semismooth Newton.  Each step fixes u = lower on the active set, solves the reduced system
    by the GPU sparse LU and updates the set from lambda = S u - b; it stops when the set repeats (the iterate then solves the
    VI exactly).  Returns (u, iterations) like obstacle_snes.snes_solve's (u, snes.getIterationNumber())."""
    S = sp.csr_matrix(S)
    n = S.shape[0]
    lower = np.asarray(lower, dtype=np.float64)
    upper = np.full(n, np.inf) if upper is None else np.asarray(upper, dtype=np.float64)
    fixed = lower == upper
    u = np.clip(np.zeros(n) if x_init is None else np.asarray(x_init, dtype=np.float64), lower, upper)
    lam = np.zeros(n)
    lu = _MaskedLU(S, _coords_for(None, coords, n), device=device)
    active_prev = None
    it = 0
    try:
        for it in range(1, max_it + 1):
            active = ((lam - c * (u - lower) > 0.0) | fixed)
            if active_prev is not None and np.array_equal(active, active_prev):
                it -= 1
                break
            ua = np.where(active, lower, 0.0)
            lu.factor(active)
            rhs = np.where(active, lower, b - S @ ua)
            u = lu.solve(rhs)
            u[active] = lower[active]
            lam = np.where(active & ~fixed, S @ u - b, 0.0)
            if monitor:
                print(f"  vi {it:3d}  active {int(active.sum()) - int(fixed.sum()):7d}  min(u - lower) = {(u - lower).min():.3e}  "
                      f"min(lambda) = {lam.min():.3e}")
            active_prev = active
    finally:
        lu.close()
    return u, it
