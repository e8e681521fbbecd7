"""CPU twins (scipy SuperLU) of the comparison solvers in proximalgalerkin_amd/optimization.py - projected Newton for the
bound-constrained quadratic programme and the primal-dual active-set / semismooth Newton method for the obstacle VI (reference
counterparts: Galahad TRB through /root/reference/src/lvpp/optimization.py:42-96 and PETSc vinewtonssls through
examples/01_obstacle_problem/obstacle_snes.py:83-115; those are external packages, so the comparison here is algorithm against
algorithm: same iterates, same iteration counts).  TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench cpu leg)."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def _masked(A, bound):
    A = sp.csr_matrix(A)
    keep = sp.diags((~bound).astype(np.float64))
    return (keep @ A @ keep + sp.diags(bound.astype(np.float64))).tocsc()


def projected_newton(S, Mf, lo, up, x0, max_iter=100, tol=1e-6):
    S = sp.csr_matrix(S)
    proj = lambda z: np.minimum(np.maximum(z, lo), up)  # noqa: E731
    obj = lambda x: 0.5 * x @ (S @ x) - Mf @ x  # noqa: E731
    x = proj(np.asarray(x0, dtype=np.float64))
    g = S @ x - Mf
    pg0 = np.linalg.norm(proj(x - g) - x)
    it = 0
    for it in range(1, max_iter + 1):
        pg = np.linalg.norm(proj(x - g) - x)
        if pg <= tol * max(1.0, pg0):
            it -= 1
            break
        eps = min(1e-8, pg)
        bound = ((x <= lo + eps) & (g > 0.0)) | ((x >= up - eps) & (g < 0.0)) | (lo == up)
        d = spla.splu(_masked(S, bound)).solve(np.where(bound, 0.0, -g))
        d[bound] = 0.0
        f0, t = obj(x), 1.0
        for _ in range(60):
            xn = proj(x + t * d)
            if obj(xn) <= f0 + 1e-4 * (g @ (xn - x)):
                break
            t *= 0.5
        x = xn
        g = S @ x - Mf
    return x, it


def primal_dual_active_set(S, b, lower, upper=None, max_it=1000, c=1.0):
    S = sp.csr_matrix(S)
    n = S.shape[0]
    upper = np.full(n, np.inf) if upper is None else upper
    fixed = lower == upper
    u = np.clip(np.zeros(n), lower, upper)
    lam = np.zeros(n)
    prev = None
    it = 0
    sets = []
    for it in range(1, max_it + 1):
        active = (lam - c * (u - lower) > 0.0) | fixed
        if prev is not None and np.array_equal(active, prev):
            it -= 1
            break
        ua = np.where(active, lower, 0.0)
        u = spla.splu(_masked(S, active)).solve(np.where(active, lower, b - S @ ua))
        u[active] = lower[active]
        lam = np.where(active & ~fixed, S @ u - b, 0.0)
        prev = active
        sets.append(int(active.sum()))
    return u, it, sets
