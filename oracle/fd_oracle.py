"""Finite-difference LVPP for the obstacle problem: a numpy/scipy TRANSCRIPTION of the reference's only
self-contained statement of the algorithm,
/root/reference/examples/01_obstacle_problem/obstacle_finite_difference.jl (Julia is not installed here, so the
script itself cannot run; this file follows it line by line).

TEST INFRASTRUCTURE ONLY (same rule as oracle/pg_oracle.py: tests/, smoke() and bench.py's cpu_baseline leg).

What it pins.  The FD scheme is, row for row, the P1 finite-element LVPP system of obstacle_pg.py:116-125 on the
right-diagonal triangulation of the same grid when every integral uses the VERTEX quadrature rule (points = the
three vertices, weights 1/6): the P1 stiffness matrix of a uniform right-triangle mesh IS the 5-point stencil
(K = h^2 Lap_h), the mass matrix and D(psi) become diag(m_i) and diag(m_i e^{psi_i}) with m_i = h^2 at interior
vertices, and `(phi, w)` becomes m_i phi(x_i).  The script scales its 1-D stencil by (N-1)^2 = 4/h^2 on a grid of
spacing h = 2/(N-1) (obstacle_finite_difference.jl:48-50), i.e. A_fd = 4 Lap_h, so

      FE rows with alpha_fe = 4 alpha_fd   ==   diag(m_i) x (FD rows with alpha_fd)          (interior rows)

and, a row scaling leaving Newton increments unchanged, the two produce the SAME iterates.  tests/test_oracle_fd.py
checks that for oracle/pg_oracle.py, tests/test_gpu_fd_pin.py for the HIP path: the only comparison in this
repository whose right-hand side is an algorithm the REFERENCE wrote down in full (no FEniCSx/PETSc underneath).

Line map (obstacle_finite_difference.jl):  phi :13-27 | residual :29-35 | jacobian :37-43 | grid, stencil, boundary
index set :46-62 | data, initial values (psi = 1, u = w = 0) :65-68 | alpha rule :71,78 | Newton loop with the 1e-4
relative-residual test and at most 50 steps :80-102 | w <- psi :103 | stop at |u - u_|_2 < 1e-9 :106-110.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def phi(x, y):
    """obstacle_finite_difference.jl:13-27 (the same profile as obstacle_pg.py:92-104)."""
    r = np.sqrt(x * x + y * y)
    r0, beta = 0.5, 0.9
    b = r0 * beta
    t = np.sqrt(r0 * r0 - b * b)
    B = t + b * b / t
    C = -b / t
    return np.where(r > b, B + C * r, np.sqrt(np.maximum(r0 * r0 - r * r, 0.0)))


class FDProblem:
    """Grid, 5-point matrix, boundary index set and data of fd_lvpp_solve(N) (:46-66).  Unknown i + j*N sits at
    (xx[i], xx[j]) (Julia's column-major `vec`)."""

    def __init__(self, N: int):
        self.N = N
        self.xx = np.linspace(-1.0, 1.0, N)
        A1 = sp.diags([-np.ones(N - 1), 2.0 * np.ones(N), -np.ones(N - 1)], [-1, 0, 1]) * float((N - 1) ** 2)
        I1 = sp.identity(N)
        self.A = (sp.kron(A1, I1) + sp.kron(I1, A1)).tocsr()
        self.n = N * N
        i, j = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
        on = (i == 0) | (i == N - 1) | (j == 0) | (j == N - 1)
        self.bcs = np.flatnonzero(on.ravel(order="F"))
        X, Y = np.meshgrid(self.xx, self.xx, indexing="ij")
        self.phiv = phi(X, Y).ravel(order="F")
        self.fv = np.zeros(self.n)
        keep = np.ones(self.n)
        keep[self.bcs] = 0.0
        self._keep = sp.diags(keep)
        self._bcdiag = sp.diags(1.0 - keep)

    def residual(self, u, psi, alpha, w):
        """:29-35"""
        g = np.concatenate([alpha * (self.A @ u) + psi - alpha * self.fv - w, u - np.exp(psi) - self.phiv])
        g[self.bcs] = 0.0
        return g

    def jacobian(self, alpha, psi):
        """:37-43 - [[alpha A, I],[I, -diag(e^psi)]] with the rows and columns in `bcs` replaced by identity"""
        Kp = self._keep
        Auu = Kp @ (alpha * self.A) @ Kp + self._bcdiag
        return sp.bmat([[Auu, Kp], [Kp, -sp.diags(np.exp(psi))]], format="csc")


def alpha_rule(k: int, alpha: float, C=1.0, r=1.5, q=1.5) -> float:
    """:78 - NOTE the capped alpha is carried (obstacle_pg.py:177-183 carries the uncapped value)."""
    try:
        grown = C * r ** (q**k)
    except OverflowError:  # Julia: Inf
        grown = np.inf
    return float(min(max(grown - alpha, C), 1e2))


def fd_lvpp_solve(N: int, newton_step=None, max_outer: int = 101, record=None):
    """fd_lvpp_solve(N) (:45-112).  Returns (xx, U (N,N), newton_its, newton_per_step).

    newton_step(u, psi, alpha, w) -> (u_new, psi_new): replaces the two lines `dz = J \\ b; u += ...; psi += ...`
    (:90-95) so that another implementation of the SAME Newton step (the FE oracle with the vertex rule, the HIP path)
    can be driven through this loop; every decision (residual test, stopping test, alpha) stays here.
    record: optional list receiving (k, iter, u, psi) after every Newton step."""
    P = FDProblem(N)
    n = P.n
    psi, w, u, u_ = np.ones(n), np.zeros(n), np.zeros(n), np.zeros(n)
    alpha = 1.0
    newton_its, per_step = 0, []
    for k in range(max_outer):
        alpha = alpha_rule(k, alpha)
        b = -P.residual(u, psi, alpha, w)
        normres0 = np.linalg.norm(b)
        its = 0
        for _ in range(50):
            if newton_step is None:
                dz = spla.splu(P.jacobian(alpha, psi)).solve(b)
                u = u + dz[:n]
                psi = psi + dz[n:]
            else:
                u, psi = newton_step(u, psi, alpha, w)
            newton_its += 1
            its += 1
            if record is not None:
                record.append((k, its, u.copy(), psi.copy()))
            b = -P.residual(u, psi, alpha, w)
            if np.linalg.norm(b) / normres0 < 1e-4:
                break
        per_step.append(its)
        w = psi.copy()
        if np.linalg.norm(u - u_) < 1e-9:
            break
        u_ = u.copy()
    return P.xx, u.reshape(N, N, order="F"), newton_its, per_step
