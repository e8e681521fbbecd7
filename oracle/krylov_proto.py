"""numpy/scipy prototype of the HIP path's Newton linear solver ("oracle-Krylov").

TEST INFRASTRUCTURE ONLY (see oracle/pg_oracle.py header).  It restates, on the CPU and with
scipy sparse matrices, the algorithm csrc/pgx.hip implements with stencil kernels, so that
tests can separate "is the algorithm right" from "is the kernel right":

  FGMRES (right-preconditioned, CGS2) on the exact Jacobian  [[aK, M],[M, -D(psi)]]  (+BC rows)
  preconditioner = one multigrid V(nu,nu) cycle on that same saddle-point matrix:
     * nested right-diagonal meshes N -> N/2 -> ... , P1 interpolation, Galerkin coarse operators
       (for nested P1 spaces these stay 7-point stencils)
     * smoother = damped *collective* Jacobi: at every vertex solve the 2x2 block
       [[a*K_ii, M_ii],[M_ii, -D_ii]] for (du_i, dpsi_i)   (det = -aK_ii D_ii - M_ii^2 < 0 always,
       so the block is invertible however small exp(psi) gets - nothing divides by D alone)

Why not what the reference does: the reference solves the Newton system with sparse LU (MUMPS,
/root/reference/examples/01_obstacle_problem/obstacle_pg.py:129-131); a direct factorisation of an
8.4M-unknown saddle point does not map to the GPU.  Precedent in the reference for an iterative
Newton solve: preconditioned GMRES in obstacle_spectral.jl:102-111, MINRES + block preconditioner
in examples/09_eikonal/ex40.cpp:261-281.  SURVEY.md section 0 finding 7 records the approaches
that fail (Jacobi-PCG on the Schur complement, MINRES + block-Jacobi).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def interp_matrix(N: int, M: int | None = None):
    """P1 prolongation from the (N/2)x(M/2) to the NxM right-diagonal mesh (vertex-based)."""
    M = N if M is None else M
    Nc, Mc = N // 2, M // 2
    nf, nc = (N + 1) * (M + 1), (Nc + 1) * (Mc + 1)
    I, J = np.meshgrid(np.arange(Nc + 1), np.arange(Mc + 1), indexing="xy")
    I, J = I.ravel(), J.ravel()
    cid = J * (Nc + 1) + I
    rows, cols, vals = [], [], []
    for di, dj, w in [(0, 0, 1.0), (1, 0, 0.5), (-1, 0, 0.5), (0, 1, 0.5), (0, -1, 0.5), (1, 1, 0.5), (-1, -1, 0.5)]:
        fi, fj = 2 * I + di, 2 * J + dj
        ok = (fi >= 0) & (fi <= N) & (fj >= 0) & (fj <= M)
        rows.append((fj * (N + 1) + fi)[ok])
        cols.append(cid[ok])
        vals.append(np.full(int(ok.sum()), w))
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nf, nc))


def boundary_mask(N: int, M: int | None = None):
    M = N if M is None else M
    i, j = np.meshgrid(np.arange(N + 1), np.arange(M + 1), indexing="xy")
    return ((i == 0) | (i == N) | (j == 0) | (j == M)).ravel()


class CollectiveMG:
    def __init__(self, K, Mm, D, alpha, N, isbc, nu=2, omega=0.8, nmin=2, coarse_sweeps=60):
        self.nu, self.omega, self.coarse_sweeps = nu, omega, coarse_sweeps
        self.levels = []
        n = N
        K, Mm, D = K.tocsr(), Mm.tocsr(), D.tocsr()
        mask = isbc.copy()
        while True:
            keep = (~mask).astype(float)
            A = (sp.diags(keep) @ (alpha * K) @ sp.diags(keep) + sp.diags(1.0 - keep)).tocsr()
            B = (sp.diags(keep) @ Mm).tocsr()  # u-rows x psi-cols; psi-rows x u-cols is B^T
            L = dict(A=A, B=B, BT=B.T.tocsr(), D=D, N=n, mask=mask)
            a, b, d = A.diagonal(), B.diagonal(), D.diagonal()
            L["blk"] = (a, b, d, -(a * d) - b * b)
            self.levels.append(L)
            if n <= nmin or n % 2:
                break
            P = interp_matrix(n)
            L["P"] = P
            K, Mm, D = (P.T @ K @ P).tocsr(), (P.T @ Mm @ P).tocsr(), (P.T @ D @ P).tocsr()
            n //= 2
            mask = boundary_mask(n)

    def _apply(self, L, xu, xp):
        return L["A"] @ xu + L["B"] @ xp, L["BT"] @ xu - L["D"] @ xp

    def _smooth(self, L, xu, xp, ru, rp, its):
        a, b, d, det = L["blk"]
        om_u = np.where(L["mask"], 1.0, self.omega)  # BC rows are solved exactly
        for _ in range(its):
            yu, yp = self._apply(L, xu, xp)
            su, s_p = ru - yu, rp - yp
            xu = xu + om_u * (-d * su - b * s_p) / det
            xp = xp + self.omega * (-b * su + a * s_p) / det
        return xu, xp

    def vcycle(self, ru, rp, l=0):
        L = self.levels[l]
        xu, xp = np.zeros_like(ru), np.zeros_like(rp)
        if "P" not in L:
            return self._smooth(L, xu, xp, ru, rp, self.coarse_sweeps)
        xu, xp = self._smooth(L, xu, xp, ru, rp, self.nu)
        yu, yp = self._apply(L, xu, xp)
        keep_c = (~self.levels[l + 1]["mask"]).astype(float)
        cu, cp = self.vcycle(keep_c * (L["P"].T @ (ru - yu)), L["P"].T @ (rp - yp), l + 1)
        xu, xp = xu + L["P"] @ cu, xp + L["P"] @ cp
        return self._smooth(L, xu, xp, ru, rp, self.nu)


def fgmres(A, b, prec, rtol=1e-10, maxit=200):
    """Right-preconditioned flexible GMRES with CGS2; returns (x, iterations, true residual history)."""
    beta = float(np.linalg.norm(b))
    if beta == 0.0:
        return np.zeros_like(b), 0, []
    V, Z = [b / beta], []
    H = np.zeros((maxit + 1, maxit))
    hist = []
    y = None
    for j in range(maxit):
        z = prec(V[j])
        Z.append(z)
        w = A @ z
        for _ in range(2):
            for i in range(j + 1):
                h = V[i] @ w
                H[i, j] += h
                w = w - h * V[i]
        H[j + 1, j] = np.linalg.norm(w)
        V.append(w / H[j + 1, j])
        e = np.zeros(j + 2)
        e[0] = beta
        y = np.linalg.lstsq(H[: j + 2, : j + 1], e, rcond=None)[0]
        rn = float(np.linalg.norm(H[: j + 2, : j + 1] @ y - e))
        hist.append(rn / beta)
        if rn <= rtol * beta:
            break
    x = sum(yi * zi for yi, zi in zip(y, Z))
    return x, len(hist), hist


def make_linear_solve(prob, N, rtol=1e-10, maxit=200, nu=2, omega=0.8, stats=None):
    """linear_solve(J, b) callback for oracle.pg_oracle.newton_solve using FGMRES + collective MG.
    `prob` supplies K, M and the consistent D(psi) blocks; alpha and psi are recovered from J."""
    n = prob.n

    def solve(J, b):
        J = J.tocsr()
        Dm = -J[n:, n:]
        # alpha from an interior diagonal entry of the uu block
        i = int(np.flatnonzero(~prob.isbc)[0])
        alpha = J[i, i] / prob.K[i, i]
        mg = CollectiveMG(prob.K, prob.M, Dm, alpha, N, prob.isbc, nu=nu, omega=omega)

        def prec(r):
            zu, zp = mg.vcycle(r[:n], r[n:])
            return np.concatenate([zu, zp])

        x, its, _ = fgmres(J, b, prec, rtol, maxit)
        if stats is not None:
            stats.append(its)
        return x

    return solve
