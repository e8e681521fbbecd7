"""CPU oracle for the LVPP Newton loop of example 05 (thermoforming quasi-variational inequality, three fields).
TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (DOLFINx / PETSc / MUMPS absent, no reference tests or golden data).

Restated from /root/reference/examples/05_obstacle_type_qvi/thermoforming_dolfinx.py:
* mesh     : create_unit_square(M, M), M = 150 (:24-25); mixed [P1, P1, P1] for (u, T, psi) (:28-33).
* data     : g(s) piecewise linear with knee q = 0.01 (:36-48); beta = 1, f = 25, alpha_0 = 2^-6 (:55-57);
             Phi0 = 1 - 2 max(|x - .5|, |y - .5|), xi = sin(pi x) sin(pi y) as UFL expressions of the coordinates (:58-59).
* residual : :62-67
               R_u   = alpha (grad u, grad v) + (psi, v) - alpha (f, v) - (psi_prev, v)
               R_T   = (grad T, grad q) + beta (T, q) - (g(exp(-psi)), q)
               R_psi = (u, w) + (exp(-psi), w) - (Phi0 + xi T, w)
* Jacobian : the MODIFIED one, derivative(F - eps/alpha (grad psi, grad w), s), eps = 1e-10 (:69-71): J != dF/ds.
* BCs      : u = 0 on the boundary, T and psi free (:73-79).
* Newton   : SNES newtonls with the backtracking line search `bt`, order 2, atol = rtol = 1e-5, stol = 10 eps_machine,
             PETSc default max_it 50 (:100-113).  The line search restates PETSc's SNESLineSearchApply_BT (quadratic
             variant) [upstream, recalled - not verifiable offline].
* outer    : T = 1 initially (:119); solve; stop when ||u - u_prev||_H1 < 1e-9 (:81-83,138-154); s_prev <- s;
             alpha <- min(4 alpha, 2^14) (:156-158); at most 100 steps.
* quadrature: the reference leaves the degree to UFL's estimation; here ONE fixed rule (the shared degree-6 table) serves
             every integral and the spatial coefficients are sampled at its points.

DOF layout: x = [u (nv) | T (nv) | psi (nv)].
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import pg_oracle as O

Q_KNEE = 0.01
EPS_MOD = 1.0e-10
SNES_DIVERGED_LINE_SEARCH = -6


class Thermoforming:
    def __init__(self, coords, cells, bc_vertices, beta=1.0, f=25.0, quadrature="tri_deg6_12"):
        self.coords = np.ascontiguousarray(coords, dtype=np.float64)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.nv, self.nc = len(self.coords), len(self.cells)
        self.ntot = 3 * self.nv
        self.beta, self.f = float(beta), float(f)
        self.bc = np.asarray(bc_vertices, dtype=np.int64)
        self.isbc = np.zeros(self.nv, dtype=bool)
        self.isbc[self.bc] = True
        self.Xq, self.wq = O.load_quadrature(quadrature)
        self.Nq = np.stack([1 - self.Xq[:, 0] - self.Xq[:, 1], self.Xq[:, 0], self.Xq[:, 1]], axis=1)
        x = self.coords[self.cells]
        J = np.stack([x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]], axis=2)
        det = J[:, 0, 0] * J[:, 1, 1] - J[:, 0, 1] * J[:, 1, 0]
        invJ = np.empty_like(J)
        invJ[:, 0, 0], invJ[:, 0, 1] = J[:, 1, 1] / det, -J[:, 0, 1] / det
        invJ[:, 1, 0], invJ[:, 1, 1] = -J[:, 1, 0] / det, J[:, 0, 0] / det
        gref = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])
        self.G = np.einsum("ak,ckd->cad", gref, invJ)
        self.wdet = np.abs(det)[:, None] * self.wq[None]
        xq = np.einsum("qa,cad->cqd", self.Nq, x)
        self.phi0_q = 1.0 - 2.0 * np.maximum(np.abs(xq[..., 0] - 0.5), np.abs(xq[..., 1] - 0.5))
        self.xi_q = np.sin(np.pi * xq[..., 0]) * np.sin(np.pi * xq[..., 1])
        r = np.repeat(self.cells, 3, axis=1).ravel()
        c = np.tile(self.cells, (1, 3)).ravel()
        self._r, self._c = r, c
        mk = lambda Ae: sp.coo_matrix((Ae.ravel(), (r, c)), shape=(self.nv, self.nv)).tocsr()  # noqa: E731
        self._mk = mk
        self.K = mk(0.5 * np.abs(det)[:, None, None] * np.einsum("cad,cbd->cab", self.G, self.G))
        self.M = mk(np.einsum("cq,qa,qb->cab", self.wdet, self.Nq, self.Nq))
        self.Mxi = mk(np.einsum("cq,qa,qb->cab", self.wdet * self.xi_q, self.Nq, self.Nq))
        self.m = np.bincount(self.cells.ravel(), weights=(self.wdet @ self.Nq).ravel(), minlength=self.nv)
        self.b_phi0 = np.bincount(self.cells.ravel(), weights=((self.wdet * self.phi0_q) @ self.Nq).ravel(), minlength=self.nv)

    def split(self, x):
        n = self.nv
        return x[:n], x[n:2 * n], x[2 * n:]

    def _latent(self, psi, with_matrix):
        pq = psi[self.cells] @ self.Nq.T
        with np.errstate(over="ignore", under="ignore", invalid="ignore"):
            s = np.exp(-pq)
            gval = np.where(s < Q_KNEE, 1.0 - s / Q_KNEE, 0.0)  # s > 0 always: the branch s < 0 of :42-47 is never taken
            cb = self.cells.ravel()
            b_g = np.bincount(cb, weights=((self.wdet * gval) @ self.Nq).ravel(), minlength=self.nv)
            b_e = np.bincount(cb, weights=((self.wdet * s) @ self.Nq).ravel(), minlength=self.nv)  # inf/nan if a trial overshoots
        if not with_matrix:
            return b_g, b_e, None, None
        D = self._mk(np.einsum("cq,qa,qb->cab", self.wdet * s, self.Nq, self.Nq))
        # d/dpsi of -(g(exp(-psi)), q) = -(g'(s) (-s) dpsi, q) = (g'(s) s dpsi, q), g' = -1/q on (0, q)
        C = self._mk(np.einsum("cq,qa,qb->cab", self.wdet * np.where(s < Q_KNEE, -s / Q_KNEE, 0.0), self.Nq, self.Nq))
        return b_g, b_e, D, C

    def residual(self, x, xk, alpha):
        u, T, psi = self.split(x)
        psik = xk[2 * self.nv:]
        ut = u.copy()
        ut[self.bc] = 0.0
        b_g, b_e, _, _ = self._latent(psi, False)
        Fu = alpha * (self.K @ ut) + self.M @ (psi - psik) - alpha * self.f * self.m
        FT = self.K @ T + self.beta * (self.M @ T) - b_g
        Fp = self.M @ ut + b_e - self.b_phi0 - self.Mxi @ T
        Fu[self.bc] = u[self.bc]
        return np.concatenate([Fu, FT, Fp])

    def jacobian(self, x, alpha):
        _, _, D, C = self._latent(x[2 * self.nv:], True)
        free = sp.diags((~self.isbc).astype(float))
        A = free @ (alpha * self.K) @ free + sp.diags(self.isbc.astype(float))
        Z = sp.csr_matrix((self.nv, self.nv))
        return sp.bmat([[A, Z, free @ self.M],
                        [Z, self.K + self.beta * self.M, C],
                        [self.M @ free, -self.Mxi, -D - (EPS_MOD / alpha) * self.K]], format="csr")

    def h1_increment(self, x, xk):
        d = x[: self.nv] - xk[: self.nv]
        return float(np.sqrt(max(d @ (self.M @ d) + d @ (self.K @ d), 0.0)))


def newton_bt(prob, x0, xk, alpha, rtol=1e-5, atol=1e-5, stol=10 * np.finfo(float).eps, max_it=50, divtol=1e4,
              linear_solve=None, log=None):
    """SNES newtonls with SNESLineSearchApply_BT, order 2 (quadratic), PETSc defaults alpha_ls = 1e-4, maxstep = 1e8,
    steptol = 1e-12, at most 40 backtracking steps [upstream, recalled].  Returns (x, reason, its)."""
    x = x0.copy()
    F = prob.residual(x, xk, alpha)
    fnorm = float(np.linalg.norm(F))
    fnorm0 = fnorm
    if log is not None:
        log.append(fnorm)
    if not np.isfinite(fnorm):
        return x, O.SNES_DIVERGED_FNORM_NAN, 0
    if fnorm < atol:
        return x, O.SNES_CONVERGED_FNORM_ABS, 0
    ttol = fnorm * rtol
    for it in range(1, max_it + 1):
        J = prob.jacobian(x, alpha)
        y = spla.splu(J.tocsc()).solve(F) if linear_solve is None else linear_solve(J, F)  # x_new = x - lambda y
        if not np.all(np.isfinite(y)):
            return x, O.SNES_DIVERGED_LINEAR_SOLVE, it
        # ---- SNESLineSearchApply_BT ----
        ynorm = float(np.linalg.norm(y))
        if ynorm > 1e8:
            y = y * (1e8 / ynorm)
            ynorm = 1e8
        initslope = float(F @ (J @ y))
        if initslope > 0.0:
            initslope = -initslope
        if initslope == 0.0:
            initslope = -1.0
        rellength = float(np.max(np.abs(y) / np.maximum(np.abs(x), 1.0)))
        minlambda = 1e-12 / rellength
        f = fnorm * fnorm
        lam = 1.0
        w = x - lam * y
        G = prob.residual(w, xk, alpha)
        with np.errstate(over="ignore", invalid="ignore"):
            g = float(G @ G)
        ok = True
        # a non-finite trial residual (exp(-psi) overflows when a full step throws psi far negative) counts as "no
        # sufficient decrease" and shrinks lambda by the largest allowed factor (PETSc guards its acceptance test with
        # !PetscIsInfOrNanReal(g); the shrink factor for that case is this restatement's choice)
        def shrink(lam, g, with_lam):
            if not np.isfinite(g):
                return 0.1 * lam
            lamtemp = -initslope / (g - f - 2.0 * (lam if with_lam else 1.0) * initslope)
            lamtemp = min(lamtemp, 0.5 * lam)
            return 0.1 * lam if lamtemp <= 0.1 * lam else lamtemp

        if not (np.isfinite(g) and 0.5 * g <= 0.5 * f + lam * 1e-4 * initslope):
            lam = shrink(lam, g, True)
            w = x - lam * y
            G = prob.residual(w, xk, alpha)
            with np.errstate(over="ignore", invalid="ignore"):
                g = float(G @ G)
            if not (np.isfinite(g) and 0.5 * g < 0.5 * f + lam * 1e-4 * initslope):
                count = 0
                while True:
                    if lam <= minlambda:
                        ok = False
                        break
                    lam = shrink(lam, g, False)
                    w = x - lam * y
                    G = prob.residual(w, xk, alpha)
                    with np.errstate(over="ignore", invalid="ignore"):
                        g = float(G @ G)
                    if np.isfinite(g) and 0.5 * g < 0.5 * f + lam * 1e-4 * initslope:
                        break
                    count += 1
                    if count > 40:
                        ok = False
                        break
        if not ok:
            return x, SNES_DIVERGED_LINE_SEARCH, it
        x, F = w, G
        fnorm = float(np.sqrt(g))
        if log is not None:
            log.append(fnorm)
        if fnorm < atol:
            return x, O.SNES_CONVERGED_FNORM_ABS, it
        if fnorm <= ttol:
            return x, O.SNES_CONVERGED_FNORM_RELATIVE, it
        if lam * ynorm < stol * float(np.linalg.norm(x)):
            return x, O.SNES_CONVERGED_SNORM_RELATIVE, it
        if fnorm > divtol * fnorm0:
            return x, O.SNES_DIVERGED_DTOL, it
    return x, O.SNES_DIVERGED_MAX_IT, max_it


def solve_problem(prob: Thermoforming, alpha_0=2.0**-6, alpha_max=2.0**14, termination_tol=1e-9, max_lvpp_iterations=100,
                  linear_solve=None, verbose=False, iterates=None):
    """Mirror of the script's LVPP loop (:117-158). Returns (x, num_iterations list, H1 increments)."""
    x = np.zeros(prob.ntot)
    x[prob.nv:2 * prob.nv] = 1.0  # :119
    xk = np.zeros(prob.ntot)      # s_prev starts at zero (a fresh Function)
    alpha = alpha_0
    its_all, diffs = [], []
    for i in range(1, max_lvpp_iterations + 1):
        x, reason, its = newton_bt(prob, x, xk, alpha, linear_solve=linear_solve)
        if reason <= 0:
            raise RuntimeError(f"Solver did not converge with {reason} at LVPP iteration {i}")  # :129-130
        d = prob.h1_increment(x, xk)
        its_all.append(its)
        diffs.append(d)
        if iterates is not None:
            iterates.append(x.copy())
        if verbose:
            print(f"LVPP iteration {i} alpha {alpha:g} reason {reason} Newton {its} ||u-u_prev||={d:.3e}")
        if d < termination_tol:
            break
        xk = x.copy()
        alpha = min(alpha_max, 4 * alpha)
    return x, its_all, diffs
