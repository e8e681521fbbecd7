"""CPU oracle for the LVPP loop of example 08 (intersecting constraints: an obstacle AND a gradient bound on one primal field).
TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (DOLFINx / PETSc / MUMPS absent, no reference tests or golden data for this example).

Restated from /root/reference/examples/08_intersecting_constraints/intersecting_constraints_dolfinx.py:
* mesh     : create_unit_interval(1001) (:13); mixed [P1, P1, (P1)^1] for (u, psi0, psi) (:15-23).
* data     : c = 0 (:30); phi0 = the smooth bump on (0.2, 0.8) normalised to 1 at x = 0.5 (:39-42; it overrides the box of :37);
             phi = phic for x <= 0.2 and x > 0.8, 100 between (:44-45); phic runs through 3, 2, 1, 0.5, 0.1, 0.01 (:114).
* residual : :47-58, alpha dE/dz + the latent rows of example 01 (exp) for psi0 and of example 06 (Hellinger) for psi
               R_u    = alpha [(u', v') + (c, v)] + (psi0 - psi0_iter, v) + (psi - psi_iter, v')
               R_psi0 = (u, w0) - (exp(psi0), w0) - (phi0, w0)
               R_psi  = (u', w) - (phi psi / sqrt(1 + psi^2), w)
* Jacobian : NonlinearProblem differentiates F (:75-77).
* BCs      : u = 0 at both ends (:60-63).
* Newton   : SNES newtonls, line search `l2` with maxlambda 1, atol = rtol = 1e-6, stol = 1e-14, LU (:66-79).  The line search
             restates PETSc's SNESLineSearchApply_L2 (one secant step on |F|^2 through lambda = 0, 1/2, 1; max_it 1, steptol 1e-12)
             [upstream, recalled - not verifiable offline].
* outer    : :112-175 - per phic: alpha = 1, z_iter = z; solve; a solve that fails OR converges without an iteration halves alpha
             and restores z (z_prev = 0 on the first proximal step of a phic, z_iter later), at most 50 failures; stop when
             ||u - u_iter||_L2 < 1e-4; alpha doubles after <= 4 Newton steps, halves after >= 10; z_iter <- z.
* quadrature: the reference leaves the degree to UFL's estimation - 6 for the sum of the residual's integrands (the Hellinger term:
             1 + (2 + 2) + 1), i.e. Basix's 4-point Gauss-Jacobi (= Gauss-Legendre) rule on the interval; the Jacobian here is the
             exact derivative of THAT discrete residual (UFL would estimate a higher degree for the differentiated form).

DOF layout: z = [u (nv) | psi0 (nv) | psi (nv)].
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

PHICS = (3, 2, 1, 0.5, 0.1, 0.01)  # :114
NFAIL_MAX = 50  # :113
SNES_DIVERGED_LINE_SEARCH = -6


def gauss_legendre_unit(degree):
    """Basix's default rule on the interval for `degree`: Gauss-Jacobi with m = (degree + 2) // 2 points, on (0, 1), weights sum 1."""
    m = (int(degree) + 2) // 2
    t, w = np.polynomial.legendre.leggauss(m)
    return 0.5 * (t + 1.0), 0.5 * w


def phi0_bump(x, l=0.2, r=0.8):
    """:39-42"""
    x = np.asarray(x, dtype=np.float64)
    inside = (x > l) & (x < r)
    xs = np.where(inside, x, 0.5)
    bump = np.exp(-1.0 / (10.0 * (xs - l) * (r - xs))) / np.exp(-1.0 / (10.0 * (0.5 - l) * (r - 0.5)))
    return np.where(inside, bump, 0.0)


def phi_bound(x, phic):
    """:44-45"""
    x = np.asarray(x, dtype=np.float64)
    return np.where(x <= 0.2, float(phic), np.where(x > 0.8, float(phic), 100.0))


class Intersecting:
    def __init__(self, n=1001, quadrature_degree=6, c=0.0, x=None, phi0=phi0_bump, phi=phi_bound):
        self.x = np.linspace(0.0, 1.0, n + 1) if x is None else np.ascontiguousarray(x, dtype=np.float64)
        self.nv = len(self.x)
        self.nc = self.nv - 1
        self.ntot = 3 * self.nv
        self.c = float(c)
        self.h = np.diff(self.x)
        self.tq, self.wq = gauss_legendre_unit(quadrature_degree)
        self.Nq = np.stack([1.0 - self.tq, self.tq], axis=1)  # [q][a]
        self.xq = self.x[:-1, None] + self.h[:, None] * self.tq[None]
        self.wdet = self.h[:, None] * self.wq[None]
        self.dN = np.stack([-1.0 / self.h, 1.0 / self.h], axis=1)  # [c][a]
        self.cells = np.stack([np.arange(self.nc), np.arange(1, self.nv)], axis=1)
        self.bc = np.array([0, self.nv - 1])
        self.isbc = np.zeros(self.nv, dtype=bool)
        self.isbc[self.bc] = True
        self._phi0, self._phi = phi0, phi
        self.phi0_q = phi0(self.xq)
        self.set_phic(100.0)  # :43
        r = np.repeat(self.cells, 2, axis=1).ravel()
        cc = np.tile(self.cells, (1, 2)).ravel()
        self._mk = lambda Ae: sp.coo_matrix((Ae.ravel(), (r, cc)), shape=(self.nv, self.nv)).tocsr()  # noqa: E731
        self.K = self._mk(self.h[:, None, None] * self.dN[:, :, None] * self.dN[:, None, :])
        self.M = self._mk(np.einsum("cq,qa,qb->cab", self.wdet, self.Nq, self.Nq))
        self.G = self._mk(np.einsum("cq,qa,cb->cab", self.wdet, self.Nq, self.dN))  # G[i, j] = (N_j', N_i): rows psi, columns u
        self.m = np.bincount(self.cells.ravel(), weights=(self.wdet @ self.Nq).ravel(), minlength=self.nv)

    def set_phic(self, phic):
        self.phic = float(phic)
        self.phi_q = self._phi(self.xq, phic)

    def split(self, z):
        n = self.nv
        return z[:n], z[n:2 * n], z[2 * n:]

    def _scatter(self, cq):
        """sum_q cq[c, q] N_a(q) -> vertex vector"""
        return np.bincount(self.cells.ravel(), weights=(cq @ self.Nq).ravel(), minlength=self.nv)

    def residual(self, z, z_iter, alpha):
        u, p0, p = (v.copy() for v in self.split(z))
        _, p0k, pk = self.split(z_iter)
        ubc = u[self.bc].copy()
        u[self.bc] = 0.0  # the residual is assembled with the boundary values in place
        p0q, pq = p0[self.cells] @ self.Nq.T, p[self.cells] @ self.Nq.T
        with np.errstate(over="ignore", invalid="ignore"):
            e0 = np.exp(p0q)
            hel = self.phi_q * pq / np.sqrt(1.0 + pq * pq)
            Ru = alpha * (self.K @ u + self.c * self.m) + self.M @ (p0 - p0k) + self.G.T @ (p - pk)
            Rp0 = self.M @ u - self._scatter(self.wdet * (e0 + self.phi0_q))
            Rp = self.G @ u - self._scatter(self.wdet * hel)
        Ru[self.bc] = ubc
        return np.concatenate([Ru, Rp0, Rp])

    def jacobian(self, z, alpha):
        _, p0, p = self.split(z)
        p0q, pq = p0[self.cells] @ self.Nq.T, p[self.cells] @ self.Nq.T
        with np.errstate(over="ignore", invalid="ignore"):
            D0 = self._mk(np.einsum("cq,qa,qb->cab", self.wdet * np.exp(p0q), self.Nq, self.Nq))
            D = self._mk(np.einsum("cq,qa,qb->cab", self.wdet * self.phi_q * (1.0 + pq * pq) ** -1.5, self.Nq, self.Nq))
        keep = sp.diags((~self.isbc).astype(float))
        Kb = keep @ (alpha * self.K) @ keep + sp.diags(self.isbc.astype(float))
        J = sp.bmat([[Kb, keep @ self.M, keep @ self.G.T], [self.M @ keep, -D0, None], [self.G @ keep, None, -D]], format="csr")
        return J

    def l2_increment(self, z, z_iter):
        d = self.split(z)[0] - self.split(z_iter)[0]
        return float(np.sqrt(max(d @ (self.M @ d), 0.0)))

    # ------------------------------------------------------------------------------------------------------------------
    def newton_l2(self, z, z_iter, alpha, atol=1e-6, rtol=1e-6, stol=1e-14, max_it=50, maxlambda=1.0, steptol=1e-12, ls_max_it=1,
                  divtol=1e4, monitor=False):
        """-> (z_new or z, reason, its)"""
        z = z.copy()
        F = self.residual(z, z_iter, alpha)
        fnorm = float(np.linalg.norm(F))
        fnorm0 = fnorm
        if monitor:
            print(f"  0 SNES Function norm {fnorm:.12e}")
        if not np.isfinite(fnorm):
            return z, -4, 0
        if fnorm < atol:
            return z, 2, 0
        ttol = fnorm * rtol
        for it in range(1, max_it + 1):
            J = self.jacobian(z, alpha)
            try:
                with np.errstate(all="ignore"):
                    y = spla.splu(J.tocsc()).solve(F)
            except RuntimeError:
                return z, -3, it - 1
            if not np.all(np.isfinite(y)):
                return z, -3, it - 1
            lam, lam_old, maxl = 1.0, 0.0, maxlambda
            fn_old = fnorm * fnorm
            lam_mid = 0.5 * (lam + lam_old)
            failed = False
            for _ in range(ls_max_it):
                while True:
                    fm = np.linalg.norm(self.residual(z - lam_mid * y, z_iter, alpha)) ** 2
                    fe = np.linalg.norm(self.residual(z - lam * y, z_iter, alpha)) ** 2
                    if np.isfinite(fe):
                        break
                    if lam <= steptol:
                        failed = True
                        break
                    maxl = 0.95 * lam
                    lam = 0.5 * (lam + lam_old)
                    lam_mid = 0.5 * (lam + lam_old)
                if failed:
                    break
                dl = lam - lam_old
                d1 = (3.0 * fe - 4.0 * fm + fn_old) / dl
                d1_old = (-3.0 * fn_old + 4.0 * fm - fe) / dl
                d2 = (d1 - d1_old) / dl
                if d2 > 0.0:
                    upd = lam - d1 / d2
                elif d2 < 0.0:
                    upd = lam + d1 / d2
                else:
                    break
                if upd < steptol:
                    upd = 0.5 * (lam + lam_old)
                if not np.isfinite(upd) or upd > maxl:
                    break
                lam_old, lam, fn_old = lam, upd, fe
                lam_mid = 0.5 * (lam + lam_old)
            if failed:
                return z, SNES_DIVERGED_LINE_SEARCH, it
            z = z - lam * y
            F = self.residual(z, z_iter, alpha)
            fnorm = float(np.linalg.norm(F))
            if monitor:
                print(f"      line search: lambda {lam:.6e}\n  {it} SNES Function norm {fnorm:.12e}")
            if not np.isfinite(fnorm):
                return z, -4, it
            if fnorm < atol:
                return z, 2, it
            if fnorm <= ttol:
                return z, 3, it
            if np.linalg.norm(y) < stol * np.linalg.norm(z):
                return z, 4, it
            if fnorm > divtol * fnorm0:
                return z, -9, it
        return z, -5, max_it


def solve_problem(prob: Intersecting, phis=PHICS, tol=1e-4, nfail_max=NFAIL_MAX, verbose=False, z0=None):
    """:112-175.  -> (z, num_lvpp_iterations per phic, num_newton_iterations per phic, log of (phic, k, alpha, its, reason, increment))"""
    z = np.zeros(prob.ntot) if z0 is None else np.array(z0, dtype=np.float64)
    z_prev = np.zeros(prob.ntot)  # never updated by the script (:25)
    n_newton, n_lvpp, log = [0] * len(phis), [0] * len(phis), []
    for i, phic in enumerate(phis):
        prob.set_phic(phic)
        alpha, k, r, nfail = 1.0, 1, 2.0, 0
        z_iter = z.copy()
        while nfail <= nfail_max:
            znew, reason, its = prob.newton_l2(z, z_iter, alpha)
            n_newton[i] += its
            if (its == 0 and reason > 0) or reason < 0:
                nfail += 1
                log.append((phic, k, alpha, its, reason, None))
                alpha /= 2
                z = (z_prev if k == 1 else z_iter).copy()
                if nfail >= nfail_max:
                    break
                continue
            z = znew
            nrm = prob.l2_increment(z, z_iter)
            log.append((phic, k, alpha, its, reason, nrm))
            if verbose:
                print(f"Solved k={k} phi={phic} alpha={alpha} its={its} ||u_k - u_k-1|| = {nrm}")
            n_lvpp[i] += 1
            if nrm < tol:
                break
            if its <= 4:
                alpha *= r
            elif its >= 10:
                alpha /= r
            z_iter = z.copy()
            k += 1
    return z, n_lvpp, n_newton, log
