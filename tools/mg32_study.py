#!/usr/bin/env python3
"""Does a SINGLE-PRECISION V-cycle cost Krylov iterations?  Numpy prototype of the Newton linear solver (oracle/krylov_proto.py):
one full settings-B LVPP run with the V(6,6) collective-Jacobi cycle evaluated (a) in double, (b) with every level's stencils, the
vectors and the arithmetic inside the cycle in float32 (FGMRES, the operator apply and the residuals stay double - the cycle is a
preconditioner inside a FLEXIBLE Krylov method), (c) float32 on the levels with more than `nmin32` cells per side only.
Results: DESIGN.md section 5b.     python tools/mg32_study.py 256"""
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from oracle import krylov_proto as KP  # noqa: E402
from oracle import pg_oracle as O  # noqa: E402


class MG32(KP.CollectiveMG):
    def __init__(self, *a, nmin32=0, **k):
        super().__init__(*a, **k)
        self.nmin32 = nmin32
        for L in self.levels:
            if L["N"] > nmin32:
                for nm in ("A", "B", "BT", "D"):
                    L[nm] = L[nm].astype(np.float32)
                if "P" in L:
                    L["P32"] = L["P"].astype(np.float32)
                a_, b_, d_, _ = L["blk"]
                a_, b_, d_ = a_.astype(np.float32), b_.astype(np.float32), d_.astype(np.float32)
                L["blk"] = (a_, b_, d_, -(a_ * d_) - b_ * b_)
                L["f32"] = True

    def _smooth(self, L, xu, xp, ru, rp, its):
        if not L.get("f32"):
            return super()._smooth(L, xu, xp, ru, rp, its)
        a, b, d, det = L["blk"]
        om = np.float32(self.omega)
        om_u = np.where(L["mask"], np.float32(1.0), om).astype(np.float32)
        xu, xp, ru, rp = (v.astype(np.float32) for v in (xu, xp, ru, rp))
        rc = (np.float32(1.0) / det).astype(np.float32)
        for _ in range(its):
            yu, yp = self._apply(L, xu, xp)
            su, s_p = ru - yu, rp - yp
            xu = xu + om_u * ((-d * su - b * s_p) * rc)
            xp = xp + om * ((-b * su + a * s_p) * rc)
        assert xu.dtype == np.float32 and xp.dtype == np.float32
        return xu, xp

    def vcycle(self, ru, rp, l=0):
        L = self.levels[l]
        if not L.get("f32"):
            zu, zp = super().vcycle(ru.astype(np.float64), rp.astype(np.float64), l)
            return zu, zp
        ru, rp = ru.astype(np.float32), rp.astype(np.float32)
        xu, xp = np.zeros_like(ru), np.zeros_like(rp)
        if "P" not in L:
            return self._smooth(L, xu, xp, ru, rp, self.coarse_sweeps)
        xu, xp = self._smooth(L, xu, xp, ru, rp, self.nu)
        yu, yp = self._apply(L, xu, xp)
        keep_c = (~self.levels[l + 1]["mask"]).astype(np.float32)
        cu, cp = self.vcycle(keep_c * (L["P32"].T @ (ru - yu)), L["P32"].T @ (rp - yp), l + 1)
        xu = xu + L["P32"] @ cu.astype(np.float32)
        xp = xp + L["P32"] @ cp.astype(np.float32)
        return self._smooth(L, xu, xp, ru, rp, self.nu)


def make(prob, N, nmin32, stats, omega):
    n = prob.n

    def solve(J, b):
        J = J.tocsr()
        i = int(np.flatnonzero(~prob.isbc)[0])
        kw = dict(nu=6, omega=omega)
        if nmin32 is None:
            mg = KP.CollectiveMG(prob.K, prob.M, -J[n:, n:], J[i, i] / prob.K[i, i], N, prob.isbc, **kw)
        else:
            mg = MG32(prob.K, prob.M, -J[n:, n:], J[i, i] / prob.K[i, i], N, prob.isbc, nmin32=nmin32, **kw)
        x, its, _ = KP.fgmres(J, b, lambda r: np.concatenate([v.astype(np.float64) for v in mg.vcycle(r[:n], r[n:])]), 1e-10, 200)
        stats.append(its)
        return x

    return solve


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    coords, cells = O.create_rectangle(N, N)
    prob = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    ref = None
    for name, nmin32 in (("double V(6,6)", None), ("float32 on every level", 0), ("float32 above 32 cells per side", 32)):
        stats, t = [], time.time()
        x, h = O.solve_problem(prob, 500, "double_exponential", 1e2, 1e-4, linear_solve=make(prob, N, nmin32, stats, 0.75))
        if ref is None:
            ref = x
        du = np.linalg.norm(x[: prob.n] - ref[: prob.n]) / np.linalg.norm(ref[: prob.n])
        print(f"{name:36s} Newton {h['Newton steps']}  Krylov per solve {stats}  total {sum(stats)}  |u - u_double|/|u| {du:.1e}"
              f"  ({time.time() - t:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
