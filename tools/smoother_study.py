#!/usr/bin/env python3
"""Smoother study on the numpy prototype of the Newton linear solver (oracle/krylov_proto.py): Krylov iterations of one full
settings-B LVPP run when the damped collective Jacobi sweeps of the V-cycle are replaced by
  * a collective 3-colour Gauss-Seidel (colour (i + j) mod 3: the 7-point stencils of the right-diagonal mesh never couple two
    vertices of one colour) - VERDICT r02 item 5 asked whether V(3,3) GS matches V(6,6) Jacobi;
  * Jacobi sweeps with Chebyshev weights (omega_k = 1 / k-th Chebyshev node of [hi / ratio, hi]; no extra kernel cost).
Results at 128^2 / 256^2 and what they mean for the HIP kernels: DESIGN.md section 5b.     python tools/smoother_study.py 128"""
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from oracle import krylov_proto as KP  # noqa: E402
from oracle import pg_oracle as O  # noqa: E402


def cheb_omegas(K, lo, hi):
    k = np.arange(1, K + 1)
    return tuple(1.0 / (0.5 * (lo + hi) + 0.5 * (hi - lo) * np.cos(np.pi * (2 * k - 1) / (2 * K))))


class StudyMG(KP.CollectiveMG):
    """kind 'jac': Jacobi sweeps with the weight sequence `omegas` (cycled); kind 'gs': 3-colour collective Gauss-Seidel sweeps.
    Levels with <= 33^2 vertices keep the product's setting (6 Jacobi sweeps, omega 0.8: k_mg_tail2 is not part of the study)."""

    def __init__(self, *a, kind="jac", omegas=(0.8,), **k):
        super().__init__(*a, **k)
        self.kind, self.omegas = kind, omegas
        for L in self.levels:
            n = L["N"]
            i, j = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="xy")
            col = ((i + j) % 3).ravel()
            L["cidx"] = [np.flatnonzero(col == c) for c in range(3)]
            for nm in ("A", "B", "BT", "D"):
                L[nm + "c"] = [L[nm][ix] for ix in L["cidx"]]

    def _smooth(self, L, xu, xp, ru, rp, its):
        if "P" not in L:
            return super()._smooth(L, xu, xp, ru, rp, its)
        a, b, d, det = L["blk"]
        if L["N"] <= 32:
            its, kind, omegas = 6, "jac", (0.8,)
        else:
            kind, omegas = self.kind, self.omegas
        xu, xp = xu.copy(), xp.copy()
        for s in range(its):
            om = omegas[s % len(omegas)]
            if kind == "jac":
                yu, yp = self._apply(L, xu, xp)
                su, s_p = ru - yu, rp - yp
                xu = xu + np.where(L["mask"], 1.0, om) * (-d * su - b * s_p) / det
                xp = xp + om * (-b * su + a * s_p) / det
                continue
            for c in range(3):
                ix = L["cidx"][c]
                su = ru[ix] - (L["Ac"][c] @ xu + L["Bc"][c] @ xp)
                s_p = rp[ix] - (L["BTc"][c] @ xu - L["Dc"][c] @ xp)
                xu[ix] += np.where(L["mask"][ix], 1.0, om) * (-d[ix] * su - b[ix] * s_p) / det[ix]
                xp[ix] += om * (-b[ix] * su + a[ix] * s_p) / det[ix]
        return xu, xp


def make(prob, N, kind, nu, omegas, stats):
    n = prob.n

    def solve(J, b):
        J = J.tocsr()
        i = int(np.flatnonzero(~prob.isbc)[0])
        mg = StudyMG(prob.K, prob.M, -J[n:, n:], J[i, i] / prob.K[i, i], N, prob.isbc, nu=nu, kind=kind, omegas=omegas)
        x, its, _ = KP.fgmres(J, b, lambda r: np.concatenate(mg.vcycle(r[:n], r[n:])), 1e-10, 200)
        stats.append(its)
        return x

    return solve


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    coords, cells = O.create_rectangle(N, N)
    prob = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    cases = [("Jacobi V(6,6) omega 0.80", "jac", 6, (0.8,)), ("Jacobi V(6,6) omega 0.75", "jac", 6, (0.75,)), ("Jacobi V(3,3) omega 0.80", "jac", 3, (0.8,)),
             ("3-colour GS V(1,1)", "gs", 1, (1.0,)), ("3-colour GS V(2,2)", "gs", 2, (1.0,)), ("3-colour GS V(2,2) omega 1.15", "gs", 2, (1.15,)),
             ("3-colour GS V(3,3)", "gs", 3, (1.0,))]
    for hi, r in ((2.0, 8), (2.2, 8), (2.67, 8), (3.0, 8)):
        cases.append((f"Chebyshev-6 Jacobi V(6,6) on [{hi}/{r}, {hi}]", "jac", 6, cheb_omegas(6, hi / r, hi)))
    for hi, r in ((2.0, 4), (2.2, 4), (2.67, 4)):
        cases.append((f"Chebyshev-3 Jacobi V(3,3) on [{hi}/{r}, {hi}]", "jac", 3, cheb_omegas(3, hi / r, hi)))
    for name, kind, nu, om in cases:
        stats, t = [], time.time()
        try:
            _, h = O.solve_problem(prob, 500, "double_exponential", 1e2, 1e-4, linear_solve=make(prob, N, kind, nu, om, stats))
            print(f"{name:48s} Newton {sum(h['Newton steps'])}  Krylov total {sum(stats)}  max {max(stats)}  ({time.time() - t:.0f} s)", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"{name:48s} FAILED {e!r}"[:140], flush=True)


if __name__ == "__main__":
    main()
