"""Would a TRUNCATED orthogonalisation (FGMRES orthogonalising against the last k basis vectors only: incomplete orthogonalisation,
DQGMRES-style) keep the Krylov counts of the headline solver?  Gram-Schmidt is ~24 % of a 2048^2 solve.  numpy twin
(oracle/krylov_proto.py: V(6,6), omega 0.75), every Newton system of a settings-B run, convergence judged by the TRUE residual.
    python tools/trunc_study.py 128"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import krylov_proto as KP  # noqa: E402
from oracle import pg_oracle as O  # noqa: E402


def fgmres_trunc(A, b, prec, k, rtol=1e-10, maxit=200):
    beta = float(np.linalg.norm(b))
    V, Z = [b / beta], []
    H = np.zeros((maxit + 1, maxit))
    for j in range(maxit):
        z = prec(V[j])
        Z.append(z)
        w = A @ z
        lo = 0 if k is None else max(0, j + 1 - k)
        for _ in range(2):
            for i in range(lo, j + 1):
                h = V[i] @ w
                H[i, j] += h
                w = w - h * V[i]
        H[j + 1, j] = np.linalg.norm(w)
        V.append(w / H[j + 1, j])
        e = np.zeros(j + 2)
        e[0] = beta
        y = np.linalg.lstsq(H[: j + 2, : j + 1], e, rcond=None)[0]
        x = sum(yi * zi for yi, zi in zip(y, Z))
        if np.linalg.norm(b - A @ x) <= rtol * beta:
            return x, j + 1
    return x, maxit


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    coords, cells = O.create_rectangle(N, N)
    prob = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    n = prob.n
    for k in (None, 5, 3, 2):
        stats, t = [], time.time()

        def solve(J, b):
            J = J.tocsr()
            i = int(np.flatnonzero(~prob.isbc)[0])
            mg = KP.CollectiveMG(prob.K, prob.M, -J[n:, n:], J[i, i] / prob.K[i, i], N, prob.isbc, nu=6, omega=0.75)
            x, its = fgmres_trunc(J, b, lambda r: np.concatenate(mg.vcycle(r[:n], r[n:])), k)
            stats.append(its)
            return x

        x, h = O.solve_problem(prob, 500, "double_exponential", 1e2, 1e-4, linear_solve=solve)
        print(f"window {k}: Newton {h['Newton steps']}  Krylov per solve {stats}  total {sum(stats)}  ({time.time() - t:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
