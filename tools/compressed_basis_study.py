"""Would an fp32-STORED Krylov basis (compressed-basis GMRES: V_j kept in float, dot products and updates accumulated in double, the
operator apply, the Hessenberg and the TRUE-residual stop in double) keep the Krylov counts of the headline solver?  VERDICT r04
item 3.  numpy twin (oracle/krylov_proto.py: V(6,6), omega 0.75), every Newton system of a settings-B run.  The twin follows the
library's loop: a cycle ends when the ARNOLDI estimate reaches rtol * |b|, the loop head then forms the true residual and restarts
from it if that is still above the target.
    python tools/compressed_basis_study.py 128
Output per variant: Krylov iterations per solve, restarts, and the basis-vector reads of the Gram-Schmidt passes in units of one
fp64 vector (the bytes the compression is meant to save)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import krylov_proto as KP  # noqa: E402
from oracle import pg_oracle as O  # noqa: E402


def fgmres(A, b, prec, store, rtol=1e-10, m=30, maxit=200):
    """store: dtype the basis vectors are kept in.  Returns x, iterations, restarts, vector reads (fp64 units)."""
    bn = float(np.linalg.norm(b))
    x = np.zeros_like(b)
    its = restarts = 0
    reads = 0.0
    unit = np.dtype(store).itemsize / 8.0
    r = b.copy()
    while True:
        beta = float(np.linalg.norm(r))
        if beta <= rtol * bn or its >= maxit:
            return x, its, restarts, reads
        V = [(r / beta).astype(store)]
        Z = []
        H = np.zeros((m + 1, m))
        g = np.zeros(m + 1)
        g[0] = beta
        cs, sn = np.zeros(m), np.zeros(m)
        j = 0
        for j in range(m):
            z = prec(V[j].astype(np.float64))
            Z.append(z)
            w = A @ z
            for _ in range(2):  # CGS2 (the library skips the second pass when the first removed little: same arithmetic when taken)
                Vm = np.stack([v.astype(np.float64) for v in V], axis=1)
                h = Vm.T @ w
                w = w - Vm @ h
                H[: j + 1, j] += h
                reads += 2 * (j + 1) * unit
            hn = float(np.linalg.norm(w))
            H[j + 1, j] = hn
            V.append((w / hn).astype(store))
            for i in range(j):
                t = cs[i] * H[i, j] + sn[i] * H[i + 1, j]
                H[i + 1, j] = -sn[i] * H[i, j] + cs[i] * H[i + 1, j]
                H[i, j] = t
            rr = np.hypot(H[j, j], H[j + 1, j])
            cs[j], sn[j] = H[j, j] / rr, H[j + 1, j] / rr
            H[j, j], H[j + 1, j] = rr, 0.0
            g[j + 1] = -sn[j] * g[j]
            g[j] = cs[j] * g[j]
            its += 1
            if abs(g[j + 1]) <= rtol * bn or its >= maxit:
                break
        k = j + 1
        y = np.linalg.solve(np.triu(H[:k, :k]), g[:k])
        x = x + np.stack(Z, axis=1) @ y
        r = b - A @ x
        restarts += 1


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    coords, cells = O.create_rectangle(N, N)
    prob = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    n = prob.n
    for store in (np.float64, np.float32):
        stats, t = [], time.time()

        def solve(J, b):
            J = J.tocsr()
            i = int(np.flatnonzero(~prob.isbc)[0])
            mg = KP.CollectiveMG(prob.K, prob.M, -J[n:, n:], J[i, i] / prob.K[i, i], N, prob.isbc, nu=6, omega=0.75)
            x, its, rs, reads = fgmres(J, b, lambda r: np.concatenate(mg.vcycle(r[:n], r[n:])), store)
            stats.append((its, rs, reads))
            return x

        x, h = O.solve_problem(prob, 500, "double_exponential", 1e2, 1e-4, linear_solve=solve)
        its = [s[0] for s in stats]
        print(f"basis stored as {np.dtype(store).name}: Newton {h['Newton steps']}  Krylov per solve {its}  total {sum(its)}  cycles {sum(s[1] for s in stats)} "
              f"for {len(stats)} solves  Gram-Schmidt basis reads {sum(s[2] for s in stats):.0f} fp64-vectors  ({time.time() - t:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
