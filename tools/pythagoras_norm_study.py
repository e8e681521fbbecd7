"""Can the norm of the projected Krylov vector come from Pythagoras - |w - V h|^2 = |w|^2 - |h|^2 with [h; |w|^2] from ONE batch of dot
products - whenever the selective second Gram-Schmidt pass is not taken, so that a sharded iteration needs one all-reduce less
?  numpy twin of the library loop (oracle/krylov_proto.py, V(6,6)), every Newton system of a settings-B run:
    python tools/pythagoras_norm_study.py 128
With the library threshold eta^2 = 1e-4 (second pass in 7 % of the iterations) the shortcut TRIPLES the Krylov iterations (162 -> 529 at
128^2, 431 at 256^2; the difference of squares even goes negative): up to four digits cancel inside the "safe" region.  With eta^2 = 0.5
it is harmless - and the second pass runs in 155 of 162 iterations, which costs more than the all-reduce saves.  Round 2 measured the
same on the GPU; the measured norm stays."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from oracle import krylov_proto as KP
from oracle import pg_oracle as O

def fgmres(A, b, prec, mode, rtol=1e-10, m=30, maxit=200, eta2=1e-4):
    bn = float(np.linalg.norm(b)); x = np.zeros_like(b); its = 0; second_passes = 0; r = b.copy(); restarts = 0
    while True:
        beta = float(np.linalg.norm(r))
        if beta <= rtol * bn or its >= maxit: return x, its, restarts, second_passes
        V = [r / beta]; Z = []; H = np.zeros((m + 1, m)); g = np.zeros(m + 1); g[0] = beta; cs = np.zeros(m); sn = np.zeros(m)
        for j in range(m):
            z = prec(V[j]); Z.append(z); w = A @ z
            Vm = np.stack(V, axis=1)
            h1 = Vm.T @ w; ww = float(w @ w)
            w = w - Vm @ h1
            wp2_true = float(w @ w)
            est = ww - float(h1 @ h1)
            wp2 = wp2_true if mode == "measured" else est
            second = wp2 < eta2 * (wp2 + float(h1 @ h1))
            H[: j + 1, j] = h1
            if second:
                second_passes += 1
                h2 = Vm.T @ w
                wp2_2 = wp2_true if mode != "measured" else wp2_true  # the second pass measures |w'|^2 in its own batch
                w = w - Vm @ h2
                H[: j + 1, j] += h2
                hn = np.sqrt(max(wp2_2 - float(h2 @ h2), 0.0))
            else:
                hn = np.sqrt(max(wp2, 0.0))
            H[j + 1, j] = hn
            V.append(w / hn)
            for i in range(j):
                t = cs[i] * H[i, j] + sn[i] * H[i + 1, j]; H[i + 1, j] = -sn[i] * H[i, j] + cs[i] * H[i + 1, j]; H[i, j] = t
            rr = np.hypot(H[j, j], H[j + 1, j]); cs[j], sn[j] = H[j, j] / rr, H[j + 1, j] / rr
            H[j, j], H[j + 1, j] = rr, 0.0; g[j + 1] = -sn[j] * g[j]; g[j] = cs[j] * g[j]; its += 1
            if abs(g[j + 1]) <= rtol * bn or its >= maxit: break
        k = j + 1
        y = np.linalg.solve(np.triu(H[:k, :k]), g[:k]); x = x + np.stack(Z, axis=1) @ y; r = b - A @ x; restarts += 1

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
coords, cells = O.create_rectangle(N, N)
prob = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N)); n = prob.n
for mode in ("measured", "pythagoras_when_safe"):
    stats = []; t = time.time()
    def solve(J, b):
        J = J.tocsr(); i = int(np.flatnonzero(~prob.isbc)[0])
        mg = KP.CollectiveMG(prob.K, prob.M, -J[n:, n:], J[i, i] / prob.K[i, i], N, prob.isbc, nu=6, omega=0.75)
        x, its, rs, sp = fgmres(J, b, lambda r: np.concatenate(mg.vcycle(r[:n], r[n:])), mode); stats.append((its, rs, sp)); return x
    x, h = O.solve_problem(prob, 500, "double_exponential", 1e2, 1e-4, linear_solve=solve)
    print(mode, "Newton", h["Newton steps"], "Krylov", [s[0] for s in stats], "total", sum(s[0] for s in stats), "cycles", sum(s[1] for s in stats), "second passes", sum(s[2] for s in stats), round(time.time() - t), "s", flush=True)
