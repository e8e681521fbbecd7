import sys, threading
sys.path.insert(0, '.')
import numpy as np
from oracle import pg_oracle as O
from proximalgalerkin_amd.direct import DirectSolver
from proximalgalerkin_amd import comm as pcomm
N = 40
coords, cells = O.create_rectangle(N, N)
p1 = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
its = []
O.solve_problem(p1, 500, "double_exponential", 1e2, 1e-4, iterates=its)
J = p1.jacobian(its[-2], 100.0).tocsr(); J.sort_indices()
nod = np.concatenate([np.arange(p1.n)] * 2)
b = np.random.default_rng(2).standard_normal(J.shape[0])
def run(c, tag):
    ds = DirectSolver(J.indptr, J.indices, nod, p1.coords, leaf_nodes=8, device=0, comm=c)
    ds.factor(J.data); x1 = ds.solve(b)
    ds.factor(J.data); x1b = ds.solve(b)
    ds.factor(J.data * 2.0); x2 = ds.solve(b)
    n = np.linalg.norm(x1)
    print(tag, "repeat", np.linalg.norm(x1b - x1) / n, "scaled", np.linalg.norm(2 * x2 - x1) / n, "res", np.linalg.norm(J @ x1 - b) / np.linalg.norm(b), flush=True)
    ds.close()
run(None, "single")
cs = pcomm.local_group(2)
th = [threading.Thread(target=run, args=(cs[r], f"rank{r}")) for r in range(2)]
[t.start() for t in th]; [t.join() for t in th]
