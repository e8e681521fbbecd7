"""Time the direct solver on the ex 01 P1 Newton matrix of an N x N mesh (a late-step-like state):
python tools/nd_bench.py N [leaf_nodes] [reps]"""
import os
os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")  # PGX_* switches reach the library through the loader's opt-in bridge
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from proximalgalerkin_amd import fem  # noqa: E402
from proximalgalerkin_amd.direct import DirectSolver  # noqa: E402
from proximalgalerkin_amd.obstacle import setup_problem  # noqa: E402

N = int(sys.argv[1])
leaf = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
problem, sol, sol_k, alpha = setup_problem(msh)
n = problem.ndofs // 2
X = msh.geometry
r = np.hypot(X[:, 0], X[:, 1])
x = np.zeros(2 * n)
x[n:] = np.where(r < 0.35, -300.0, -1.0)  # contact-zone-like psi
alpha.value = 50.0
problem.assemble_jacobian(x)
rowptr, col, K, M, D = problem.export_blocks()
isbc = np.zeros(n, dtype=bool)
isbc[problem._keep[5]] = True
mk = lambda v: sp.csr_matrix((v, col, rowptr), shape=(n, n))  # noqa: E731
free = sp.diags((~isbc).astype(float))
A = free @ mk(50.0 * K) @ free + sp.diags(isbc.astype(float))
J = sp.bmat([[A, free @ mk(M)], [mk(M) @ free, mk(-D)]], format="csr")
# keep the full structural pattern (explicit zeros were dropped by the products): rebuild on the union pattern
S = mk(np.ones_like(K))
P = sp.bmat([[S, S], [S, S]], format="csr")
J = (J + 0.0 * P).tocsr()
J.sort_indices()
print(f"N={N} n={J.shape[0]} nnz={J.nnz}", flush=True)
t = time.time()
ds = DirectSolver(J.indptr, J.indices, np.concatenate([np.arange(n)] * 2), X, leaf_nodes=leaf, device=0)
st = ds.stats()
print(f"create (symbolic + upload) {time.time() - t:.2f}s fronts={st['n_fronts']} levels={st['n_levels']} max_front={st['max_front']} "
      f"arena={st['arena_doubles'] * 8 / 1e9:.2f} GB Gflop={st['flops'] / 1e9:.0f} (padded {st['flops_padded'] / 1e9:.0f})", flush=True)
ds.factor(J.data)
b = np.random.default_rng(0).standard_normal(J.shape[0])
xs = ds.solve(b)
print("residual", np.linalg.norm(J @ xs - b) / np.linalg.norm(b), flush=True)
x2 = xs + ds.solve(b - J @ xs)
print("refined ", np.linalg.norm(J @ x2 - b) / np.linalg.norm(b), flush=True)
ds.timing(True)
for _ in range(reps):
    ds.factor(J.data)
    ds.solve(b)
f, s = ds.timing(False)
print(f"factor {f / reps:.1f} ms ({st['flops'] / (f / reps) / 1e9:.2f} TFLOP/s unpadded)  solve {s / reps:.2f} ms", flush=True)
