"""Symbolic statistics of the nested-dissection direct solver for the ex 01 mixed P1 Newton matrix on an N x N mesh
(host only, no GPU): python tools/nd_stats.py N [leaf_nodes]"""
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from proximalgalerkin_amd.direct import DirectSolver  # noqa: E402


def p1_mixed_pattern(N):
    n1 = N + 1
    nv = n1 * n1
    i, j = np.meshgrid(np.arange(n1), np.arange(n1), indexing="xy")
    v = (j * n1 + i).ravel()
    rows, cols = [v], [v]
    for di, dj in ((1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1)):
        ok = ((i + di >= 0) & (i + di < n1) & (j + dj >= 0) & (j + dj < n1)).ravel()
        rows.append(v[ok])
        cols.append(v[ok] + dj * n1 + di)
    S = sp.csr_matrix((np.ones(sum(map(len, rows)), dtype=np.int8), (np.concatenate(rows), np.concatenate(cols))), shape=(nv, nv))
    J = sp.bmat([[S, S], [S, S]], format="csr")
    J.sort_indices()
    coords = np.stack([i.ravel() / N, j.ravel() / N], axis=1).astype(float)
    return J, np.concatenate([np.arange(nv)] * 2), coords


if __name__ == "__main__":
    N = int(sys.argv[1])
    leaf = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    J, nod, coords = p1_mixed_pattern(N)
    t = time.time()
    ds = DirectSolver(J.indptr, J.indices, nod, coords, leaf_nodes=leaf, device=-1)
    st = ds.stats()
    print(f"N={N} leaf={leaf} n={J.shape[0]} nnz={J.nnz} symbolic {time.time() - t:.1f}s: fronts={st['n_fronts']} levels={st['n_levels']} "
          f"max_front={st['max_front']} arena={st['arena_doubles'] * 8 / 1e9:.2f} GB factor_nnz={st['factor_nnz'] / 1e6:.0f}M "
          f"Gflop={st['flops'] / 1e9:.0f} padded={st['flops_padded'] / 1e9:.0f}")
    sym = ds.export_symbolic()
    for l, (P, B) in enumerate(zip(sym["P"], sym["B"])):
        cnt = sym["lev_start"][l + 1] - sym["lev_start"][l]
        print(f"  level {l}: {cnt} fronts, P={P} B={B}")
