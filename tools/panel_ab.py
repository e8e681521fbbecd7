"""A/B of the panel-solve kernels of the multifrontal LU (PGX_ND_PANEL = 0 MFMA blocked / 1 LDS-blocked scalar / 2 register-column
scalar) on synthetic grid matrices: factorisation time (pgx_nd_timing), normwise backward error and the difference between the
solutions.  python tools/panel_ab.py [2d N dofs | 3d n dofs] ...   (default: 2d 640 3, 3d 48 3)"""
import os
os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")  # PGX_* switches reach the library through the loader's opt-in bridge
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def grid_matrix(dim, n, nd, seed=0):
    """nd unknowns per node of an n^dim grid, (3^dim)-point stencil, nonsymmetric values, diagonally dominant."""
    rng = np.random.default_rng(seed)
    idx = np.arange(n**dim).reshape((n,) * dim)
    rows, cols = [], []
    for off in np.ndindex(*(3,) * dim):
        o = np.array(off) - 1
        src = tuple(slice(max(0, -k), n - max(0, k)) for k in o)
        dst = tuple(slice(max(0, k), n - max(0, -k)) for k in o)
        rows.append(idx[src].ravel())
        cols.append(idx[dst].ravel())
    r, c = np.concatenate(rows), np.concatenate(cols)
    nn = n**dim
    # all nd x nd couplings between neighbouring nodes
    R = (r[:, None, None] + nn * np.arange(nd)[None, :, None] + 0 * np.arange(nd)[None, None, :]).ravel()
    Cc = (c[:, None, None] + 0 * np.arange(nd)[None, :, None] + nn * np.arange(nd)[None, None, :]).ravel()
    v = rng.standard_normal(R.size)
    A = sp.csr_matrix((v, (R, Cc)), shape=(nn * nd, nn * nd))
    A = A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() * 1.05)
    A = A.tocsr()
    A.sort_indices()
    coords = np.stack(np.meshgrid(*(np.arange(n, dtype=float),) * dim, indexing="ij"), axis=-1).reshape(nn, dim)
    if dim == 2:
        coords = np.concatenate([coords, np.zeros((nn, 1))], axis=1)
    return A, np.tile(np.arange(nn), nd), coords


def main():
    from proximalgalerkin_amd.direct import DirectSolver

    args = sys.argv[1:] or ["2d", "640", "3", "3d", "48", "3"]
    for k in range(0, len(args), 3):
        dim, n, nd = int(args[k][0]), int(args[k + 1]), int(args[k + 2])
        A, nod, coords = grid_matrix(dim, n, nd)
        b = np.random.default_rng(1).standard_normal(A.shape[0])
        xs = {}
        for kind in (1, 0, 2, 0, 1):
            os.environ["PGX_ND_PANEL"] = str(kind)
            ds = DirectSolver(A.indptr, A.indices, nod, coords, device=0)
            ds.timing(True)
            ds.factor(A.data)
            ds.factor(A.data)
            t0 = time.perf_counter()
            ds.factor(A.data)
            x = ds.solve(b)
            wall = time.perf_counter() - t0
            fm, sm = ds.timing(True)
            st = ds.stats()
            be = np.linalg.norm(A @ x - b) / (abs(A).sum(axis=0).max() * np.linalg.norm(x) + np.linalg.norm(b))
            xs.setdefault(kind, x)
            d = np.linalg.norm(x - xs[1]) / np.linalg.norm(xs[1])
            print(f"{dim}d n={n} dofs/node={nd} unknowns={A.shape[0]} panel kind {kind}: factor {fm:8.2f} ms  solve {sm:6.2f} ms  "
                  f"({st['flops'] / fm / 1e9:6.2f} TFLOP/s)  berr {be:.2e}  |x - x_kind1|/|x| {d:.2e}  wall {wall * 1e3:.0f} ms", flush=True)
            ds.close()


if __name__ == "__main__":
    main()
