"""oracle/p2_patch_proto.py - the vertex-star patch smoother of the P2 level (prototype of pgx_patch.hip) on every Newton system of a
small LVPP run: the two-level cycle must converge in a handful of FGMRES iterations, early and late systems alike. CPU only."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import p2_patch_proto as P
from oracle import pg_oracle as O
from oracle.krylov_proto import fgmres


def test_star_patch_two_level_cycle_is_robust_on_all_systems_of_a_run():
    N = 12
    coords, cells = O.create_rectangle(N, N)
    prob = O.ObstacleLagrange(coords, cells, degree=2)
    systems = []

    def rec(J, b):
        systems.append((J.copy(), b.copy()))
        return spla.splu(J.tocsc()).solve(b)

    _, h = O.solve_problem(prob, 100, "double_exponential", 1e2, 1e-4, linear_solve=rec)
    assert len(systems) == sum(h["Newton steps"]) >= 15
    T1 = P.p1_to_p2(prob)
    T = sp.block_diag([T1, T1]).tocsr()
    worst = 0
    for J, b in systems:
        sm = P.StarSmoother(prob, J)
        # patches: 2 (1 + deg) slots, an edge dof in two patches, a vertex dof in one
        assert sm.idx.shape[1] == 14 and set(np.unique(sm.w)) <= {0.5, 1.0}
        luc = spla.splu((T.T @ J @ T).tocsc())

        def prec(r, J=J, sm=sm, luc=luc):
            x = np.zeros_like(r)
            for _ in range(2):
                x = sm.sweep(x, r, 1.0)
            x = x + T @ luc.solve(T.T @ (r - J @ x))
            for _ in range(2):
                x = sm.sweep(x, r, 1.0)
            return x

        xs, its, hist = fgmres(J, b, prec, 1e-10, 60)
        assert hist[-1] <= 1e-10
        assert np.linalg.norm(J @ xs - b) <= 1e-9 * np.linalg.norm(b)
        worst = max(worst, its)
    assert worst <= 14, worst
