"""oracle/nd_lu.py - the oracle's nested-dissection multifrontal LU (the reference's `pc_type lu` / MUMPS,
/root/reference/examples/01_obstacle_problem/obstacle_pg.py:129-131) against SuperLU and against the oracle's default solver.
CPU only; sizes that finish in seconds."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from oracle import nd_lu as ND
from oracle import pg_oracle as O


def _p1(N):
    coords, cells = O.create_rectangle(N, N)
    return O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))


def _late_iterate(prob, rng):
    """An iterate like the late proximal steps: psi down to -800 inside a disk, so exp(psi) underflows to exact zeros there."""
    n = prob.n
    c = getattr(prob, "dof_coords", prob.coords)[:n]
    r = np.hypot(c[:, 0], c[:, 1])
    x = np.concatenate([0.1 * rng.standard_normal(n), np.where(r < 0.4, -800.0, -1.0) + 0.1 * rng.standard_normal(n)])
    return x


@pytest.mark.parametrize("N,alpha", [(24, 1.0), (40, 85.0)])
def test_solution_matches_superlu_on_newton_matrices(N, alpha):
    prob = _p1(N)
    rng = np.random.default_rng(N)
    x = _late_iterate(prob, rng)
    J = prob.jacobian(x, alpha)
    assert (J.diagonal()[prob.n:] == 0.0).any()  # exact-zero pivots candidates on the latent diagonal
    nd = ND.NDLU(J, *ND.nodes_of_problem(prob), leaf_nodes=16)
    nd.factor(J)
    b = rng.standard_normal(2 * prob.n)
    y = nd.solve(b)

    def backward_error(A, v):  # normwise: the solve is exact for a matrix within this relative distance of A
        return np.linalg.norm(A @ v - b) / (spla.norm(A) * np.linalg.norm(v) + np.linalg.norm(b))

    assert backward_error(J, y) <= 1e-14
    y_ref = spla.splu(J.tocsc()).solve(b)
    n = prob.n
    # forward agreement is limited by the conditioning of these matrices (the solution has entries of 1e4 at alpha = 85)
    assert np.linalg.norm(y[:n] - y_ref[:n]) <= (1e-9 if alpha == 1.0 else 1e-6) * np.linalg.norm(y_ref[:n])
    # a second factorisation in the same arena (new values, same pattern) is independent of the first
    J2 = prob.jacobian(_late_iterate(prob, rng), 2.0 * alpha)
    nd.factor(J2)
    y2 = nd.solve(b)
    assert backward_error(J2, y2) <= 1e-14
    with pytest.raises(ValueError):
        nd.factor(J2[:, ::-1].tocsr())


def test_numpy_fallback_equals_the_c_helper(monkeypatch):
    prob = _p1(20)
    rng = np.random.default_rng(3)
    J = prob.jacobian(_late_iterate(prob, rng), 3.0)
    b = rng.standard_normal(2 * prob.n)
    nd = ND.NDLU(J, *ND.nodes_of_problem(prob), leaf_nodes=8)
    nd.factor(J)
    y1 = nd.solve(b)
    assert ND._HELPER is not None, "oracle/_build/libndhelper.so did not build (gcc missing?)"
    monkeypatch.setattr(ND, "_HELPER", None)
    nd.factor(J)
    y2 = nd.solve(b)
    assert np.array_equal(y1, y2)


@pytest.mark.parametrize("degree,N", [(1, 32), (2, 12)])
def test_lvpp_run_reproduces_the_superlu_oracle(degree, N):
    """The whole LVPP run (settings B) with the ND solver in the `linear_solve` slot: identical Newton counts per proximal step and
    the same primal field as the default SuperLU path of the oracle."""
    coords, cells = O.create_rectangle(N, N)
    prob = _p1(N) if degree == 1 else O.ObstacleLagrange(coords, cells, degree=2)
    ls = ND.NDLinearSolve(*ND.nodes_of_problem(prob), leaf_nodes=16)
    x, h = O.solve_problem(prob, 100, "double_exponential", 1e2, 1e-4, linear_solve=ls)
    x_ref, h_ref = O.solve_problem(prob, 100, "double_exponential", 1e2, 1e-4)
    assert h["Newton steps"] == h_ref["Newton steps"]
    n = prob.n
    assert np.linalg.norm(x[:n] - x_ref[:n]) <= 1e-11 * np.linalg.norm(x_ref[:n])
    assert ls.n_factor == sum(h["Newton steps"]) and ls.last_relres < 1e-11


def test_ordering_statistics_follow_nested_dissection():
    """Fill and flops of the 2-D dissection: O(n log n) / O(n^1.5).  Doubling N must multiply the flops by about 8 and the factor
    entries by about 4.4, nowhere near COLAMD's N^3.2 on these saddle points (profiles/r02_cpu_ladder.json)."""
    st = []
    for N in (32, 64, 128):
        prob = _p1(N)
        J = prob.jacobian(np.zeros(2 * prob.n), 1.0)
        nd = ND.NDLU(J, *ND.nodes_of_problem(prob))
        st.append((nd.flops, nd.factor_entries))
    assert 5.0 < st[2][0] / st[1][0] < 9.5 and 3.5 < st[2][1] / st[1][1] < 5.5


def test_nd_lu_runs_example_06_like_superlu():
    """The nested-dissection LU as the linear solver of the example-06 oracle (node = vertex with (u, psi_x, psi_y) or edge midpoint
    with u): same Newton counts and solution as the SuperLU default - it is what bench.py --workload ex06 times as cpu_baseline."""
    from oracle import gc_oracle as G

    c, e = O.create_rectangle(12, 12, (0.0, 0.0), (1.0, 1.0))
    prob = G.GradientConstraintP2(c, e)
    x0, its0, _ = G.solve_problem(prob)
    ls = ND.NDLinearSolve(*ND.nodes_of_problem(prob))
    x1, its1, _ = G.solve_problem(prob, linear_solve=ls)
    assert list(its0) == list(its1)
    assert np.linalg.norm(x0 - x1) <= 1e-9 * np.linalg.norm(x0)
    assert ls.last_relres < 1e-12


def test_nd_lu_runs_example_02_like_superlu():
    """3-D: the Signorini oracle (vertex node = three displacement components + psi on the contact boundary; the Dirichlet rows cut
    components off, i.e. fronts without border) with the nested-dissection LU: same Newton counts as SuperLU, same solution to the
    Newton tolerance - what bench.py --workload ex02 times as cpu_baseline."""
    from oracle import sg_oracle as S

    m = 5
    c, t = S.create_unit_cube_tets(m, m, m)
    prob = S.SignoriniP1(c, t, S.boundary_facets_where(c, t, lambda x: np.isclose(x[:, 2], 0.0)), np.flatnonzero(np.isclose(c[:, 2], 1.0)))
    x0, _, its0 = S.solve_contact_problem(prob)
    ls = ND.NDLinearSolve(*ND.nodes_of_problem(prob))
    x1, _, its1 = S.solve_contact_problem(prob, linear_solve=ls)
    assert list(its0) == list(its1)
    assert np.linalg.norm(x0 - x1) <= 1e-4 * np.linalg.norm(x0)  # both stop at the Newton tolerance 1e-6 (1e-5 on the first step)
    assert ls.last_relres < 1e-12


def test_tree_parallel_factorisation_equals_the_serial_one():
    """NDLU.factor(workers=4): the four subtrees at tree depth 2 factorised by forked workers into the shared arena, the two levels
    above them by the parent - the solution of a late-Newton-like system equals the serial schedule's to the solver's accuracy,
    pivots and factors of EVERY front are readable from the parent (the solve phase walks all of them), and a second factorisation
    reuses the shared buffers."""
    import numpy as np

    from oracle import nd_lu as ND
    from oracle import pg_oracle as O

    N = 48
    coords, cells = O.create_rectangle(N, N)
    prob = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    x = np.zeros(2 * prob.n)
    r = np.hypot(coords[:, 0], coords[:, 1])
    x[prob.n:] = np.where(r < 0.35, -300.0, -1.0)  # exp(psi) underflows to exact zeros in the contact zone
    J = prob.jacobian(x, 50.0).tocsr()
    b = np.random.default_rng(3).standard_normal(J.shape[0])
    serial = ND.NDLinearSolve(*ND.nodes_of_problem(prob))
    xs = serial(J, b)
    par = ND.NDLinearSolve(*ND.nodes_of_problem(prob), workers=4)
    xp = par(J, b)
    assert par.nd._plan is not None and len(par.nd._plan["roots"]) == 4
    # (alpha = 50 and exact zeros in D: condition number ~1e10 - refinement stops at 1e-11 for either schedule)
    assert par.last_relres < 1e-9 and serial.last_relres < 1e-9 and par.last_relres < 100 * serial.last_relres + 1e-12
    assert np.linalg.norm(xp - xs) <= 1e-6 * np.linalg.norm(xs)
    xp2 = par(J, b)  # refactorisation into the same shared buffers
    assert np.array_equal(xp2, xp) or np.linalg.norm(xp2 - xp) <= 1e-12 * np.linalg.norm(xp)
