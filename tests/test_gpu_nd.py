"""GPU numeric phase of the sparse direct solver (pgx_nd) against SuperLU, through the C ABI, on Newton matrices of
examples 01 (P1, P2) and 06 built by the CPU oracle.  Tolerances: normwise backward error |b - Jx| / (|J|_1 |x| + |b|)
<= 1e-13 (what a backward-stable fp64 LU delivers; the plain relative residual is bounded by cond(J) * eps, and the late
matrices have condition numbers of 1e10 and more), and the solution within rtol_x of SuperLU's."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from oracle import gc_oracle as G
from oracle import pg_oracle as O

pytestmark = pytest.mark.gpu


def _berr(J, x, b):
    return np.linalg.norm(J @ x - b) / (abs(J).sum(axis=0).max() * np.linalg.norm(x) + np.linalg.norm(b))


def _check(J, node_of_dof, node_coords, leaf, rtol_x=1e-7):
    from proximalgalerkin_amd.direct import DirectSolver
    J = J.tocsr()
    J.sort_indices()
    ds = DirectSolver(J.indptr, J.indices, node_of_dof, node_coords, leaf_nodes=leaf, device=0)
    ds.factor(J.data)
    rng = np.random.default_rng(3)
    lu = spla.splu(J.tocsc())
    for _ in range(2):
        b = rng.standard_normal(J.shape[0])
        x = ds.solve(b)
        assert np.all(np.isfinite(x))
        xr = lu.solve(b)
        assert _berr(J, x, b) <= 1e-13, (_berr(J, x, b), _berr(J, xr, b))
        assert np.linalg.norm(x - xr) <= rtol_x * np.linalg.norm(xr)
    # refactor with other values on the same pattern (what every Newton step does)
    sc = 1.0 + 0.3 * np.sin(np.arange(J.shape[0]))
    J2 = J.copy()
    J2.data = J.data * sc[np.repeat(np.arange(J.shape[0]), np.diff(J.indptr))] * sc[J.indices]  # S J S: same pattern
    ds.factor(J2.data)
    b = rng.standard_normal(J.shape[0])
    x = ds.solve(b)
    assert _berr(J2, x, b) <= 1e-13
    ds.close()


@pytest.mark.parametrize("N,leaf", [(12, 8), (40, 32), (64, 0)])
def test_ex01_p1_newton_matrix(require_gpu, N, leaf):
    coords, cells = O.create_rectangle(N, N)
    p1 = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    its = []
    O.solve_problem(p1, 500, "double_exponential", 1e2, 1e-4, iterates=its)
    J = p1.jacobian(its[-2], 100.0)  # late step: exp(psi) underflows in the contact zone
    _check(J, np.concatenate([np.arange(p1.n)] * 2), p1.coords, leaf)


def test_ex01_p2_newton_matrix(require_gpu):
    N = 24
    coords, cells = O.create_rectangle(N, N)
    p2 = O.ObstacleLagrange(coords, cells, degree=2)
    its = []
    O.solve_problem(p2, 500, "double_exponential", 1e2, 1e-4, iterates=its)
    _check(p2.jacobian(its[-2], 100.0), np.concatenate([np.arange(p2.n)] * 2), p2.dof_coords, 24)


def test_ex06_newton_matrix(require_gpu):
    N = 20
    c6, e6 = O.create_rectangle(N, N, (0.0, 0.0), (1.0, 1.0))
    g = G.GradientConstraintP2(c6, e6)
    its = []
    G.solve_problem(g, max_iterations=8, iterates=its)
    nod = np.concatenate([np.arange(g.n2), np.arange(g.nv), np.arange(g.nv)])
    _check(g.jacobian(its[1], 4.0), nod, g.dof_coords, 16)
    _check(g.jacobian(its[-1], 256.0), nod, g.dof_coords, 16, rtol_x=1e-4)


def test_subtree_sequencing_gives_the_same_factorisation(require_gpu, monkeypatch):
    """Large factorisations cut the tree at depth 3 and factorise the subtrees one after the other (PGX_ND_CUT_GB, default
    160 GB of device storage); forced here on a small matrix: same solutions as SuperLU, less storage than without the cut."""
    from proximalgalerkin_amd.direct import DirectSolver

    N = 48
    coords, cells = O.create_rectangle(N, N)
    p1 = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    its = []
    O.solve_problem(p1, 500, "double_exponential", 1e2, 1e-4, iterates=its)
    J = p1.jacobian(its[-2], 100.0).tocsr()
    J.sort_indices()
    nod = np.concatenate([np.arange(p1.n)] * 2)
    monkeypatch.setenv("PGX_ND_CUT_GB", "-1")
    plain = DirectSolver(J.indptr, J.indices, nod, p1.coords, device=-1).stats()
    monkeypatch.setenv("PGX_ND_CUT_GB", "0")
    cut = DirectSolver(J.indptr, J.indices, nod, p1.coords, device=-1).stats()
    assert cut["n_levels"] > plain["n_levels"] and cut["arena_doubles"] < plain["arena_doubles"]
    assert cut["flops"] == plain["flops"] and cut["factor_nnz"] == plain["factor_nnz"]
    _check(J, nod, p1.coords, 0)


def test_perturbed_pivots_are_reported(require_gpu):
    """A structurally fine but numerically singular matrix: the static pivot perturbation must not pass silently
    (pgx_nd_stats.perturbed_pivots, pgx_nd_last_error, RuntimeWarning from the host mirror); a regular matrix reports 0."""
    import scipy.sparse as sp
    from proximalgalerkin_amd.direct import DirectSolver

    n = 12
    idx = np.arange(n * n).reshape(n, n)
    A = sp.lil_matrix((n * n, n * n))
    for (di, dj) in ((0, 0), (0, 1), (1, 0), (0, -1), (-1, 0)):
        src = idx[max(0, -di):n - max(0, di), max(0, -dj):n - max(0, dj)].ravel()
        dst = idx[max(0, di):n - max(0, -di), max(0, dj):n - max(0, -dj)].ravel()
        A[src, dst] = 4.0 if (di, dj) == (0, 0) else -1.0
    A = A.tocsr()
    A.sort_indices()
    coords = np.stack(np.meshgrid(np.arange(n, dtype=float), np.arange(n, dtype=float), indexing="ij"), -1).reshape(-1, 2)
    coords = np.concatenate([coords, np.zeros((n * n, 1))], axis=1)
    ds = DirectSolver(A.indptr, A.indices, np.arange(n * n), coords, device=0)
    ds.factor(A.data)
    assert ds.stats()["perturbed_pivots"] == 0
    Z = A.copy()
    Z.data[:] = 0.0  # same pattern, every pivot exactly zero
    with pytest.warns(RuntimeWarning, match="zero pivot"):
        ds.factor(Z.data)
    assert ds.stats()["perturbed_pivots"] > 0
    ds.factor(A.data)  # and the report is per factorisation
    assert ds.stats()["perturbed_pivots"] == 0
    ds.close()


def test_fused_leaves_agree_with_the_batched_kernels(require_gpu, monkeypatch):
    """k_nd_leaf (one wave assembles and eliminates a childless front; default) against the level-batched kernels
    (PGX_ND_LEAF_FUSED=0) on a late ex 06 matrix with zero pivot padding, leaves of mixed size and |psi| up to 1e3: both are LU
    factorisations without pivoting of the same matrix in the same order - solutions agree to rounding, both backward stable."""
    from proximalgalerkin_amd.direct import DirectSolver

    N = 24
    c6, e6 = O.create_rectangle(N, N, (0.0, 0.0), (1.0, 1.0))
    g = G.GradientConstraintP2(c6, e6)
    its = []
    G.solve_problem(g, max_iterations=8, iterates=its)
    J = g.jacobian(its[-1], 128.0).tocsr()
    J.sort_indices()
    nod = np.concatenate([np.arange(g.n2), np.arange(g.nv), np.arange(g.nv)]).astype(np.int32)
    b = np.random.default_rng(9).standard_normal(J.shape[0])
    xs = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("PGX_ND_LEAF_FUSED", fused)
        ds = DirectSolver(J.indptr, J.indices, nod, g.dof_coords, device=0)
        ds.factor(J.data)
        xs[fused] = ds.solve(b)
        assert ds.stats()["perturbed_pivots"] == 0
        assert _berr(J, xs[fused], b) <= 1e-13
        ds.factor(J.data)  # refactorisation is bitwise reproducible on either path
        assert np.array_equal(ds.solve(b), xs[fused])
        ds.close()
    assert np.linalg.norm(xs["1"] - xs["0"]) <= 1e-9 * np.linalg.norm(xs["0"])
    assert not np.array_equal(xs["1"], xs["0"])  # different summation order in the leaves: the switch does select another kernel


def test_assembly_variants_agree_on_an_unstructured_p2_matrix(require_gpu, monkeypatch):
    """Parent-centric assembly (k_nd_gather + the GATHER Schur updates, default) against zero fill + push-style extend-add
    (PGX_ND_GATHER=0) on the P2 Newton matrix of a DISK mesh: an unbalanced dissection tree with leaves at several depths,
    fronts without a second child and padded batches.  Same factorisation order, different summation order of the children's
    contributions: solutions agree to rounding, both are backward stable, each is bitwise reproducible."""
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.direct import DirectSolver

    msh = fem.create_disk(0.06)
    p2 = O.ObstacleLagrange(msh.geometry, msh.cells, degree=2)
    rng = np.random.default_rng(4)
    x = 0.3 * rng.standard_normal(2 * p2.n)
    x[p2.n:] -= 60.0 * (np.hypot(*p2.dof_coords.T) < 0.35)  # exp(psi) underflows towards 0 in a contact zone
    J = p2.jacobian(x, 20.0).tocsr()
    J.sort_indices()
    nod = np.concatenate([np.arange(p2.n)] * 2).astype(np.int32)
    b = rng.standard_normal(J.shape[0])
    xs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PGX_ND_GATHER", mode)
        ds = DirectSolver(J.indptr, J.indices, nod, p2.dof_coords, device=0)
        ds.factor(J.data)
        xs[mode] = ds.solve(b)
        assert _berr(J, xs[mode], b) <= 1e-13, _berr(J, xs[mode], b)
        ds.factor(J.data)
        assert np.array_equal(ds.solve(b), xs[mode])
        ds.close()
    xr = spla.splu(J.tocsc()).solve(b)
    assert np.linalg.norm(xs["1"] - xr) <= 1e-7 * np.linalg.norm(xr)
    assert np.linalg.norm(xs["1"] - xs["0"]) <= 1e-9 * np.linalg.norm(xs["0"])


@pytest.mark.parametrize("switch", ["PGX_ND_SOLVE_SMALL", "PGX_ND_TRSV_BIG", "PGX_ND_LSHAPE", "PGX_ND_LEFTLOOK", "PGX_ND_OUTER", "PGX_ND_TILEORDER"])
def test_round5_schedules_agree_with_the_round4_ones(require_gpu, monkeypatch, switch):
    """Every piece of the numeric phase that round 5 rebuilt has its round-4 form behind a tuning key: the one-wave solve sweeps of the
    small fronts, the register-resident slab solve of the large ones, the L-shaped trailing update, the left-looking 64-pivot steps,
    the 512-pivot outer blocks (PGX_ND_OUTER=256 / 128: three to six outer blocks in the root of this matrix).  On a late example-06
    matrix with fronts of every kind (leaves, frame-fused, P > 64 with slabs, a 770-pivot root) both forms are backward stable LU
    factorisations / solves of the same matrix in the same order: solutions agree to rounding; the per-depth profile
    (pgx_nd_depth_profile) accounts for every factorisation and sweep."""
    from proximalgalerkin_amd.direct import DirectSolver

    N = 128
    c6, e6 = O.create_rectangle(N, N, (0.0, 0.0), (1.0, 1.0))
    g = G.GradientConstraintP2(c6, e6)
    x = np.zeros(g.ntot)
    x[g.n2:] = 0.3 * np.sin(np.arange(2 * g.nv))
    J = g.jacobian(x, 4.0).tocsr()
    J.sort_indices()
    nod = np.concatenate([np.arange(g.n2), np.arange(g.nv), np.arange(g.nv)]).astype(np.int32)
    b = np.random.default_rng(11).standard_normal(J.shape[0])
    xs = {}
    for val in ("default", "128" if switch == "PGX_ND_OUTER" else "0"):
        if val == "default":
            monkeypatch.delenv(switch, raising=False)
        else:
            monkeypatch.setenv(switch, val)
        ds = DirectSolver(J.indptr, J.indices, nod, g.dof_coords, device=0)
        ds.depth_profile(True)
        ds.factor(J.data)
        xs[val] = ds.solve(b)
        ds.solve(b)
        prof = ds.depth_profile(False)
        assert prof["calls"] == (1, 2, 2)
        assert len(prof["factor_ms"]) >= 10 and np.all(prof["factor_ms"] > 0) and np.all(prof["fwd_ms"] > 0) and np.all(prof["bwd_ms"] > 0)
        assert ds.stats()["perturbed_pivots"] == 0
        assert _berr(J, xs[val], b) <= 1e-13
        ds.close()
    a, c = xs.values()
    assert np.linalg.norm(a - c) <= 1e-9 * np.linalg.norm(c)



def test_symmetric_mode_factorises_half_and_agrees_with_the_general_lu(require_gpu, monkeypatch):
    """pgx_nd_set_symmetric (round 5): on a symmetric indefinite matrix - an example-06 Jacobian with fronts of every kind (leaves,
    frame-fused, P > 64, a 770-pivot root with outer blocks when PGX_ND_OUTER=128) - the factorisation that writes the U panels as
    scaled transposes, updates the lower part of the pivot blocks only and computes the Schur tiles on and below the diagonal is a
    backward stable factorisation of the same matrix: backward error <= 1e-13, solution equal to the general LU's to rounding,
    bitwise reproducible; its depth profile takes visibly less time on the root levels."""
    from proximalgalerkin_amd.direct import DirectSolver

    N = 128
    c6, e6 = O.create_rectangle(N, N, (0.0, 0.0), (1.0, 1.0))
    g = G.GradientConstraintP2(c6, e6)
    x = np.zeros(g.ntot)
    x[g.n2:] = 0.3 * np.sin(np.arange(2 * g.nv))
    J = g.jacobian(x, 4.0).tocsr()
    J.sort_indices()
    assert abs(J - J.T).max() <= 1e-12 * abs(J).max()
    nod = np.concatenate([np.arange(g.n2), np.arange(g.nv), np.arange(g.nv)]).astype(np.int32)
    b = np.random.default_rng(21).standard_normal(J.shape[0])
    for outer in (None, "128"):
        if outer:
            monkeypatch.setenv("PGX_ND_OUTER", outer)
        xs = {}
        for sym in (False, True):
            ds = DirectSolver(J.indptr, J.indices, nod, g.dof_coords, device=0)
            assert ds.set_symmetric(sym) == sym
            ds.factor(J.data)
            xs[sym] = ds.solve(b)
            assert ds.stats()["perturbed_pivots"] == 0
            assert _berr(J, xs[sym], b) <= 1e-13, (sym, outer, _berr(J, xs[sym], b))
            ds.factor(J.data)
            assert np.array_equal(ds.solve(b), xs[sym])
            ds.close()
        assert np.linalg.norm(xs[True] - xs[False]) <= 1e-9 * np.linalg.norm(xs[False])
    # the CUT schedule of large factorisations (subtrees one after the other, extend-add at the cut: PGX_ND_CUT_GB=0 forces it here)
    monkeypatch.setenv("PGX_ND_CUT_GB", "0")
    ds = DirectSolver(J.indptr, J.indices, nod, g.dof_coords, device=0)
    assert ds.set_symmetric(True) is True
    ds.factor(J.data)
    xc = ds.solve(b)
    assert _berr(J, xc, b) <= 1e-13 and np.linalg.norm(xc - xs[False]) <= 1e-9 * np.linalg.norm(xs[False])
    ds.close()
    monkeypatch.delenv("PGX_ND_CUT_GB")
    monkeypatch.setenv("PGX_ND_SYM", "0")  # the A/B key: the request is ignored
    ds = DirectSolver(J.indptr, J.indices, nod, g.dof_coords, device=0)
    assert ds.set_symmetric(True) is False
    ds.close()
