"""CPU tests of the sparse direct solver's symbolic phase (C++ in libpgx.so, no GPU touched): the exported level /
front / destination maps are fed to a numpy emulation of the device numeric phase (tests/nd_emulate.py) and the result
is compared with SuperLU on Newton matrices of examples 01 (P1, P2) and 06."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from oracle import gc_oracle as G
from oracle import pg_oracle as O
from proximalgalerkin_amd.direct import DirectSolver
from tests import nd_emulate as E


def _cases():
    N = 12
    coords, cells = O.create_rectangle(N, N)
    p1 = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    rng = np.random.default_rng(1)
    x = 0.3 * rng.standard_normal(2 * p1.n)
    x[p1.n:] -= 40.0 * (np.hypot(*p1.coords.T) < 0.4)  # exp(psi) underflows towards 0 in a "contact zone"
    yield "ex01-P1", p1.jacobian(x, 7.0), np.concatenate([np.arange(p1.n)] * 2), p1.coords, 8
    p2 = O.ObstacleLagrange(coords, cells, degree=2)
    x = 0.3 * rng.standard_normal(2 * p2.n)
    yield "ex01-P2", p2.jacobian(x, 3.0), np.concatenate([np.arange(p2.n)] * 2), p2.dof_coords, 12
    c6, e6 = O.create_rectangle(10, 10, (0.0, 0.0), (1.0, 1.0))
    g = G.GradientConstraintP2(c6, e6)
    x = rng.standard_normal(g.ntot) * 3.0
    yield "ex06", g.jacobian(x, 16.0), np.concatenate([np.arange(g.n2), np.arange(g.nv), np.arange(g.nv)]), g.dof_coords, 10


@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: c[0])
def test_symbolic_maps_reproduce_lu(case):
    name, J, node_of_dof, node_coords, leaf = case
    J = J.tocsr()
    J.sort_indices()
    ds = DirectSolver(J.indptr, J.indices, node_of_dof, node_coords, leaf_nodes=leaf, device=-1)
    st = ds.stats()
    assert st["n_levels"] >= 3 and st["flops_padded"] >= st["flops"] > 0
    sym = ds.export_symbolic()
    # every dof is owned by exactly one front
    assert np.array_equal(np.sort(sym["own_dofs"]), np.arange(J.shape[0]))
    fac = E.factor(sym, J.data)
    b = np.random.default_rng(0).standard_normal(J.shape[0])
    x = E.solve(sym, fac, b)
    xr = spla.splu(J.tocsc()).solve(b)
    assert np.linalg.norm(J @ x - b) <= 1e-10 * np.linalg.norm(b)
    assert np.linalg.norm(x - xr) <= 1e-8 * np.linalg.norm(xr)


def test_subtree_sequencing_keeps_the_maps_valid(monkeypatch):
    """PGX_ND_CUT_GB=0 forces the large-problem schedule (tree cut at depth 3, batches per subtree): the exported maps still
    reproduce LU, with more, smaller batches and less device storage; flop count and factor size do not change."""
    N = 40
    coords, cells = O.create_rectangle(N, N)
    p1 = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    x = 0.3 * np.random.default_rng(5).standard_normal(2 * p1.n)
    J = p1.jacobian(x, 5.0).tocsr()
    J.sort_indices()
    nod = np.concatenate([np.arange(p1.n)] * 2)
    monkeypatch.setenv("PGX_ND_CUT_GB", "-1")
    plain = DirectSolver(J.indptr, J.indices, nod, p1.coords, leaf_nodes=8, device=-1)
    monkeypatch.setenv("PGX_ND_CUT_GB", "0")
    cut = DirectSolver(J.indptr, J.indices, nod, p1.coords, leaf_nodes=8, device=-1)
    sp_, sc_ = plain.stats(), cut.stats()
    assert sc_["n_levels"] > sp_["n_levels"] and sc_["arena_doubles"] < sp_["arena_doubles"]
    assert sc_["flops"] == sp_["flops"] and sc_["factor_nnz"] == sp_["factor_nnz"] and sc_["n_fronts"] == sp_["n_fronts"]
    sym = cut.export_symbolic()
    assert np.all(np.diff(sym["depth"]) >= 0)  # batches stay ordered by tree depth
    fac = E.factor(sym, J.data)
    b = np.random.default_rng(0).standard_normal(J.shape[0])
    xs = E.solve(sym, fac, b)
    assert np.linalg.norm(J @ xs - b) <= 1e-10 * np.linalg.norm(b)


def test_symbolic_handle_refuses_numeric_phase_without_gpu():
    coords, cells = O.create_rectangle(4, 4)
    p1 = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(4, 4))
    J = p1.jacobian(np.zeros(2 * p1.n), 1.0).tocsr()
    J.sort_indices()
    ds = DirectSolver(J.indptr, J.indices, np.concatenate([np.arange(p1.n)] * 2), p1.coords, device=-1)
    from proximalgalerkin_amd._lib import PgxError
    with pytest.raises(PgxError):
        ds.factor(J.data)


def test_unsymmetric_pattern_is_rejected():
    import scipy.sparse as sp
    A = sp.csr_matrix(np.array([[1.0, 2.0, 0.0], [0.0, 1.0, 0.0], [0.0, 3.0, 1.0]]))
    from proximalgalerkin_amd._lib import PgxError
    with pytest.raises(PgxError):
        DirectSolver(A.indptr, A.indices, np.arange(3), np.array([[0.0, 0.0], [1.0, 0.0], [2.0, 0.0]]), leaf_nodes=1,
                     device=-1)


@pytest.mark.parametrize("R", [2, 4])
def test_distributed_symbolic_views_reproduce_lu(R):
    """The N>1 logic of the sparse LU without a GPU: every rank's symbolic view (pgx_nd_create_symbolic_dist: ownership of
    the dissection subtrees, ghost subtree roots on rank 0, assembly destinations, child->parent maps) drives a numpy
    emulation of the distributed numeric phase incl. the gather / scatter exchanges (tests/nd_emulate.py); the result must
    solve the system like SuperLU does."""
    N = 14
    coords, cells = O.create_rectangle(N, N)
    p1 = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    rng = np.random.default_rng(4)
    x0 = 0.3 * rng.standard_normal(2 * p1.n)
    x0[p1.n:] -= 40.0 * (np.hypot(*p1.coords.T) < 0.4)
    J = p1.jacobian(x0, 7.0).tocsr()
    J.sort_indices()
    nod = np.concatenate([np.arange(p1.n)] * 2)
    syms = []
    for r in range(R):
        ds = DirectSolver(J.indptr, J.indices, nod, p1.coords, leaf_nodes=6, device=-1, symbolic_rank=(r, R))
        syms.append(ds.export_symbolic())
    # every dof is eliminated on exactly one rank, every matrix entry assembled on exactly one rank
    owned = np.concatenate([s["own_dofs"] for s in syms])
    assert np.array_equal(np.sort(owned), np.arange(J.shape[0]))
    assert np.array_equal(sum((s["dest"] >= 0).astype(int) for s in syms), np.ones(J.nnz, dtype=int))
    assert all(s["dist"]["kdist"] == int(np.log2(R)) for s in syms)
    sts = E.factor_dist(syms, J.data)
    b = rng.standard_normal(J.shape[0])
    x = E.solve_dist(syms, sts, b)
    xr = spla.splu(J.tocsc()).solve(b)
    assert np.linalg.norm(J @ x - b) <= 1e-10 * np.linalg.norm(b)
    assert np.linalg.norm(x - xr) <= 1e-8 * np.linalg.norm(xr)


def test_threaded_maps_do_not_depend_on_the_thread_count(monkeypatch):
    """The index maps and assembly destinations are built by worker threads, one front at a time (PGX_ND_THREADS; private index
    arrays): every output belongs to exactly one front, so 1, 3 and 8 threads export identical structures.  Also pins the
    lighter-side separators: on a P2 lattice (nodes = vertices and edge midpoints) the root separator is ONE node line."""
    N = 16
    coords, cells = O.create_rectangle(N, N)
    p2 = O.ObstacleLagrange(coords, cells, degree=2)
    J = p2.jacobian(0.3 * np.random.default_rng(2).standard_normal(2 * p2.n), 3.0).tocsr()
    J.sort_indices()
    nod = np.concatenate([np.arange(p2.n)] * 2)
    syms = []
    for nthr in ("1", "3", "8"):
        monkeypatch.setenv("PGX_ND_THREADS", nthr)
        ds = DirectSolver(J.indptr, J.indices, nod, p2.dof_coords, leaf_nodes=8, device=-1)
        syms.append(ds.export_symbolic())
        ds.close()
    for other in syms[1:]:
        for key in ("lev_start", "P", "B", "fp", "fb", "parent", "slot01", "dof_ptr", "own_dofs", "rel_ptr", "rel", "dest"):
            assert np.array_equal(syms[0][key], other[key]), key
    # root front: one lattice line of the (2N+1)^2 P2 nodes, two dofs (u, psi) per node - not the two lines a one-sided rule takes
    root = int(np.flatnonzero(syms[0]["parent"] < 0)[0])
    assert syms[0]["fp"][root] == 2 * (2 * N + 1), syms[0]["fp"][root]
