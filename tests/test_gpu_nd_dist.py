"""Distributed sparse LU (pgx_nd_create_dist) and the distributed example-02 handle (pgx_sg_create_dist), driven on ONE
GPU through the in-process transport (pgx_comm_local_group: one host thread per rank; gpurun boxes have a single GPU and
RCCL refuses two ranks on one device).  The same code path runs over RCCL with one process per GPU.
Checks: every rank returns the solution of the single-handle factorisation to rounding; example 02 on 2 and 4 ranks
reproduces the single-GPU LVPP run (Newton counts, displacement <= 1e-10)."""
import os
import threading

import numpy as np
import pytest
import scipy.sparse.linalg as spla

from oracle import pg_oracle as O
from oracle import sg_oracle as S

pytestmark = pytest.mark.gpu
# replicated handles must stay bitwise identical WITHOUT broadcasting residuals (atomic-free assembly, pgx_scatter.h): the
# library asserts it on every residual evaluation of the distributed runs below
os.environ["PGX_CHECK_REPLICAS"] = "1"


def _run_ranks(comms, fn):
    out, err = [None] * len(comms), [None] * len(comms)

    def work(r):
        try:
            out[r] = fn(comms[r])
        except BaseException as e:  # noqa: BLE001
            err[r] = e

    th = [threading.Thread(target=work, args=(r,)) for r in range(len(comms))]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("R", [2, 4])
def test_distributed_lu_matches_superlu(require_gpu, R):
    from proximalgalerkin_amd import comm as pcomm
    from proximalgalerkin_amd.direct import DirectSolver

    N = 40
    coords, cells = O.create_rectangle(N, N)
    p1 = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    its = []
    O.solve_problem(p1, 500, "double_exponential", 1e2, 1e-4, iterates=its)
    J = p1.jacobian(its[-2], 100.0).tocsr()
    J.sort_indices()
    nod = np.concatenate([np.arange(p1.n)] * 2)
    rng = np.random.default_rng(2)
    bs = [rng.standard_normal(J.shape[0]) for _ in range(2)]
    lu = spla.splu(J.tocsc())

    def rank_main(c):
        ds = DirectSolver(J.indptr, J.indices, nod, p1.coords, leaf_nodes=8, device=0, comm=c)
        ds.factor(J.data)
        xs = [ds.solve(b) for b in bs]
        ds.factor(J.data * 2.0)  # refactorisation with new values on the same pattern
        xs.append(ds.solve(bs[0]))
        ds.close()
        return xs

    res = _run_ranks(pcomm.local_group(R), rank_main)
    for xs in res:
        for x, b in zip(xs[:2], bs):
            xr = lu.solve(b)
            assert np.linalg.norm(J @ x - b) <= 1e-13 * (abs(J).sum(axis=0).max() * np.linalg.norm(x) + np.linalg.norm(b))
            assert np.linalg.norm(x - xr) <= 1e-7 * np.linalg.norm(xr)
        assert np.linalg.norm(2.0 * xs[2] - xs[0]) <= 1e-12 * np.linalg.norm(xs[0])
    for xs in res[1:]:  # every rank returns the same vectors
        for a, b in zip(xs, res[0]):
            assert np.array_equal(a, b)


def test_too_few_levels_is_an_error(require_gpu):
    from proximalgalerkin_amd import comm as pcomm
    from proximalgalerkin_amd._lib import PgxError
    from proximalgalerkin_amd.direct import DirectSolver

    coords, cells = O.create_rectangle(2, 2)
    p1 = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(2, 2))
    J = p1.jacobian(np.zeros(2 * p1.n), 1.0).tocsr()
    J.sort_indices()
    nod = np.concatenate([np.arange(p1.n)] * 2)

    def rank_main(c):
        with pytest.raises(PgxError):
            DirectSolver(J.indptr, J.indices, nod, p1.coords, leaf_nodes=64, device=0, comm=c)
        return True

    assert all(_run_ranks(pcomm.local_group(4), rank_main))


@pytest.mark.parametrize("R", [2, 4])
def test_example02_distributed_run_matches_single_gpu(require_gpu, R):
    from proximalgalerkin_amd import comm as pcomm
    from proximalgalerkin_amd import signorini as G

    n = (6, 6, 6)
    mesh = G.create_unit_cube(*n)
    mt, bcs = G.native_tags(mesh)
    it1, its1, x1, _ = G.solve_contact_problem(mesh, mt, bcs, verbose=False, return_solution=True)

    def rank_main(c):
        return G.solve_contact_problem(mesh, mt, bcs, verbose=False, return_solution=True, comm=c)

    for it, its, x, _ in _run_ranks(pcomm.local_group(R), rank_main):
        assert it == it1 and list(its) == list(its1)
        nu3 = 3 * mesh.geometry.shape[0]
        assert np.linalg.norm(x[:nu3] - x1[:nu3]) <= 1e-10 * np.linalg.norm(x1[:nu3])
    coords, cells = S.create_unit_cube_tets(*n)
    prob = S.SignoriniP1(coords, cells, S.boundary_facets_where(coords, cells, lambda c: np.isclose(c[:, 2], 0.0)),
                         np.flatnonzero(np.isclose(coords[:, 2], 1.0)))
    x_ref, it_ref, its_ref = S.solve_contact_problem(prob)
    assert list(its1) == list(its_ref)


@pytest.mark.parametrize("kind,degree", [("tet", 1), ("tet", 2), ("hex", 1)])
def test_example02_elements_are_partitioned_over_the_ranks(require_gpu, monkeypatch, kind, degree):
    """pgx_sg_create_dist: every rank assembles the elasticity blocks of ITS slab of cells (equal counts, together all cells), one
    all-reduce sums the constant matrix: the Jacobian and residual of every rank equal the single handle's to rounding (the slabs
    are summed in another order), and bitwise each other's.  PGX_SG_PARTITION=0 restores the replicated assembly of rounds 2-3."""
    from proximalgalerkin_amd import comm as pcomm
    from proximalgalerkin_amd import signorini as G

    mesh = G.create_unit_cube(5, 4, 7) if kind == "tet" else G.create_unit_cube_hex(4, 3, 6)
    mt, bcs = G.native_tags(mesh)
    contact = np.concatenate([mt.find(t) for t in bcs["contact"]])
    bcf = np.concatenate([mt.find(t) for t in bcs["displacement"]])

    def make(c):
        bv = np.unique(bcf.ravel()) if (degree == 1 and kind == "tet") else None
        p = G.SignoriniProblem(mesh, contact, bv, 2.0e4, 0.3, 0.0, -0.25, comm=c, degree=degree, bc_facets=bcf)
        rng = np.random.default_rng(1)
        x = 0.01 * rng.standard_normal(p.ndofs)
        p.set_alpha(0.7)
        p.set_prev(0.5 * x)
        F, _ = p.residual(x)
        J = p.jacobian(x)
        info = p.partition_info()
        p.close()
        return F, J, info

    F1, J1, info1 = make(None)
    assert info1[0] == info1[1]
    R = 4
    res = _run_ranks(pcomm.local_group(R), make)
    assert sum(r[2][0] for r in res) == info1[1] and all(0 < r[2][0] < info1[1] for r in res)
    for F, J, _ in res:
        assert np.array_equal(F, res[0][0]) and np.array_equal(J.data, res[0][1].data)
        assert np.linalg.norm(F - F1) <= 1e-13 * np.linalg.norm(F1)
        assert abs(J - J1).max() <= 1e-13 * abs(J1).max()
    monkeypatch.setenv("PGX_SG_PARTITION", "0")
    for F, J, info in _run_ranks(pcomm.local_group(2), make):
        assert info[0] == info[1] and np.array_equal(F, F1) and np.array_equal(J.data, J1.data)


def test_example06_distributed_run_matches_single_gpu(require_gpu):
    from proximalgalerkin_amd import comm as pcomm
    from proximalgalerkin_amd.gradient_constraint import solve_problem

    N = 12
    its1, d1, x1 = solve_problem(N, N, verbose=False, return_solution=True)

    def rank_main(c):
        return solve_problem(N, N, verbose=False, return_solution=True, comm=c)

    n2 = (2 * N + 1) ** 2
    for its, d, x in _run_ranks(pcomm.local_group(2), rank_main):
        assert list(its) == list(its1)
        assert np.linalg.norm(x[:n2] - x1[:n2]) <= 1e-10 * np.linalg.norm(x1[:n2])


@pytest.mark.parametrize("degree,N", [(2, 24), (1, 40)])
def test_example01_replicas_with_distributed_lu(require_gpu, degree, N):
    """pgx_create_lu_dist (BASELINE config 3: P2, multi-GPU): replicated handles, distributed LU preconditioner."""
    from proximalgalerkin_amd import comm as pcomm
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem

    def run(c):
        msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
        problem, sol, sol_k, alpha = setup_problem(msh, degree, lu_comm=c)
        hist = run_outer_loop(problem, sol, sol_k, alpha, 100, "double_exponential", 1e2, 1e-4)
        x = sol.x.array.copy()
        problem.close()
        return x, hist

    x1, h1 = run(None)
    coords, cells = O.create_rectangle(N, N)
    prob = O.ObstacleLagrange(coords, cells, degree)
    x_ref, h_ref = O.solve_problem(prob, 100, "double_exponential", 1e2, 1e-4)
    assert h1["Newton steps"] == h_ref["Newton steps"]
    for x, h in _run_ranks(pcomm.local_group(2), run):
        assert h["Newton steps"] == h_ref["Newton steps"]
        assert np.linalg.norm(x[: prob.n] - x_ref[: prob.n]) <= 1e-10 * np.linalg.norm(x_ref[: prob.n])
