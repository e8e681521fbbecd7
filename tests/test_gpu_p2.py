"""P2 (obstacle_pg.py -p 2) HIP path vs the CPU oracle, through the C ABI.
Tolerances as in test_gpu_parity.py: kernels 1e-12, full run identical counts and u <= 1e-10 relative L2."""
import numpy as np
import pytest

from oracle import pg_oracle as O

pytestmark = pytest.mark.gpu
DOMAIN = ((-1.0, -1.0), (1.0, 1.0))


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _setup(N, M=None):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import setup_problem

    M = N if M is None else M
    msh = fem.create_rectangle(DOMAIN, (N, M))
    problem, sol, sol_k, alpha = setup_problem(msh, 2)
    coords, cells = O.create_rectangle(N, M)
    prob = O.ObstacleLagrange(coords, cells, 2)
    assert sol.function_space.block_size == prob.n
    return problem, sol, sol_k, alpha, prob


def _iterates(n2, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n2) * 0.1
    xk = rng.standard_normal(n2) * 0.1
    n = n2 // 2
    x[n:] = -np.abs(rng.standard_normal(n)) * np.where(rng.random(n) < 0.3, 200.0, 2.0)
    return x, xk


@pytest.mark.parametrize("N,M", [(4, 4), (9, 6), (32, 32)])
def test_p2_kernels_match_oracle(require_gpu, N, M):
    problem, sol, sol_k, alpha, prob = _setup(N, M)
    x, xk = _iterates(2 * prob.n, 3)
    alpha.value = 1.75
    sol_k.x.array[:] = xk
    F, fn = problem.residual(x)
    Fr = prob.residual(x, xk, 1.75)
    assert _rel(F, Fr) < 1e-12 and abs(fn - np.linalg.norm(Fr)) < 1e-12 * np.linalg.norm(Fr)
    problem.assemble_jacobian(x)
    rowptr, col, K, Mv, D = problem.export_blocks()
    assert np.array_equal(rowptr, prob.indptr_s.astype(np.int32)) and np.array_equal(col, prob.indices_s)
    assert _rel(K, prob.K.data) < 1e-12 and _rel(Mv, prob.M.data) < 1e-12
    assert _rel(D, prob.jacobian_blocks(x)) < 1e-12
    J = prob.jacobian(x, 1.75)
    v = np.random.default_rng(5).standard_normal(2 * prob.n)
    assert _rel(problem.spmv(v), J @ v) < 1e-12
    sol.x.array[:] = np.clip(x, -40, None)
    assert np.allclose(problem.observables(), prob.observables(np.clip(x, -40, None), xk, 1.75), rtol=1e-12, atol=1e-14)
    problem.close()


@pytest.mark.parametrize("scheme,alpha_max,tol,N", [("double_exponential", 1e2, 1e-4, 32), ("constant", 1e5, 1e-6, 16)])
def test_p2_full_run_matches_oracle(require_gpu, scheme, alpha_max, tol, N):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import COLUMNS, solve_problem

    msh = fem.create_rectangle(DOMAIN, (N, N))
    sol, newton, hist = solve_problem(msh, 2, 100, scheme, alpha_max, tol, verbose=False, return_history=True)
    coords, cells = O.create_rectangle(N, N)
    prob = O.ObstacleLagrange(coords, cells, 2)
    x_ref, h_ref = O.solve_problem(prob, 100, scheme, alpha_max, tol)
    assert hist["Newton steps"] == h_ref["Newton steps"]
    n = prob.n
    assert _rel(sol.x.array[:n], x_ref[:n]) < 1e-10
    for c in COLUMNS:
        assert np.allclose(hist[c], h_ref[c], rtol=1e-7, atol=1e-11), c


def test_p2_unstructured_numbering(require_gpu):
    """permuted vertex ids -> general mesh: P2 assembly/SpMV must not depend on structure; the solve then uses
    the P2 smoother + single-level P1 smoother as preconditioner"""
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import setup_problem

    N = 8
    coords, cells = O.create_rectangle(N, N)
    rng = np.random.default_rng(9)
    perm = rng.permutation(len(coords))
    inv = np.argsort(perm)
    msh = fem.Mesh(coords[inv], perm[cells].astype(np.int32)[rng.permutation(len(cells))])
    problem, sol, sol_k, alpha = setup_problem(msh, 2)
    prob = O.ObstacleLagrange(msh.geometry, msh.cells, 2)
    x, xk = _iterates(2 * prob.n, 6)
    sol_k.x.array[:] = xk
    F, _ = problem.residual(x)
    assert _rel(F, prob.residual(x, xk, 1.0)) < 1e-12
    problem.assemble_jacobian(x)
    v = rng.standard_normal(2 * prob.n)
    assert _rel(problem.spmv(v), prob.jacobian(x, 1.0) @ v) < 1e-12
    sol.x.array[:] = 0.0
    sol_k.x.array[:] = 0.0
    problem.solve()
    z = np.zeros(2 * prob.n)
    x_ref, r_ref, its_ref = O.newton_solve(prob, z, z, 1.0, O.SnesOptions(rtol=1e-6, max_it=100))
    assert problem.solver.getIterationNumber() == its_ref
    assert _rel(sol.x.array[:prob.n], x_ref[:prob.n]) < 1e-9
    problem.close()


@pytest.mark.parametrize("N,M", [(40, 40), (96, 33), (257, 5)])
def test_p2_nnz_balanced_spmv_matches_oracle_and_the_row_block_kernel(require_gpu, monkeypatch, N, M):
    """k_bspmv_bal (blocks of at most 2048 entries; the sparse boundary rows make blocks of exactly 256 rows, the case in which the
    kernel needs one row pointer more than it has threads) against the oracle's J @ v and against the 256-rows-per-block kernel."""
    out = {}
    for bal in ("1", "0"):
        monkeypatch.setenv("PGX_SPMV_BAL", bal)
        problem, sol, sol_k, alpha, prob = _setup(N, M)
        x, xk = _iterates(2 * prob.n, 11)
        alpha.value = 3.5
        sol_k.x.array[:] = xk
        problem.assemble_jacobian(x)
        v = np.random.default_rng(7).standard_normal(2 * prob.n)
        out[bal] = problem.spmv(v)
        if bal == "1":
            assert _rel(out[bal], prob.jacobian(x, 3.5) @ v) < 1e-12
        problem.close()
    assert _rel(out["1"], out["0"]) < 1e-14


@pytest.mark.parametrize("N,M", [(40, 40), (96, 33)])
def test_p2_spmv_km_dictionary_equals_the_streamed_values(require_gpu, monkeypatch, N, M):
    """k_bspmv_bal<true> (K and M read through the one-byte dictionary of their distinct pairs on a uniform mesh) against the same
    kernel streaming K and M, and against the oracle's J @ v."""
    out = {}
    monkeypatch.setenv("PGX_P2_STENCIL", "0")  # the CSR kernel itself (the structured apply of the interior reads no K / M stream)
    for d in ("1", "0"):
        monkeypatch.setenv("PGX_SPMV_DICT", d)
        problem, sol, sol_k, alpha, prob = _setup(N, M)
        x, xk = _iterates(2 * prob.n, 5)
        alpha.value = 0.7
        sol_k.x.array[:] = xk
        problem.assemble_jacobian(x)
        v = np.random.default_rng(3).standard_normal(2 * prob.n)
        out[d] = problem.spmv(v)
        problem.close()
    assert _rel(out["1"], out["0"]) < 1e-11  # entries rounded to 2^-40 of the largest one (build_km_dictionary)
    assert _rel(out["1"], prob.jacobian(x, 0.7) @ v) < 1e-11


@pytest.mark.parametrize("N,M", [(48, 48), (96, 20)])
def test_p2_patch_inverses_in_symmetric_packing_equal_the_full_rows(require_gpu, monkeypatch, N, M):
    """The patch smoother streams its float inverses in symmetric packing (chunks at or left of a row's diagonal block, the group
    rebuilds the matrix in LDS): the same run with the full rows (PGX_P2_PATCH_SYM=0) and with double inverses takes the same Newton
    steps and ends in the same primal field (the two float forms differ by the asymmetry of a float-rounded inverse)."""
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import solve_problem

    out = {}
    for name, env in (("sym", {}), ("full", {"PGX_P2_PATCH_SYM": "0"}), ("double", {"PGX_P2_PATCH_F32": "0"}),
                      ("bf16", {"PGX_P2_PATCH_F32": "2"})):  # opt-in: bfloat16 entries in the symmetric packing
        for k in ("PGX_P2_PATCH_SYM", "PGX_P2_PATCH_F32"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        msh = fem.create_rectangle(DOMAIN, (N, M))
        sol, newton, hist = solve_problem(msh, 2, 100, "double_exponential", 1e2, 1e-4, verbose=False, return_history=True)
        out[name] = (hist["Newton steps"], sol.x.array[: sol.function_space.block_size].copy())
    for other in ("full", "double", "bf16"):
        assert out["sym"][0] == out[other][0]
        assert _rel(out["sym"][1], out[other][1]) < 1e-10


@pytest.mark.parametrize("N,M", [(8, 8), (9, 6), (40, 40), (96, 33), (257, 9), (64, 130)])
def test_p2_structured_apply_equals_the_csr_kernel_and_the_oracle(require_gpu, monkeypatch, N, M):
    """pgx_p2st.hip: interior groups through the table-driven kernel (no column indices, K / M as constants, D(psi) from its
    structure-of-arrays copy), the frame through the CSR form - against k_bspmv_bal on the same Jacobian and vectors (1e-13: another
    summation order) and against the oracle's J @ v; at two alpha (the table's alpha K is refreshed per call) and after a SECOND
    Jacobian (the SoA copy is refreshed).  9 x 6 has no interior rectangle: the CSR kernel serves it alone."""
    out = {}
    for st in ("1", "0"):
        monkeypatch.setenv("PGX_P2_STENCIL", st)
        problem, sol, sol_k, alpha, prob = _setup(N, M)
        res = []
        for seed, a in ((11, 3.5), (12, 0.25)):
            x, xk = _iterates(2 * prob.n, seed)
            alpha.value = a
            sol_k.x.array[:] = xk
            problem.assemble_jacobian(x)
            v = np.random.default_rng(seed).standard_normal(2 * prob.n)
            y = problem.spmv(v)
            if st == "1":
                info = problem.p2_stencil_info()  # in use wherever the mesh has an interior: groups 2 ... n - 2 in both directions
                assert info == ((2, 2, N - 3, 2, M - 3) if min(N, M) >= 8 else (0, 0, 0, 0, 0)), info
                assert problem.spmv_select() == (3 if min(N, M) >= 8 else 0)
                assert _rel(y, prob.jacobian(x, a) @ v) < 1e-12
                assert np.array_equal(problem.spmv(v), y)  # bitwise reproducible
            res.append(y)
        out[st] = res
        problem.close()
    for a, b in zip(out["1"], out["0"]):
        assert _rel(a, b) < 1e-13


def test_p2_patch_sweep_microbenchmark(require_gpu):
    """pgx_smoother_bench on a P2 handle times the patch sweep (bench.py's `roofline_dominant` of --degree 2): positive time, the
    byte count of its streams (512 B of packed float inverses per patch dominate)."""
    problem, sol, sol_k, alpha, prob = _setup(64, 64)
    x, xk = _iterates(2 * prob.n, 4)
    sol_k.x.array[:] = xk
    problem.assemble_jacobian(x)
    ms, by = problem.smoother_bench(reps=3)
    nv = 65 * 65
    assert ms > 0 and 512 * nv < by < 1100 * nv
    problem.close()
