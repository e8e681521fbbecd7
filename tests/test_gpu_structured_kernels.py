"""The structured-mesh kernels of round 2 against the general kernels they replace on uniform right-diagonal meshes, through the C ABI:

* `k_st_spmv_r` (matrix-free operator apply, pgx_spmv_select kind 1) vs `k_bspmv_stream` (block-CSR, kind 0) vs the generic stencil
  kernel (kind 2) on the same Jacobian and vectors;
* `k_resid_fill_grid` (LDS-staged element blocks) vs `k_resid_fill_p1` (row-parallel), selected per handle with PGX_RESID_GRID:
  residual, D(psi) entries, and a complete LVPP run.

Meshes are chosen so that tiles are cut by the boundary in both directions (sizes that are no multiples of the 62 x 24 / 32 x 12 tiles),
with non-square grids and Dirichlet data; tolerance 1e-13 relative (the kernels differ in summation order only).  The comparison with the
ORACLE of the same quantities is test_gpu_parity.py / test_gpu_golden.py, which run the structured kernels by default."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _state(n2, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n2) * 0.2
    x[n2 // 2:] -= 4.0 * np.abs(rng.standard_normal(n2 // 2))  # psi mostly negative, a few orders of magnitude of exp(psi)
    return x, rng.standard_normal(n2) * 0.2


@pytest.mark.parametrize("cells", [(64, 64), (96, 48), (200, 72), (130, 260)])
def test_matrix_free_apply_equals_block_csr(require_gpu, cells):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import setup_problem

    msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), cells)
    problem, sol, sol_k, alpha = setup_problem(msh)
    n2 = sol.function_space.num_dofs
    x, xk = _state(n2, 1)
    sol_k.x.array[:] = xk
    alpha.value = 2.5
    problem.assemble_jacobian(x)
    assert problem.spmv_select() == 1  # the default on a structured mesh
    rng = np.random.default_rng(2)
    for _ in range(2):
        v = rng.standard_normal(n2)
        y = {}
        for kind in (1, 0, 2):
            assert problem.spmv_select(kind) == kind
            y[kind] = problem.spmv(v)
        problem.spmv_select(1)
        scale = np.abs(y[0]).max()
        assert np.abs(y[1] - y[0]).max() <= 1e-13 * scale, np.abs(y[1] - y[0]).max() / scale
        assert np.abs(y[2] - y[0]).max() <= 1e-13 * scale
        assert np.array_equal(problem.spmv(v), y[1])  # and bitwise reproducible
    problem.close()


@pytest.mark.parametrize("cells", [(64, 64), (100, 36), (72, 150)])
def test_element_block_kernel_equals_row_parallel_kernel(require_gpu, cells, monkeypatch):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import setup_problem

    msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), cells)
    out = {}
    for grid in ("1", "0"):
        monkeypatch.setenv("PGX_RESID_GRID", grid)  # read at create
        problem, sol, sol_k, alpha = setup_problem(msh)
        n2 = sol.function_space.num_dofs
        x, xk = _state(n2, 5)
        sol_k.x.array[:] = xk
        alpha.value = 7.0
        F, fn = problem.residual(x)
        problem.assemble_jacobian(x)
        blocks = problem.export_blocks()
        F2, _ = problem.residual(x)
        assert np.array_equal(F, F2)
        out[grid] = (F, fn, blocks[4], problem.spmv(xk))
        problem.close()
    (F1, n1, D1, y1), (F0, n0, D0, y0) = out["1"], out["0"]
    assert np.abs(F1 - F0).max() <= 1e-13 * np.abs(F0).max()
    assert abs(n1 - n0) <= 1e-13 * n0
    assert np.abs(D1 - D0).max() <= 1e-13 * np.abs(D0).max()
    assert np.array_equal(D1 == 0.0, D0 == 0.0)  # exp(psi) underflows in the same entries
    assert np.abs(y1 - y0).max() <= 1e-13 * np.abs(y0).max()


def test_full_run_is_the_same_with_either_kernel_family(require_gpu, monkeypatch):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem

    msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (160, 160))
    res = {}
    for fam, env in (("structured", {"PGX_RESID_GRID": "1", "PGX_SPMV_STENCIL": "1"}), ("general", {"PGX_RESID_GRID": "0", "PGX_SPMV_STENCIL": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        problem, sol, sol_k, alpha = setup_problem(msh)
        assert problem.spmv_select() == (1 if fam == "structured" else 0)
        hist = run_outer_loop(problem, sol, sol_k, alpha, 500, "double_exponential", 1e2, 1e-4)
        assert problem.solver.getConvergedReason() > 0
        res[fam] = (hist["Newton steps"], sol.x.array.copy())
        problem.close()
    assert res["structured"][0] == res["general"][0]
    a, b = res["structured"][1], res["general"][1]
    n = a.size // 2
    assert np.linalg.norm(a[:n] - b[:n]) <= 1e-10 * np.linalg.norm(b[:n])
