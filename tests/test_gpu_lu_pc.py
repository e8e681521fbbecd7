"""Newton solves preconditioned by the sparse direct solver (pc_type pgx_lu; automatic for P2) against the CPU oracle:
identical Newton / proximal counts, final primal field <= 1e-10 relative L2 (BASELINE.json north_star)."""
import numpy as np
import pytest

from oracle import pg_oracle as O

pytestmark = pytest.mark.gpu
DOMAIN = ((-1.0, -1.0), (1.0, 1.0))
OPTS = {"ksp_type": "preonly", "pc_type": "pgx_lu", "ksp_error_if_not_converged": True,
        "snes_error_if_not_converged": True, "snes_linesearch_type": "none", "snes_rtol": 1e-6, "snes_max_it": 100}


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _run(N, degree, opts, scheme="double_exponential", alpha_max=1e2, tol=1e-4):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem

    msh = fem.create_rectangle(DOMAIN, (N, N))
    problem, sol, sol_k, alpha = setup_problem(msh, degree, petsc_options=opts)
    hist = run_outer_loop(problem, sol, sol_k, alpha, 100, scheme, alpha_max, tol)
    x = sol.x.array.copy()
    lin = problem.solver.ksp.getIterationNumber() if hasattr(problem.solver.ksp, "getIterationNumber") else None
    problem.close()
    return x, hist, lin


@pytest.mark.parametrize("N", [24, 64])
def test_p1_lu_preconditioner_matches_oracle(require_gpu, N):
    x, hist, _ = _run(N, 1, OPTS)
    coords, cells = O.create_rectangle(N, N)
    prob = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    x_ref, h_ref = O.solve_problem(prob, 100, "double_exponential", 1e2, 1e-4)
    assert hist["Newton steps"] == h_ref["Newton steps"]
    assert _rel(x[: prob.n], x_ref[: prob.n]) < 1e-10


def test_p1_lu_and_multigrid_agree(require_gpu):
    N = 128
    xl, hl, _ = _run(N, 1, OPTS)
    xm, hm, _ = _run(N, 1, dict(OPTS, pc_type="pgx_mg"))
    assert hl["Newton steps"] == hm["Newton steps"]
    n = len(xl) // 2
    assert _rel(xl[:n], xm[:n]) < 1e-10


@pytest.mark.parametrize("mode", ["auto", "pgx_mg", "fallback"])
def test_p2_patch_multigrid_matches_oracle(require_gpu, monkeypatch, mode):
    """P2, N = 48 (round 3): the two-level cycle with the vertex-star patch smoother (pgx_patch.hip) - forced (`pgx_mg`), as the
    automatic choice for degree 2 on a structured mesh (`auto`: patch multigrid first, sparse LU for a Newton solve in which it
    stagnates), and with the stagnation threshold lowered to 3 iterations so that the LU fallback path itself runs (`fallback`).
    Same Newton counts and primal field as the exact-Newton oracle in all three."""
    if mode == "fallback":
        monkeypatch.setenv("PGX_P2_FALLBACK_ITS", "3")
    N = 48
    x, hist, _ = _run(N, 2, dict(OPTS, pc_type="pgx_mg") if mode == "pgx_mg" else None)
    coords, cells = O.create_rectangle(N, N)
    prob = O.ObstacleLagrange(coords, cells, 2)
    x_ref, h_ref = O.solve_problem(prob, 100, "double_exponential", 1e2, 1e-4)
    assert hist["Newton steps"] == h_ref["Newton steps"]
    assert _rel(x[: prob.n], x_ref[: prob.n]) < 1e-10


def test_p2_patch_multigrid_iteration_counts_do_not_grow(require_gpu):
    """The point of the patch smoother: Krylov iterations per Newton step stay bounded as the mesh is refined (settings A, where no
    iterate overshoots: 8-11 per step at every size; the point-Jacobi cycle of rounds 1-2 needed 30-90 at 64^2-128^2)."""
    per_step = {}
    for N in (32, 128):
        from proximalgalerkin_amd import fem
        from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem

        msh = fem.create_rectangle(DOMAIN, (N, N))
        problem, sol, sol_k, alpha = setup_problem(msh, 2, petsc_options=dict(OPTS, pc_type="pgx_mg"))
        lin = []
        orig = problem.solve

        def solve(orig=orig, problem=problem, lin=lin):
            r = orig()
            lin.append(problem.solver.getLinearSolveIterations())
            return r

        problem.solve = solve
        hist = run_outer_loop(problem, sol, sol_k, alpha, 100, "constant", 1e5, 1e-6)
        problem.close()
        per_step[N] = max(k / n for k, n in zip(lin, hist["Newton steps"]))
    assert per_step[32] <= 14 and per_step[128] <= 14, per_step


def test_p2_sparse_lu_matches_oracle(require_gpu):
    """P2, N = 48 through the sparse LU (`pc_type pgx_lu`, the reference's literal choice and round 2's default for degree 2)."""
    N = 48
    x, hist, _ = _run(N, 2, OPTS)
    coords, cells = O.create_rectangle(N, N)
    prob = O.ObstacleLagrange(coords, cells, 2)
    x_ref, h_ref = O.solve_problem(prob, 100, "double_exponential", 1e2, 1e-4)
    assert hist["Newton steps"] == h_ref["Newton steps"]
    assert _rel(x[: prob.n], x_ref[: prob.n]) < 1e-10


def test_p2_through_the_subtree_sequenced_lu(require_gpu, monkeypatch):
    """The schedule BASELINE config 3 (2048^2 P2) runs with on one GPU - tree cut at depth 3, subtrees factorised one after
    the other, PGX_ND_CUT_GB - forced on a small P2 problem: same LVPP run as the oracle."""
    monkeypatch.setenv("PGX_ND_CUT_GB", "0")
    N = 32
    x, hist, _ = _run(N, 2, None)
    coords, cells = O.create_rectangle(N, N)
    prob = O.ObstacleLagrange(coords, cells, 2)
    x_ref, h_ref = O.solve_problem(prob, 100, "double_exponential", 1e2, 1e-4)
    assert hist["Newton steps"] == h_ref["Newton steps"]
    assert _rel(x[: prob.n], x_ref[: prob.n]) < 1e-10


def test_p2_n256_reproduces_the_oracles_newton_divergence(require_gpu):
    """P2, N = 256, settings B: the exact-Newton CPU oracle (SuperLU) itself ends with SNES_DIVERGED_DTOL (-9) at the
    alpha 16 -> 85 step (tests/golden/obstacle_p2_n256_settingsB_divergence.json, tools/p2_divergence_oracle.py) - the
    reference's undamped Newton overshoots there.  Parity means reproducing that: same Newton counts up to the step, same
    reason, same residual history."""
    import json
    import pathlib

    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import alpha_update, setup_problem

    z = json.loads((pathlib.Path(__file__).resolve().parent / "golden" / "obstacle_p2_n256_settingsB_divergence.json").read_text())
    msh = fem.create_rectangle(DOMAIN, (256, 256))
    problem, sol, sol_k, alpha = setup_problem(msh, 2, petsc_options={"snes_linesearch_type": "none", "snes_rtol": 1e-6,
                                                                     "snes_max_it": 100})
    problem.zero_state()
    alpha_k, counts, reason = 1, [], None
    for k in range(20):
        alpha.value, alpha_k = alpha_update("double_exponential", k, alpha_k, 1e2, current=alpha.value)
        problem.solve()
        reason = problem.solver.getConvergedReason()
        if reason <= 0:
            assert k == z["failed_outer_step_0based"] and reason == z["failed_reason"]
            assert problem.solver.getIterationNumber() == z["failed_after_its"]
            break
        counts.append(problem.solver.getIterationNumber())
        sol_k.x.assign_from(sol.x)
    assert reason == -9 and counts == z["newton_steps_converged"][: len(counts)] and len(counts) == 7
    problem.close()
