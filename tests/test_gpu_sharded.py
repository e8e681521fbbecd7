"""Sharded path (include/pgx.h "Sharded path") against the single-handle solve.

One-GPU boxes cannot run RCCL between ranks, so the R strips are driven by R host threads of this process through
the in-process transport (pgx_comm_local_group): the strip meshes, ghost-row bookkeeping, sharded V-cycle, owned-dof
Krylov space and packed reductions are exactly the code the RCCL launch runs; only the byte transport differs.
The RCCL transport itself is exercised with a one-rank communicator (library loading, communicator creation,
in-stream all-reduce)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DOMAIN = ((-1.0, -1.0), (1.0, 1.0))
OPTS = {"snes_linesearch_type": "none", "snes_rtol": 1e-6, "snes_max_it": 100, "snes_error_if_not_converged": True}


def _run_ranks(comms, fn):
    """fn(comm) on one thread per rank; re-raises the first failure."""
    out, err = [None] * len(comms), [None] * len(comms)

    def work(r):
        try:
            out[r] = fn(comms[r])
        except BaseException as e:  # noqa: BLE001 - reported below
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


def _global_solution(nx, ny, scheme, alpha_max, tol):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem

    msh = fem.create_rectangle(DOMAIN, (nx, ny))
    problem, sol, sol_k, alpha = setup_problem(msh, 1, petsc_options=OPTS)
    hist = run_outer_loop(problem, sol, sol_k, alpha, 100, scheme, alpha_max, tol)
    x = sol.x.array.copy()
    problem.close()
    return x, hist


@pytest.mark.parametrize("R,nx,ny,levels", [(2, 64, 128, 0), (3, 96, 96, 0), (4, 64, 256, 2), (2, 64, 64, 1)])
def test_sharded_solve_equals_single_handle(require_gpu, R, nx, ny, levels):
    from proximalgalerkin_amd import comm as pcomm
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem

    xg, hg = _global_solution(nx, ny, "double_exponential", 1e2, 1e-4)
    sx = nx + 1
    ng = sx * (ny + 1)

    def rank_main(c):
        msh = fem.create_rectangle(DOMAIN, (nx, ny), comm=c, dist_levels=levels)
        problem, sol, sol_k, alpha = setup_problem(msh, 1, petsc_options=OPTS)
        hist = run_outer_loop(problem, sol, sol_k, alpha, 100, "double_exponential", 1e2, 1e-4)
        x = sol.x.array.copy()
        part = msh.partition
        off, cnt = problem.owned_range()
        problem.close()
        return x, hist, part, off, cnt

    res = _run_ranks(pcomm.local_group(R), rank_main)
    u = np.full(ng, np.nan)
    psi = np.full(ng, np.nan)
    for x, hist, part, off, cnt in res:
        assert hist["Newton steps"] == hg["Newton steps"]  # same algebra -> same Newton and proximal counts
        for col in ("Energy", "Complementarity", "Dual Feasibility", "Primal increments", "Latent increments"):
            assert np.allclose(hist[col], hg[col], rtol=1e-8, atol=1e-12), col  # packed all-reduce of the six
        n = len(x) // 2
        assert off == (part.own0 - part.row0) * sx and cnt == part.nown * sx
        g0 = part.own0 * sx
        u[g0:g0 + cnt] = x[off:off + cnt]
        psi[g0:g0 + cnt] = x[n + off:n + off + cnt]
        # ghost entries of the returned local vector agree with the global field too (they were exchanged)
        lo = part.row0 * sx
        assert np.linalg.norm(x[:n] - xg[lo:lo + n]) <= 1e-10 * np.linalg.norm(xg[:ng])
    assert not np.isnan(u).any()  # the strips tile the mesh
    assert np.linalg.norm(u - xg[:ng]) <= 1e-10 * np.linalg.norm(xg[:ng])  # BASELINE.json north_star tolerance
    live = xg[ng:] > -50.0  # psi is ill-conditioned where exp(psi) = 0
    assert np.linalg.norm((psi - xg[ng:])[live]) <= 1e-7 * np.linalg.norm(xg[ng:][live])


def test_sharded_residual_spmv_and_observables_match_global(require_gpu):
    """Fine-grained calls on a random state: owned rows of the local residual / operator action equal the global
    rows (assembly of ghost cells is redundant, so no reverse ghost update is needed: lvpp/problem.py:66)."""
    from proximalgalerkin_amd import comm as pcomm
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import setup_problem

    R, nx, ny = 3, 48, 96
    sx, ng = nx + 1, (nx + 1) * (ny + 1)
    rng = np.random.default_rng(5)
    xg = rng.standard_normal(2 * ng) * 0.2
    xk = rng.standard_normal(2 * ng) * 0.2
    v = rng.standard_normal(2 * ng)
    msh = fem.create_rectangle(DOMAIN, (nx, ny))
    problem, sol, sol_k, alpha = setup_problem(msh, 1, petsc_options=OPTS)
    alpha.value = 3.0
    sol_k.x.array[:] = xk
    sol.x.array[:] = xg
    Fg, fn_g = problem.residual()
    obs_g = problem.observables()
    problem.assemble_jacobian()
    yg = problem.spmv(v)
    problem.close()

    def rank_main(c):
        m = fem.create_rectangle(DOMAIN, (nx, ny), comm=c)
        p, s, sk, a = setup_problem(m, 1, petsc_options=OPTS)
        part = m.partition
        n = m.num_vertices
        lo = part.row0 * sx
        loc = lambda z: np.concatenate([z[lo:lo + n], z[ng + lo:ng + lo + n]])  # noqa: E731
        a.value = 3.0
        xs = loc(xg)
        if part.rank > 0:
            xs[:sx] = 77.0  # stale ghost row on purpose: pgx_sync_ghosts must repair it from the owner
        sk.x.array[:] = loc(xk)
        s.x.array[:] = xs
        F, fn = p.residual()
        obs = p.observables()
        p.assemble_jacobian()
        y = p.spmv(loc(v))
        off, cnt = p.owned_range()
        p.close()
        own = np.r_[off:off + cnt]
        return (np.linalg.norm(F[own] - loc(Fg)[own]), np.linalg.norm(F[n + own] - loc(Fg)[n + own]), fn, obs,
                np.linalg.norm(y[own] - loc(yg)[own]), np.linalg.norm(y[n + own] - loc(yg)[n + own]))

    for eu, ep, fn, obs, yu, yp in _run_ranks(pcomm.local_group(R), rank_main):
        assert eu <= 1e-13 * np.linalg.norm(Fg) and ep <= 1e-13 * np.linalg.norm(Fg)
        assert abs(fn - fn_g) <= 1e-12 * fn_g  # one all-reduce over the owned entries
        assert np.allclose(obs, obs_g, rtol=1e-11, atol=1e-14)
        assert yu <= 1e-13 * np.linalg.norm(yg) and yp <= 1e-13 * np.linalg.norm(yg)


def test_rccl_transport_one_rank(require_gpu):
    """The RCCL communicator (dlopen of librccl, ncclCommInitRank, in-stream ncclAllReduce) on the one GPU we have:
    a one-strip 'sharded' solve must reproduce the plain solve."""
    from proximalgalerkin_amd import comm as pcomm
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem

    nx = ny = 64
    xg, hg = _global_solution(nx, ny, "double_exponential", 1e2, 1e-4)
    c = pcomm.rccl_single(0)
    msh = fem.create_rectangle(DOMAIN, (nx, ny), comm=c)
    assert msh.partition.nrows == ny + 1 and msh.partition.nown == ny + 1
    problem, sol, sol_k, alpha = setup_problem(msh, 1, petsc_options=OPTS)
    hist = run_outer_loop(problem, sol, sol_k, alpha, 100, "double_exponential", 1e2, 1e-4)
    n = msh.num_vertices
    assert hist["Newton steps"] == hg["Newton steps"]
    assert np.linalg.norm(sol.x.array[:n] - xg[:n]) <= 1e-10 * np.linalg.norm(xg[:n])
    problem.close()
    c.free()


def test_partition_errors_are_reported(require_gpu):
    from proximalgalerkin_amd import _lib
    from proximalgalerkin_amd import comm as pcomm
    from proximalgalerkin_amd import fem

    cs = pcomm.local_group(2)
    with pytest.raises(_lib.PgxError, match="divisible"):
        fem.create_rectangle(DOMAIN, (32, 33), comm=cs[0])
    with pytest.raises(_lib.PgxError, match="too thin|divisible"):
        fem.create_rectangle(DOMAIN, (32, 4), comm=cs[0])


def test_sharded_full_size_2048_on_8_strips(require_gpu):
    """BASELINE.json config 2 cut the way the 8-GPU launch cuts it (8 strips of 256 vertex rows, 3 distributed
    levels): same Newton counts as the single-handle solve, primal field within 1e-10."""
    from proximalgalerkin_amd import comm as pcomm
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem

    N, R = 2048, 8
    sx, ng = N + 1, (N + 1) ** 2

    def solve(c):
        msh = fem.create_rectangle(DOMAIN, (N, N), comm=c)
        problem, sol, sol_k, alpha = setup_problem(msh, 1)
        hist = run_outer_loop(problem, sol, sol_k, alpha, 500, "double_exponential", 1e2, 1e-4)
        x = sol.x.array.copy()
        rng = problem.owned_range()
        problem.close()
        return x, hist, msh.partition, rng

    xg, hg, _, _ = solve(None)
    u = np.full(ng, np.nan)
    for x, hist, part, (off, cnt) in _run_ranks(pcomm.local_group(R), solve):
        assert part.dist_levels == 3 and part.nown in (256, 257)
        assert hist["Newton steps"] == hg["Newton steps"]
        u[part.own0 * sx:part.own0 * sx + cnt] = x[off:off + cnt]
    assert np.linalg.norm(u - xg[:ng]) <= 1e-10 * np.linalg.norm(xg[:ng])


def test_bench_sharded_branch_runs_over_rccl_with_one_rank(require_gpu):
    """bench.py's N>1 branch (torch.distributed init, ncclUniqueId broadcast, RCCL communicator, strip mesh, collective
    solve, strong-scaling JSON) executed end to end with ONE rank - all a one-GPU box can host (RCCL refuses two ranks on
    one device)."""
    import json
    import os
    import pathlib
    import socket
    import subprocess
    import sys

    root = pathlib.Path(__file__).resolve().parents[1]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BENCH_FORCE_SHARDED="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port))
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "1", "--cells", "256", "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--watchdog", "300"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["scaling"] == "strong" and d["config"]["parallelism"].startswith("sharded: ONE 256x256 solve on 1 strips")
    assert d["config"]["mixed_unknowns"] == 2 * 257 * 257
    assert d["config"]["newton_iterations_per_step"] == 19 and d["value"] > 0  # 256^2 settings B: 19 Newton steps
    assert d["roofline"]["traffic"] is None and "cpu_baseline" not in d


@pytest.mark.parametrize("R,nx,ny,levels,scheme", [(2, 32, 64, 1, "constant"), (3, 48, 96, 0, "constant"),
                                                   (4, 32, 128, 2, "double_exponential")])
def test_sharded_p2_solve_equals_single_handle(require_gpu, R, nx, ny, levels, scheme):
    """P2 (`obstacle_pg.py -p 2`, BASELINE config 3) on R strips (round 3): vertex dofs by rows, the edge dofs of a vertex row as one
    contiguous block owned with it; halo exchange of vertex rows AND edge blocks; two-level cycle with the vertex-star patch
    smoother on the strip, the sharded P1 hierarchy below.  Same Newton / proximal counts and the same primal field as the
    single-handle solve on the whole mesh; observables through ONE packed all-reduce."""
    from proximalgalerkin_amd import comm as pcomm
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem

    a_max, tol = (1e5, 1e-6) if scheme == "constant" else (1e2, 1e-4)
    opts = dict(OPTS, pc_type="pgx_mg")  # the single handle takes the same preconditioner family (no LU fallback on either side)
    msh = fem.create_rectangle(DOMAIN, (nx, ny))
    problem, sol, sol_k, alpha = setup_problem(msh, 2, petsc_options=opts)
    hg = run_outer_loop(problem, sol, sol_k, alpha, 100, scheme, a_max, tol)
    xg = sol.x.array.copy()
    ndg = sol.function_space.block_size
    problem.close()
    sx, eb = nx + 1, 3 * nx + 1
    nvg = sx * (ny + 1)

    def rank_main(c):
        m = fem.create_rectangle(DOMAIN, (nx, ny), comm=c, dist_levels=levels)
        pr, s, sk, al = setup_problem(m, 2, petsc_options=opts)
        hist = run_outer_loop(pr, s, sk, al, 100, scheme, a_max, tol)
        x = s.x.array.copy()
        # round 4: the strips apply the P2 operator through the structured kernel too (pgx_p2st.hip) wherever a strip has an interior
        st = pr.p2_stencil_info()
        assert (st[0] == 2 and pr.spmv_select() == 3) or min(nx, m.num_vertices // (nx + 1) - 1) < 8, st
        out = (x, hist, m.partition, pr.owned_range(), pr.owned_edge_range(), s.function_space.block_size, m.num_vertices)
        pr.close()
        return out

    res = _run_ranks(pcomm.local_group(R), rank_main)
    u = np.full(ndg, np.nan)
    for x, hist, part, (voff, vcnt), (eoff, ecnt), nd, nv in res:
        assert hist["Newton steps"] == hg["Newton steps"]
        for col in ("Energy", "Primal increments", "Latent increments"):
            assert np.allclose(hist[col], hg[col], rtol=1e-7, atol=1e-12), col
        assert voff == (part.own0 - part.row0) * sx and vcnt == part.nown * sx
        assert eoff == nv + (part.own0 - part.row0) * eb
        u[part.own0 * sx: part.own0 * sx + vcnt] = x[voff: voff + vcnt]
        ge = nvg + part.own0 * eb  # global edge block of the first owned row
        u[ge: ge + ecnt] = x[eoff: eoff + ecnt]
    assert not np.isnan(u).any()  # the owned ranges of the ranks tile the global dof set
    assert np.linalg.norm(u - xg[:ndg]) <= 1e-10 * np.linalg.norm(xg[:ndg])


def test_sharded_p2_config3_ci_point_512_settings_b_on_4_strips(require_gpu):
    """BASELINE config 3's CI-settings point (512^2 P2, settings B = compare_all.py:80-87) in its sharded form.
    The single handle takes the sparse LU for the one Newton solve in which the patch cycle stagnates (proximal step 7: the iterate
    undamped Newton overshot); a sharded handle has no LU, so its Krylov solver simply keeps going on the un-restarted basis
    (obstacle_pg.py:128-139 asks for an exact solve, not for a particular solver).  Same Newton counts as the nested-dissection
    oracle's golden of that run, and the golden's primal field."""
    import pathlib

    from proximalgalerkin_amd import comm as pcomm
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem

    gold = np.load(pathlib.Path(__file__).parent / "golden" / "obstacle_p2_n512_settingsB_nd.npz")
    counts = [int(c) for c in gold["hist_Newton_steps"]]
    assert counts == [5, 4, 3, 2, 1, 1, 4, 1]
    N, R = 512, 4
    sx, eb = N + 1, 3 * N + 1

    def rank_main(c):
        m = fem.create_rectangle(DOMAIN, (N, N), comm=c, dist_levels=0)
        pr, s, sk, al = setup_problem(m, 2, petsc_options=OPTS)
        hist = run_outer_loop(pr, s, sk, al, 500, "double_exponential", 1e2, 1e-4)
        out = (s.x.array.copy(), hist, m.partition, pr.owned_range(), m.num_vertices)
        pr.close()
        return out

    res = _run_ranks(pcomm.local_group(R), rank_main)
    nvg = sx * (N + 1)
    uv = np.full(nvg, np.nan)
    for x, hist, part, (voff, vcnt), nv in res:
        assert hist["Newton steps"] == counts, hist["Newton steps"]
        uv[part.own0 * sx: part.own0 * sx + vcnt] = x[voff: voff + vcnt]
    assert not np.isnan(uv).any()
    # the golden's fingerprint of the vertex values of u (tests/test_gpu_golden.py): sub-lattice + block sums of every vertex
    stride = int(gold["stride"])
    grid = uv.reshape(N + 1, N + 1)
    nb = N // stride
    sample, blocksum = grid[::stride, ::stride], grid[: nb * stride, : nb * stride].reshape(nb, stride, nb, stride).sum(axis=(1, 3))
    for got, ref in ((sample, gold["u_sample"]), (blocksum, gold["u_blocksum"])):
        assert np.linalg.norm(got - ref) <= 1e-10 * np.linalg.norm(ref)
