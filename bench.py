#!/usr/bin/env python3
"""bench.py - LVPP / proximal-Galerkin Newton loop on the 2048x2048 P1 obstacle problem.

    python bench.py --gpus 1 --steps K --warmup W

A "step" is ONE complete proximal-point solve of the obstacle problem (the loop of
/root/reference/examples/01_obstacle_problem/obstacle_pg.py:173-227: every proximal iteration, each
with its Newton iterations, observables and stopping test) from the zero state, on a mesh that is
already resident in HBM.  `value` = Newton iterations per second over the timed K steps
(BASELINE.json metric "proximal-Newton iterations/sec"); proximal iterations/s is reported beside it.

`--workload ex06` / `--workload ex02` time BASELINE configs 4 and 5 instead (example 06 at --cells 1024, example 02 on
--cells 70 cubes of 6 tetrahedra): same contract, `roofline` then describes the fp64-MFMA GEMM of the sparse LU
(bound "mfma"); with N>1 example 02 runs ONE solve with the factorisation distributed over the ranks.

N>1: one process per GPU under torch.distributed.run.  The SAME problem is cut into N horizontal strips (sharded
path of include/pgx.h: RCCL halo exchange of ghost vertex rows + packed all-reduces), so the numbers at N = 1, 2, 4, 8
are a strong-scaling series of one fixed workload; `--replicas` runs N independent solves instead (and says so).
DESIGN.md section 7.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling

SETTINGS = {
    # script defaults of obstacle_pg.py:295,304,320
    "A": dict(alpha_scheme="constant", alpha_max=1e5, tol_exit=1e-6, max_outer=100),
    # what the reference's CI runs: compare_all.py:80-87
    "B": dict(alpha_scheme="double_exponential", alpha_max=1e2, tol_exit=1e-4, max_outer=500),
}


def _ladder(name):
    """Committed CPU ladder (profiles/<name>, written in the build container by tools/make_golden_large.py /
    tools/cpu_ladder.py): seconds per Newton step of the oracle at a series of mesh sizes + the fitted exponent."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            return json.load(fh)
    except OSError:
        return None


def extrapolate(v_measured, n_measured, n_workload, ladder, size_key="N"):
    """Newton it/s of the CPU oracle at the benchmarked mesh, from the live measurement at n_measured and the exponent p of
    the committed ladder (time per Newton step = c N^p).  Labelled as an extrapolation wherever it is printed."""
    if not ladder or "fit" not in ladder:
        return None
    p = ladder["fit"]["p"]
    pts = ladder["points"]
    return {"value": v_measured * (n_measured / n_workload) ** p, "unit": "Newton iterations/s", "mesh": n_workload,
            "method": f"EXTRAPOLATED, not measured: seconds per Newton step fitted as c*N^p with p = {p:.2f} on the committed "
                      f"ladder ({size_key} = {', '.join(str(q[size_key]) for q in pts)}; largest measured point "
                      f"{pts[-1]['s_per_newton_step']:.1f} s per step at {size_key} = {pts[-1][size_key]}), applied to the live "
                      f"measurement at {size_key} = {n_measured}"}


def cpu_baseline(n_sample: int, settings: dict, budget_s: float = 20.0, threads=(1,), degree: int = 1, min_steps: int = 4):
    """cpu_baseline_inprocess in a CHILD process (`bench.py --cpu-leg <json>`): the all-cores leg forks worker processes for the
    subtrees of the dissection, and the benchmark process itself holds a HIP context by now - the child never touches the GPU (a
    forked child of a GPU process is allowed to, but its forks then copy the driver's mappings: measured, the tree-parallel
    factorisation gained nothing when forked from the GPU process and 3.4 x from a clean one).  The child is started as a child
    process and waited for; nothing is exec'ed over this one."""
    import subprocess

    arg = json.dumps({"n_sample": n_sample, "settings": settings, "budget_s": budget_s, "threads": list(threads), "degree": degree,
                      "min_steps": min_steps})
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")  # the oracle is numpy / scipy: no device
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg", arg], capture_output=True, text=True, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("[")]
    if r.returncode or not lines:
        raise RuntimeError(f"cpu_baseline child failed (rc {r.returncode}): {r.stderr.strip()[-400:]}")
    return [tuple(x) for x in json.loads(lines[-1])]


def cpu_baseline_inprocess(n_sample: int, settings: dict, budget_s: float = 20.0, threads=(1,), degree: int = 1, min_steps: int = 4):
    """Oracle (numpy assembly + exact Newton with the nested-dissection multifrontal LU of oracle/nd_lu.py - the ordering class and
    the BLAS-3 structure a CPU user gets from `pc_type lu` / MUMPS, obstacle_pg.py:129-131) timed on this host: the SAME LVPP run on
    an n_sample^2 mesh of Lagrange degree `degree`, stopped after the first Newton step that ends beyond budget_s AND after at
    least `min_steps` Newton steps.  One timed run per entry of `threads` (BLAS threads; 1 = the reference's OMP_NUM_THREADS=1) on
    the same setup.  Timed region = the loop of obstacle_pg.py:173-227 (residual + Jacobian assembly, numeric factorisation,
    solves with iterative refinement, observables); untimed, like the GPU side's setup: mesh / pattern construction and the
    symbolic analysis of the pattern (done once per mesh by any direct solver).  Returns [(value, steps, seconds, detail), ...]."""
    from threadpoolctl import threadpool_limits

    from oracle import nd_lu as ND  # CPU baseline leg only
    from oracle import pg_oracle as O  # CPU baseline leg only

    t_setup = time.perf_counter()
    coords, cells = O.create_rectangle(n_sample, n_sample)
    if degree == 1:
        prob = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(n_sample, n_sample))
    else:
        prob = O.ObstacleLagrange(coords, cells, degree)
    ls = ND.NDLinearSolve(*ND.nodes_of_problem(prob))
    ls.nd = ND.NDLU(prob.jacobian(np.zeros(2 * prob.n), 1.0), ls.node_of_dof, ls.node_coords, ls.leaf_nodes)
    ls.nd.aoff = np.concatenate(([0], np.cumsum(ls.nd.p * ls.nd.p + 2 * ls.nd.p * ls.nd.b)))
    ls.nd.arena = np.ones(int(ls.nd.aoff[-1]))  # factor storage allocated and touched before the clock starts
    t_setup = time.perf_counter() - t_setup
    runs = []
    for nthreads in threads:
        x = np.zeros(2 * prob.n)
        xk = x.copy()
        sched = O.AlphaSchedule(settings["alpha_scheme"], settings["alpha_max"])
        ND.MAX_THREADS = min(nthreads, 16)  # (the top fronts of a 2-D dissection do not feed more; 64 threads of a shared host thrash)
        # more than one thread: TREE-PARALLEL factorisation - forked workers on the subtrees of the dissection (one BLAS thread each),
        # the levels above them with all threads (oracle/nd_lu.py NDLU.factor(workers=)): threaded BLAS alone cannot feed the cores
        # from the small fronts of a 2-D dissection (round 4 measured it SLOWER than one thread)
        ls.workers = min(nthreads, 16) if nthreads > 1 else 0
        if ls.workers:
            ls.nd.prepare_parallel(ls.workers)
        ls.t_factor = ls.t_solve = 0.0
        ls.n_factor = 0
        steps, t0 = 0, time.perf_counter()

        def over():
            return time.perf_counter() - t0 > budget_s and steps >= min_steps

        with threadpool_limits(min(nthreads, 16)):
            for k in range(settings["max_outer"]):
                alpha = sched.update(k)
                F = prob.residual(x, xk, alpha)
                f0 = np.linalg.norm(F)
                for _ in range(100):
                    x = x + ls(prob.jacobian(x, alpha), -F)
                    steps += 1
                    F = prob.residual(x, xk, alpha)
                    if np.linalg.norm(F) <= 1e-6 * f0 or over():
                        break
                if over() or prob.observables(x, xk, alpha)[4] < settings["tol_exit"]:
                    break
                xk = x.copy()
        dt = time.perf_counter() - t0
        ND.MAX_THREADS = 0
        detail = {"setup_s_untimed": t_setup, "symbolic_s": ls.nd.symbolic_s, "factor_s": ls.t_factor, "solve_refine_s": ls.t_solve,
                  "assembly_and_rest_s": dt - ls.t_factor - ls.t_solve, "factor_gflops": ls.nd.flops / 1e9,
                  "factor_gflops_per_s": ls.nd.flops * ls.n_factor / max(ls.t_factor, 1e-9) / 1e9,
                  "factor_storage_GB": 8e-9 * ls.nd.factor_entries}
        runs.append((steps / dt, steps, dt, detail))
    return runs


def host_info():
    model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    return {"host_cores": os.cpu_count(), "host_cores_usable": usable, "host_cpu": model}


FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix peak (dense)


def bench_lu_workload(args, rank, world, local_rank, dist, backend):
    """BASELINE configs 4 (ex 06, gradient constraint) and 5 (ex 02, Signorini): a step = one full LVPP solve from the
    zero state with the reference's default settings; the linear solves are the sparse LU of include/pgx_nd.h."""
    import torch

    comm = None
    if world > 1:
        comm = make_comm(local_rank)
    t_setup = time.perf_counter()
    if args.workload == "ex06":
        from proximalgalerkin_amd import fem
        from proximalgalerkin_amd.gradient_constraint import GradientConstraintProblem, f_default, phi_default

        N = args.n if args.n != 2048 else 1024
        problem = GradientConstraintProblem(fem.create_unit_square(N, N), phi_default, f_default, device=local_rank, comm=comm)
        workload = (f"examples/06_gradient_constraints: {N}x{N} unit square, primal P2 / latent vector-P1 ({problem.ndofs} unknowns), "
                    "phi = 0.1+0.2x+0.4y, f = 15 sin^2(pi x), alpha doubling from 1, tol 1e-8, SNES atol=rtol=stol=1e-9")

        def one_step():
            problem.set_state(np.zeros(problem.ndofs))
            problem.set_prev(np.zeros(problem.ndofs))
            its = []
            for i in range(25):
                problem.set_alpha(2.0**i)
                its.append(problem.solve()[1])
                if problem.l2_increment() < 1e-8:
                    break
                problem.advance_prev()
            return its

        def cpu_leg(budget):
            # a BOUNDED sample: the first Newton steps of the same LVPP run, each an exact solve by the oracle's nested-dissection
            # multifrontal LU (oracle/nd_lu.py, LAPACK/BLAS on one thread; symbolic analysis untimed, as on the GPU side)
            from oracle import gc_oracle as G
            from oracle import nd_lu
            from oracle import pg_oracle as O

            m = args.cpu_n // 4
            c, e = O.create_rectangle(m, m, (0.0, 0.0), (1.0, 1.0))
            prob = G.GradientConstraintP2(c, e)
            nd_lu.MAX_THREADS = 1
            ls = nd_lu.NDLinearSolve(*nd_lu.nodes_of_problem(prob))

            class _Budget(Exception):
                pass

            state = {"steps": 0, "t0": None}

            def solve(J, rhs):
                if state["t0"] is None:  # the first call builds the symbolic analysis and touches the arena: untimed
                    dx = ls(J, rhs)
                    state["t0"] = time.perf_counter()
                    return dx
                dx = ls(J, rhs)
                state["steps"] += 1
                state["dt"] = time.perf_counter() - state["t0"]
                if state["dt"] > budget:
                    raise _Budget
                return dx

            try:
                G.solve_problem(prob, linear_solve=solve)
            except _Budget:
                pass
            return state["steps"], state["dt"], (f"the first Newton steps of the LVPP run on a {m}x{m} mesh ({prob.ntot} unknowns; "
                                                  "a Newton step = assembly + factorisation + solve with refinement)")
    else:
        from proximalgalerkin_amd import signorini as G

        n = args.n if args.n != 2048 else 70
        mesh = G.create_unit_cube(n, n, n)
        mt, _ = G.native_tags(mesh)
        problem = G.SignoriniProblem(mesh, mt.find(2), np.unique(mt.find(1).ravel()), 2.0e4, 0.3, 0.0, -0.25, device=local_rank,
                                     comm=comm)
        workload = (f"examples/02_signorini: unit cube, {n}^3 x 6 = {mesh.cells.shape[0]} P1 tetrahedra ({problem.ndofs} unknowns), "
                    "E 2e4, nu 0.3, disp -0.25, gap 0, alpha doubling, Newton tol 1e-6 (1e-5 first step), LVPP tol 1e-6")

        def one_step():
            problem.set_state(np.zeros(problem.ndofs))
            problem.set_prev(np.zeros(problem.ndofs))
            its = []
            for it in range(1, 26):
                problem.set_alpha(2.0**it)
                tol = 1e-5 if it < 2 else 1e-6
                problem.solver.setTolerances(atol=tol, rtol=tol)
                its.append(problem.solve()[1])
                if problem.u_increment() <= 1e-6:
                    break
                problem.advance_prev()
            return its

        def cpu_leg(budget):
            # the full LVPP run on a 24^3 x 6 mesh (6 Newton steps) with the oracle's nested-dissection multifrontal LU; the first linear
            # solve carries the symbolic analysis and is not timed
            from oracle import nd_lu
            from oracle import sg_oracle as S

            m = 24
            c, t = S.create_unit_cube_tets(m, m, m)
            prob = S.SignoriniP1(c, t, S.boundary_facets_where(c, t, lambda x: np.isclose(x[:, 2], 0.0)),
                                 np.flatnonzero(np.isclose(c[:, 2], 1.0)))
            nd_lu.MAX_THREADS = 1
            ls = nd_lu.NDLinearSolve(*nd_lu.nodes_of_problem(prob))
            state = {"steps": 0, "t0": None, "dt": 0.0}

            def solve(J, rhs):
                dx = ls(J, rhs)
                if state["t0"] is None:
                    state["t0"] = time.perf_counter()
                else:
                    state["steps"] += 1
                    state["dt"] = time.perf_counter() - state["t0"]
                return dx

            S.solve_contact_problem(prob, linear_solve=solve)
            return state["steps"], state["dt"], (f"the LVPP run on {m}^3 x 6 tetrahedra ({prob.ntot} unknowns) after its first linear solve; "
                                                  "a Newton step = assembly + factorisation + solve with refinement")
    t_setup = time.perf_counter() - t_setup

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        its = one_step()
    barrier()
    t0 = time.perf_counter()
    newton_total = 0
    for _ in range(args.steps):
        its = one_step()
        newton_total += int(sum(its))
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        dt, _, _ = reduce_over_ranks(dist, dt, newton_total, 0, "cuda" if backend == "nccl" else "cpu")
    # roofline of the dominant kernel: one more solve with per-phase HIP-event timing (adds syncs: not part of `value`)
    problem.profile(True)
    its_p = one_step()
    prof = problem.profile(False)
    st = problem.lu_stats()
    # ALGORITHMIC flops (the unpadded multifrontal factorisation) decide `achieved`; the level-batched kernels execute
    # the padded count (fronts of a level share one shape), reported beside it
    # SYMMETRIC mode (round 5: the Newton matrices of ex 06 and - with the latent rows negated - ex 02 are symmetric): the factorisation
    # is L D L^T in LU clothing and its flop count HALF the LU's (n^3 / 3 against 2 n^3 / 3 per dense front); `achieved` is priced on
    # that count, the LU-equivalent rate is reported beside it
    sym = bool(st.get("symmetric"))
    fl = 0.5 if sym else 1.0
    tflops = fl * st["flops"] * sum(its_p) / (prof["lu_factor"] * 1e-3) / 1e12
    tflops_exec = fl * st["flops_padded"] * sum(its_p) / (prof["lu_factor"] * 1e-3) / 1e12
    # HBM bytes per factorisation from the committed rocprofv3 --pmc profile of the same workload (tools/profile_nd_traffic.sh:
    # FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled per MI355X_MICROARCH.md); its library hash travels with it
    tj = _ladder("r05_nd_traffic_ex06_1024.json" if args.workload == "ex06" else "r05_nd_traffic_ex02_70.json")
    traffic = traffic_src = None
    default_size = (args.workload == "ex06" and N == 1024) or (args.workload != "ex06" and n == 70)
    if tj and tj.get("per_factorisation") and default_size and world == 1:
        traffic = tj["per_factorisation"]["traffic_GB"] * 1e9
        traffic_src = {"file": "profiles/" + ("r05_nd_traffic_ex06_1024.json" if args.workload == "ex06" else "r05_nd_traffic_ex02_70.json"),
                       "libpgx_sha256_16": tj.get("libpgx_sha256_16"), "per": "factorisation (all k_nd_* kernels of pgx_nd_factor)",
                       "traffic_over_arena": tj["per_factorisation"].get("traffic_over_arena"),
                       "note": "FETCH_SIZE counts every read that leaves an XCD's L2, Infinity-Cache hits included; about two thirds of "
                               "the total are compulsory for a multifrontal factorisation (factors written once, every Schur block "
                               "written once and gathered once per tree level), the rest panel re-reads of the blocked elimination"}
    out = None
    if rank == 0:
        out = {
            "metric": f"proximal-Newton iterations/sec, {args.workload} (LVPP Newton inner loop, sparse-LU linear solves)",
            "value": newton_total / dt, "unit": "Newton iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong" if world > 1 else "none", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "step": "one full LVPP solve from the zero state",
                       "newton_iterations_per_step": newton_total / args.steps, "newton_per_lvpp_step": [int(i) for i in its],
                       "parallelism": "single" if world == 1 else f"replicated iterate, sparse LU distributed over {world} ranks "
                                                                   "(one dissection subtree each, RCCL gather/scatter + all-reduce)"},
            "setup_s": t_setup,
            "roofline": {"kernel": "k_nd_gemm8 / k_nd_gemm<2> inside pgx_nd_factor (fp64 MFMA GEMM of the multifrontal LU; the figure is "
                                   "the WHOLE factorisation: algorithmic flops of this rank / its device time - leaf fronts, frame "
                                   "assembly, diagonal blocks and panel solves included; the deep tree levels of 2-D problems are "
                                   "HBM-bound, DESIGN.md section 9)",
                         "bound": "mfma", "achieved": tflops, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tflops / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "factorisation": "L D L^T in LU clothing (pgx_nd_set_symmetric): half the LU's flops" if sym else "LU",
                         "algorithmic_flops_per_factorisation": fl * st["flops"],
                         "executed_flops_per_factorisation_padded": fl * st["flops_padded"], "executed_TFLOPs": tflops_exec,
                         "lu_equivalent_TFLOPs": tflops / fl,
                         "arena_GB": st["arena_doubles"] * 8 / 1e9,
                         "lu_factor_ms_per_newton_step": prof["lu_factor"] / max(sum(its_p), 1),
                         "lu_solve_ms_per_newton_step": prof["lu_solve"] / max(sum(its_p), 1)},
        }
        if not args.no_cpu_baseline and world == 1:
            steps, secs, what = cpu_leg(25.0)
            n_cpu = args.cpu_n // 4 if args.workload == "ex06" else 24
            n_gpu = N if args.workload == "ex06" else n
            out["cpu_baseline"] = {"value": steps / secs, "unit": "Newton iterations/s", "cores": 1, "kind": "port",
                                   # MEASURED here, on the mesh named in `mesh`; `at_workload` carries it to the benchmarked mesh
                                   "mesh": f"{n_cpu} cells per side", "workload_mesh": f"{n_gpu} cells per side",
                                   "sample": f"{steps} Newton steps ({secs:.1f} s) of {what}: numpy assembly + "
                                             "nested-dissection multifrontal LU (oracle/nd_lu.py) exact Newton, 1 thread (the oracle; a stand-in for, not a "
                                             "measurement of, FEniCSx+MUMPS)",
                                   **host_info()}
            ex = extrapolate(steps / secs, n_cpu, n_gpu, _ladder("r03_cpu_ladder_ex06_nd.json" if args.workload == "ex06" else "r03_cpu_ladder_ex02_nd.json"))
            if ex:
                ex["gpu_over_cpu"] = out["value"] / ex["value"]
                out["cpu_baseline"]["at_workload"] = ex
    problem.close()
    if comm is not None:
        comm.free()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def make_comm(local_rank):
    """The communicator of the N>1 launch: RCCL over xGMI.  BENCH_COMM=shm (test hook) swaps in the host-staged shared-memory
    transport so that the SAME launch runs with all ranks on one GPU, where RCCL refuses to start (tests/test_gpu_multiprocess.py)."""
    from proximalgalerkin_amd import comm as pcomm

    try:
        if os.environ.get("BENCH_COMM", "rccl") == "shm":
            c = pcomm.shm_from_torch_distributed()
            c.selfcheck(float(os.environ.get("PGX_COMM_SELFCHECK_TIMEOUT", "10")))
            return c
        return pcomm.rccl_from_torch_distributed(local_rank)  # runs the same self-check after ncclCommInitRank
    except Exception as e:  # first contact between the ranks failed: say which call, leave at once (a hung collective cannot be undone)
        sys.stderr.write(f"bench.py rank {os.environ.get('RANK', '0')}: communicator not usable - {e}\n")
        sys.stderr.flush()
        os._exit(3)


def reduce_over_ranks(dist, dt, newton_total, outer_total, device):
    """Launch-contract aggregation: wall time = MAX over ranks, work counts = SUM over ranks (whole-job value)."""
    import torch

    t = torch.tensor([dt], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([newton_total, outer_total], device=device, dtype=torch.float64)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), int(c[0].item()), int(c[1].item())


def self_launch(n):
    """`python bench.py --gpus N ...` -> `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port <free> bench.py --gpus N ...` as a child process; its stdout/stderr pass through, its return code is ours."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: no launcher environment, starting " + " ".join(cmd) + "\n")
    sys.stderr.flush()
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-leg":  # the CPU-baseline child (cpu_baseline): no torch, no GPU
        a = json.loads(sys.argv[2])
        runs = cpu_baseline_inprocess(a["n_sample"], a["settings"], a["budget_s"], tuple(a["threads"]), a["degree"], a["min_steps"])
        print(json.dumps([list(r) for r in runs]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    # not "--n": that is an ambiguous prefix of torch.distributed.run's own options
    ap.add_argument("--cells", dest="n", type=int, default=2048, help="cells per side (BASELINE config: 2048)")
    ap.add_argument("--settings", choices=["A", "B"], default="B")
    ap.add_argument("--degree", type=int, choices=[1, 2], default=1, help="Lagrange degree (obstacle_pg.py -p)")
    ap.add_argument("--cpu-n", type=int, default=1024, help="mesh size of the bounded CPU-baseline sample (cells per side)")
    ap.add_argument("--cpu-threads", type=int, default=1, help="BLAS threads of the CPU baseline (1 = the reference's OMP_NUM_THREADS=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the threaded-BLAS repeat of the CPU baseline sample")
    ap.add_argument("--solves-only", action="store_true",
                    help="profiling aid: skip the roofline microbenchmarks after the timed solves (their V-cycle replays would "
                         "distort a per-kernel time breakdown); `roofline` is then null")
    ap.add_argument("--profile", action="store_true", help="per-phase device times (adds syncs; not for `value`)")
    ap.add_argument("--opts", default="", help="extra solver options key=val,key=val (e.g. ksp_gmres_restart=20)")
    ap.add_argument("--replicas", action="store_true", help="N>1: N independent solves instead of one sharded solve")
    ap.add_argument("--dist-levels", type=int, default=0, help="sharded: multigrid levels kept distributed (0 = auto)")
    ap.add_argument("--watchdog", type=float, default=900.0, help="N>1: abort the process after this many seconds")
    ap.add_argument("--workload", choices=["ex01", "ex06", "ex02"], default="ex01",
                    help="ex01 = BASELINE metric (config 2); ex06 / ex02 = configs 4 / 5 through the sparse LU")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the launcher ourselves, as a CHILD process and before anything in
        # this process has touched the GPU (never exec), relay its one JSON line and its return code
        return self_launch(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s); refusing to print a line "
                         f"whose n_gpus would not be what was asked for\n")
        raise SystemExit(2)
    import torch

    dist = None
    # test hooks (a 1-GPU box cannot host two NCCL ranks): BENCH_DIST_BACKEND=gloo, BENCH_FORCE_DEVICE=0
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if "BENCH_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
    # BENCH_FORCE_SHARDED=1: take the sharded (RCCL) branch even with ONE rank - everything the N>1 launch executes except
    # traffic between ranks; this is how a one-GPU box exercises it (tests/test_gpu_sharded.py)
    force_sharded = os.environ.get("BENCH_FORCE_SHARDED") == "1"
    if world > 1 or force_sharded:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if dist is not None:
        # every call on a sharded / distributed-LU handle is collective: if one rank dies the others would wait in RCCL for
        # ever.  Bound the damage for EVERY workload: the whole run has args.watchdog seconds.
        import threading

        def _abort():
            sys.stderr.write(f"bench.py rank {rank}: no result after {args.watchdog} s - aborting (stuck collective?)\n")
            sys.stderr.flush()
            os._exit(3)

        wd = threading.Timer(args.watchdog, _abort)
        wd.daemon = True
        wd.start()

    if args.workload != "ex01":
        return bench_lu_workload(args, rank, world, local_rank, dist, backend)

    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem

    S = SETTINGS[args.settings]
    N = args.n
    # ---- setup (untimed): mesh, obstacle at quadrature points, plan, constant blocks, MG hierarchy ----
    t_setup = time.perf_counter()
    sharded = (world > 1 and not args.replicas) or force_sharded
    comm = None
    if sharded:
        comm = make_comm(local_rank)  # id / segment-name broadcast through the torch.distributed group
    msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N), comm=comm, dist_levels=args.dist_levels)
    petsc_options = None
    if args.opts:
        petsc_options = {"snes_error_if_not_converged": True, "snes_linesearch_type": "none", "snes_rtol": 1e-6,
                         "snes_max_it": 100}  # obstacle_pg.py:128-139
        for kv in args.opts.split(","):
            k, v = kv.split("=")
            for cast in (int, float, str):
                try:
                    petsc_options[k] = cast(v)
                    break
                except ValueError:
                    pass
    problem, sol, sol_k, alpha = setup_problem(msh, args.degree, petsc_options=petsc_options, device=local_rank)
    t_setup = time.perf_counter() - t_setup
    if os.environ.get("BENCH_TEST_DIE_RANK") == str(rank):  # test hook: a rank that vanishes before the first collective solve
        os._exit(17)
    if args.profile:
        problem.profile(enable=True, reset=True)

    def one_step():
        return run_outer_loop(problem, sol, sol_k, alpha, S["max_outer"], S["alpha_scheme"], S["alpha_max"],
                              S["tol_exit"], device_resident=True, verbose=False)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        hist = one_step()
    if args.profile:
        problem.profile(reset=True)
    barrier()
    if sharded:
        problem.comm_counts(reset=True)
    t0 = time.perf_counter()
    newton_total, outer_total = 0, 0
    for _ in range(args.steps):
        hist = one_step()
        newton_total += int(sum(hist["Newton steps"]))
        outer_total += hist["outer_iterations"]
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        dt, nsum, osum = reduce_over_ranks(dist, dt, newton_total, outer_total, "cuda" if backend == "nccl" else "cpu")
        if sharded:  # ONE solve: every rank counted the same iterations; the job's work is that solve, once
            assert nsum == newton_total * world and osum == outer_total * world, "ranks disagree on iteration counts"
        else:
            newton_total, outer_total = nsum, osum
    comm_counts = problem.comm_counts() if sharded else None  # collectives of the timed solves, this rank
    units = 1 if sharded else world  # solves per step over the whole job
    lin_its = problem.solver.getLinearSolveIterations()

    # ---- roofline of the operator-apply kernel ("SpMV"): HIP events on the library's own stream ----
    # structured P1: the solver applies J matrix-free (k_st_spmv_r<true>: 57 B per vertex - the iterate is the float2 z_j the single-precision
    # cycle left; 65 B with an fp64 iterate); the block-CSR stream kernel (general
    # meshes, P2; 232 B per P1 row) is timed beside it on the same matrix and reported as `roofline_csr`
    problem.assemble_jacobian()  # Jacobian at the final iterate
    spmv_kind = problem.spmv_select()
    spmv_ms = spmv_bytes = csr_ms = csr_bytes = None
    cold_ms, cold_bytes = {}, None
    if not args.solves_only:
        spmv_ms, spmv_bytes = problem.spmv_bench(reps=20)
        cold_ms[spmv_kind], cold_bytes = problem.spmv_bench_cold(reps=20)  # (the cold pass applies to an fp64 vector: its own byte count)
        if spmv_kind != 0:
            problem.spmv_select(0)
            csr_ms, csr_bytes = problem.spmv_bench(reps=20)
            cold_ms[0] = problem.spmv_bench_cold(reps=20)[0]
            problem.spmv_select(spmv_kind)
        else:
            csr_ms, csr_bytes = spmv_ms, spmv_bytes
    n = msh.num_vertices
    smoother = None
    if not sharded and args.degree == 1 and not args.solves_only:
        sm_ms, sm_bytes = problem.smoother_bench(reps=50)
        # the finest level's smoother launch of the single-precision V-cycle (pgx_mg32.hip): D as one float4, vectors as float2 -
        # 40 B per vertex + the coarse correction; with PGX_MG_F32=0 the fp64 kernel k_st_smoothR (84 B per vertex) is timed instead
        f32 = sm_bytes < 60.0 * n
        sm_traffic, sm_src = (None, None)
        if f32 and N == 2048:
            tj = _ladder("r05_fsmooth_pmc_traffic.json")
            if tj and tj.get("cells") == N:
                sm_traffic = tj["hbm_traffic_bytes_per_launch"]
                sm_src = {k: tj.get(k) for k in ("file", "kernel", "libpgx_sha256_16", "date", "traffic_over_algorithmic")}
        smoother = {"kernel": ("k_f_smooth<16,3,...> on the finest level (three collective-Jacobi sweeps on x + P x_c per launch, single "
                               "precision inside the multigrid preconditioner; the time-dominant kernel of the solve: "
                               "profiles/r04_bench_2048_trace_by_level.txt)") if f32 else
                              "k_st_smoothR<16,3,POST> on the finest level (fp64 V-cycle, PGX_MG_F32=0)",
                    "bound": "hbm", "achieved": sm_bytes / (sm_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": sm_bytes / (sm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": sm_traffic, "traffic_source": sm_src,
                    "algorithmic_bytes_per_launch": sm_bytes, "avg_launch_ms": sm_ms}
    if not sharded and args.degree == 2 and not args.solves_only:
        try:  # P2: the patch sweep of the two-level cycle (k_patch_apply + k_patch_edges) on the inverses of the final Jacobian
            sm_ms, sm_bytes = problem.smoother_bench(reps=20)
            p_traffic = p_src = None
            ta, te = _ladder("r05_patch_apply_pmc_traffic.json"), _ladder("r05_patch_edges_pmc_traffic.json")
            if ta and te and ta.get("cells") == N == te.get("cells"):  # the sweep = both kernels: their PMC traffic added
                p_traffic = ta["hbm_traffic_bytes_per_launch"] + te["hbm_traffic_bytes_per_launch"]
                p_src = {"files": [ta.get("file"), te.get("file")], "libpgx_sha256_16": ta.get("libpgx_sha256_16"), "date": ta.get("date"),
                         "traffic_over_algorithmic": p_traffic / (ta["algorithmic_bytes_per_launch"] + te["algorithmic_bytes_per_launch"]),
                         "k_patch_apply_traffic_over_algorithmic": ta.get("traffic_over_algorithmic"),
                         "k_patch_edges_traffic_over_algorithmic": te.get("traffic_over_algorithmic")}
            smoother = {"kernel": "k_patch_apply<7, float, SYM> + k_patch_edges (one additive vertex-star patch sweep of the P2 level: float "
                                  "inverses in symmetric packing, 512 B per patch; the time-dominant kernel pair of the P2 solve: "
                                  "profiles/r04_config3_p2_2048_kernel_stats.csv)",
                        "bound": "hbm", "achieved": sm_bytes / (sm_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": sm_bytes / (sm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": p_traffic, "traffic_source": p_src,
                        "algorithmic_bytes_per_launch": sm_bytes, "avg_launch_ms": sm_ms}
        except Exception as e:  # handles without the patch smoother (general meshes with vertex degree > 7)
            smoother = None
            sys.stderr.write(f"bench.py: no roofline_dominant for this P2 handle: {e}\n")
    coarse = None
    if not sharded and args.degree == 1 and not args.solves_only:
        # the part of one V-cycle on the levels of at most 513^2 vertices, launches back to back
        lv, part = 0, []
        while True:
            try:
                ms_l, n_l = problem.vcycle_bench(lv, reps=30)
            except Exception:
                break
            part.append((lv, n_l, ms_l))
            lv += 1
        tail_ms, tail_n = problem.vcycle_bench(-1, reps=30)
        first = next((p for p in part if p[1] <= 513 * 513), None)
        coarse = {"whole_vcycle_us": 1e3 * part[0][2], "from_level_us": {f"{n_l} vertices": 1e3 * ms_l for _, n_l, ms_l in part if n_l >= tail_n},
                  "levels_up_to_513x513_us": 1e3 * first[2] if first else None, "fused_tail_launch_us": 1e3 * tail_ms,
                  "how": "pgx_vcycle_bench: HIP events around 30 back-to-back sub-cycles starting on each level"}
    prof = problem.profile() if args.profile else None
    if sharded:
        parallelism = (f"sharded: ONE {N}x{N} solve on {world} strips of {N // world} vertex rows (+ ghost rows, "
                       f"{msh.partition.dist_levels} distributed multigrid levels, coarser levels replicated), "
                       + ("RCCL halo exchange + packed all-reduces over xGMI" if comm.kind == "rccl" else
                          f"transport '{comm.kind}' (host-staged rehearsal of the RCCL launch, not a performance number)"))
    elif world > 1:
        parallelism = "replicas: N INDEPENDENT solves, one per GPU (--replicas) - NOT a speed-up measurement"
    else:
        parallelism = "single"

    def pmc_traffic(fname):  # HBM bytes per launch from a committed rocprofv3 --pmc profile of the same kernel and mesh
        tj = _ladder(fname)
        if tj and tj.get("cells") == N and not sharded:
            return tj["hbm_traffic_bytes_per_launch"], {k: tj.get(k) for k in ("file", "kernel", "libpgx_sha256_16", "date",
                                                                               "traffic_over_algorithmic")}
        return None, None

    def spmv_roofline(kind, ms, nbytes):
        name = {0: "k_bspmv_stream (block-CSR SpMV of the Newton matrix [[aK,M],[M,-D]], one shared pattern)",
                1: "k_st_spmv_r<true> (matrix-free apply of the Newton matrix [[aK,M],[M,-D]] to the float2 z_j of FGMRES, as in the solve: constant K/M "
                   "stencils, half-stored D(psi) stencil; the outer-Krylov SpMV of this workload)",
                2: "k_st_apply<0> (generic matrix-free stencil apply)"}.get(kind, "")
        if kind == 3:
            name = ("k_p2st_apply_lds + k_p2_rows_csr (P2 operator apply on the structured mesh, csrc/pgx_p2st.hip: interior vertex/edge "
                    "groups through a 46-entry table - no column indices, K and M as constants, D(psi) from its structure-of-arrays "
                    "copy, the iterate staged in LDS: 496 B per group; the frame rows in CSR form)")
        if kind == 0 and args.degree == 2:
            name = ("k_bspmv_bal (P2 operator apply: nnz-balanced CSR-stream SpMV of [[aK,M],[M,-D]]; on this uniform mesh K and M are "
                    "read through a one-byte (K,M)-pair dictionary: 13 B per entry instead of 28)")
        traffic, src = pmc_traffic(("r05_p2stspmv_pmc_traffic.json" if kind == 3 else "r03_p2spmv_pmc_traffic.json") if args.degree == 2 else
                                   {0: "r05_spmv_pmc_traffic.json", 1: "r05_stspmv_pmc_traffic.json"}.get(kind, "none"))
        gbs = nbytes / (ms * 1e-3) / 1e9
        r = {"kernel": name + (", rank 0's strip" if sharded else ""), "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
             "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
             # HBM bytes per launch of this kernel from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, gfx950 x2 read
             # correction; tools/pmc_summary.py).  A profile of another run, not of this one: `traffic_source` says which; null
             # when no committed profile matches the workload.
             "traffic": traffic, "traffic_source": src, "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms}
        if src:
            src["note"] = ("FETCH_SIZE counts requests that reach the memory side of the fabric, Infinity-Cache (MALL) hits included: "
                           "it bounds re-reads, it does not separate HBM from MALL traffic (MI355X_MICROARCH.md)")
        if kind in cold_ms and cold_ms[kind]:
            # `frac` is measured in the cache state of a solve (the vector the operator reads was just written by the V-cycle and
            # the 273 MB operator straddles the 256 MB Infinity Cache); `cold_frac` after a 512 MB sweep of unrelated storage
            r["cold_avg_launch_ms"] = cold_ms[kind]
            cb = cold_bytes if (kind == spmv_kind and cold_bytes) else nbytes
            r["cold_frac"] = cb / (cold_ms[kind] * 1e-3) / 1e9 / HBM_PEAK_GBS
            r["cold_bytes_per_launch"] = cb
        if kind == 0 and args.degree == 1:
            r["mixed_csr_equivalent_GBs"] = (12.0 * 4 * (nbytes - 4.0 * (n + 1) - 32.0 * n) / 28.0 + 20.0 * 2 * n) / (ms * 1e-3) / 1e9
        elif csr_bytes:
            # SURVEY.md section 8(d) prices the operator as the mixed CSR the reference assembles (12 B per nnz of the 2n x 2n matrix +
            # 20 B per row = 1.578 GB at 2048^2): what a CSR SpMV would have to stream for the SAME product - a rate of work, not a
            # bandwidth (this kernel moves 65 B per vertex)
            r["mixed_csr_equivalent_GBs"] = (12.0 * 4 * (csr_bytes - 4.0 * (n + 1) - 32.0 * n) / 28.0 + 20.0 * 2 * n) / (ms * 1e-3) / 1e9
        return r

    out = None
    if rank == 0:
        out = {
            # BASELINE.json's metric is quoted on 2048^2 P1; other --cells/--degree runs say what they measured
            "metric": f"proximal-Newton iterations/sec, {N}^2 P{args.degree} obstacle (LVPP Newton inner loop)",
            "value": newton_total / dt,
            "unit": "Newton iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            # one GPU: neither weak nor strong; N > 1: the SAME problem on N strips, or N independent replicas
            "scaling": "strong" if sharded else ("weak" if world > 1 else "none"),
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{N}x{N} right-diagonal P{args.degree} obstacle problem on [-1,1]^2, phi_set obstacle, f=0, "
                            f"settings {args.settings}: alpha {S['alpha_scheme']}, alpha_max {S['alpha_max']:g}, "
                            f"tol {S['tol_exit']:g}; snes_rtol 1e-6, Newton linear solves to true relative residual 1e-10",
                "mixed_unknowns": 2 * (N + 1) ** 2 if args.degree == 1 else sol.function_space.num_dofs,
                "step": "one full LVPP solve from the zero state",
                "newton_iterations_per_step": newton_total / args.steps / units,
                "proximal_iterations_per_step": outer_total / args.steps / units,
                "parallelism": parallelism,
            },
            "proximal_iterations_per_s": outer_total / dt,
            "last_newton_linear_iterations": lin_its,
            "setup_s": t_setup,
            "roofline": spmv_roofline(spmv_kind, spmv_ms, spmv_bytes) if spmv_ms else None,
        }
        if spmv_kind != 0 and csr_ms:
            out["roofline_csr"] = spmv_roofline(0, csr_ms, csr_bytes)
        if comm_counts:
            kit = max(comm_counts["krylov_iterations"], 1)
            out["config"]["collectives_per_krylov_iteration"] = {
                "halo_exchanges": comm_counts["halo_exchanges"] / kit, "allreduces": comm_counts["allreduces"] / kit,
                # what the LIBRARY used: it never reads the environment; the loader forwards PGX_* keys only under PGX_TUNING_FROM_ENV=1
                "ghost_depth_multiplier": (int(os.environ.get("PGX_GHOST_MUL", "3"))
                                           if os.environ.get("PGX_TUNING_FROM_ENV") == "1" else 3), "counted": comm_counts}
        if smoother:
            out["roofline_dominant"] = smoother
        if coarse:
            out["vcycle_parts"] = coarse
        if prof:
            out["phase_ms"] = prof
        if not args.no_cpu_baseline and world == 1:
            # degree 2: the P2 problem on half the cells per side has the node count of the P1 sample (a `--degree 2`
            # line must not carry a P1 baseline)
            cpu_n = min(args.cpu_n, N) if args.degree == 1 else min(args.cpu_n // 2, N)
            hi = host_info()
            all_cores = min(hi["host_cores_usable"] or 1, 16)  # the box's CPU share per GPU (16): 16 subtree workers, 16 BLAS threads above them
            threads = [args.cpu_threads] + ([all_cores] if all_cores > args.cpu_threads and not args.no_cpu_all_cores else [])
            runs = cpu_baseline(cpu_n, S, threads=threads, degree=args.degree)
            v, steps, secs, detail = runs[0]
            ladder = _ladder("r03_cpu_ladder_nd.json") if args.degree == 1 else None
            out["cpu_baseline"] = {
                "value": v,
                "unit": "Newton iterations/s",
                "cores": args.cpu_threads,
                "kind": "port",
                # `value` is MEASURED, here, now, on the mesh named in `mesh`.  When that is smaller than the benchmarked mesh,
                # `at_workload` carries it there with the exponent of the committed ladder (2-D nested dissection: flops ~ N^3)
                # and says so; `--cpu-n 2048` measures on the workload itself (about 25 GB of factors, minutes per Newton step).
                "mesh": f"{cpu_n}x{cpu_n} P{args.degree}",
                "workload_mesh": f"{N}x{N} P{args.degree}",
                "sample": f"{steps} Newton step(s) ({secs:.1f} s) of the same LVPP run (settings {args.settings}) on a "
                          f"{cpu_n}x{cpu_n} P{args.degree} mesh: numpy assembly + exact Newton with a nested-dissection multifrontal LU "
                          f"on LAPACK/BLAS (oracle/nd_lu.py), {args.cpu_threads} thread(s); symbolic analysis and mesh setup untimed "
                          f"(as on the GPU side).  The oracle - a stand-in for, not a measurement of, FEniCSx+MUMPS",
                "detail": detail,
                **hi,
            }
            if len(runs) > 1:  # BASELINE.md section 3's secondary figure: the same sample with threaded BLAS
                v2, steps2, secs2, detail2 = runs[1]
                out["cpu_baseline"]["all_cores"] = {"value": v2, "unit": "Newton iterations/s", "cores": threads[1],
                                                    "sample": f"{steps2} Newton step(s) ({secs2:.1f} s), same mesh and code, "
                                                              f"{threads[1]} worker processes / BLAS threads", "detail": detail2,
                                                    "note": "tree-parallel numeric factorisation (forked workers, one per subtree of the "
                                                            "dissection, up to 16; the levels above them with threaded BLAS), threaded "
                                                            "BLAS in the solves; assembly stays serial numpy"}
            if cpu_n != N:  # (sample mesh == workload mesh: directly comparable for ANY degree - the else branch)
                ex = extrapolate(v, cpu_n, N, ladder) if ladder else None
                if ex:
                    ex["gpu_over_cpu"] = out["value"] / ex["value"]
                    out["cpu_baseline"]["at_workload"] = ex
                    if len(runs) > 1:
                        ex2 = extrapolate(runs[1][0], cpu_n, N, ladder)
                        ex2["gpu_over_cpu"] = out["value"] / ex2["value"]
                        ex2["method"] += f" ({threads[1]} threads; the ladder's exponent was measured with one thread)"
                        out["cpu_baseline"]["all_cores"]["at_workload"] = ex2
                else:
                    out["cpu_baseline"]["at_workload"] = None
                    out["cpu_baseline"]["at_workload_note"] = ("no committed CPU ladder for this degree: the sample is reported "
                                                               "on its own mesh and not carried to the workload")
            else:
                out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / v
    problem.close()
    if comm is not None:
        comm.free()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
