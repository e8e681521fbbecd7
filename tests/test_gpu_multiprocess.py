"""bench.py launched the way the driver launches it for N > 1 - `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N` - with REAL processes, on a box with one GPU: BENCH_DIST_BACKEND=gloo + BENCH_FORCE_DEVICE=0 put every rank on GPU 0
and BENCH_COMM=shm swaps RCCL (which refuses two ranks on one device) for the host-staged shared-memory transport
(include/pgx.h: pgx_comm_shm_init).  Everything else is what the multi-GPU run executes: process spawn and rendezvous, the
broadcast that bootstraps the communicator, strip meshes, the collective call sequence of every rank (a rank that took a different
branch would stop the run: the transport times out instead of hanging), the watchdog, the max/sum aggregation and the JSON line.
Reference counterpart: `mpirun -n N python obstacle_pg.py` (obstacle_pg.py:64; ghost updates src/lvpp/problem.py:56-73)."""
import json
import os
import pathlib
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parents[1]


def _launch(nproc, args, timeout=600):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_FORCE_DEVICE="0", BENCH_COMM="shm", PGX_COMM_TIMEOUT="120",
               PGX_CHECK_REPLICAS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", str(nproc)] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def _single(args):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1"] + args, capture_output=True, text=True, timeout=600,
                       cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.parametrize("nproc,cells", [(2, 256), (4, 512)])
def test_sharded_obstacle_solve_across_processes(require_gpu, nproc, cells):
    args = ["--cells", str(cells), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--watchdog", "500"]
    d = _launch(nproc, args)
    d1 = _single(args)
    assert d["n_gpus"] == nproc and d["scaling"] == "strong"
    assert d["config"]["parallelism"].startswith(f"sharded: ONE {cells}x{cells} solve on {nproc} strips")
    # the same algebra as the single-process solve: identical Newton and proximal counts
    assert d["config"]["newton_iterations_per_step"] == d1["config"]["newton_iterations_per_step"]
    assert d["config"]["proximal_iterations_per_step"] == d1["config"]["proximal_iterations_per_step"]
    assert abs(d["last_newton_linear_iterations"] - d1["last_newton_linear_iterations"]) <= 2
    assert d["value"] > 0 and d["roofline"]["traffic"] is None


@pytest.mark.parametrize("workload,cells", [("ex02", 10), ("ex06", 48)])
def test_distributed_lu_workloads_across_processes(require_gpu, workload, cells):
    args = ["--workload", workload, "--cells", str(cells), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--watchdog", "500"]
    d = _launch(2, args)
    d1 = _single(args)
    assert d["n_gpus"] == 2
    assert d["config"]["newton_per_lvpp_step"] == d1["config"]["newton_per_lvpp_step"]


def test_a_dead_rank_stops_the_job_instead_of_hanging_it(require_gpu):
    """One rank exits before the solve (PGX_TEST_DIE_RANK): the survivors' next collective fails within PGX_COMM_TIMEOUT and the
    launcher returns non-zero - the behaviour the watchdog / transport timeouts exist for."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_FORCE_DEVICE="0", BENCH_COMM="shm", PGX_COMM_TIMEOUT="5",
               BENCH_TEST_DIE_RANK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--cells", "128", "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--watchdog", "60"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=str(ROOT))
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_launches_itself_when_asked_for_more_than_one_gpu(require_gpu):
    """`python bench.py --gpus 2` WITHOUT a launcher must not silently measure one GPU: it starts torch.distributed.run as a child
    (before touching the GPU, never exec) and relays the one JSON line; n_gpus is what was asked for."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_FORCE_DEVICE="0", BENCH_COMM="shm", PGX_COMM_TIMEOUT="120")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--cells", "256", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--watchdog", "500"], env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"


def test_sharded_p2_obstacle_solve_across_processes(require_gpu):
    """BASELINE config 3's shape - P2, the mesh cut into strips, one process per strip - launched exactly as the driver launches it
    (settings A; two ranks on one GPU through the shm transport): same counts as the single-process run."""
    args = ["--degree", "2", "--cells", "128", "--settings", "A", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--watchdog", "500",
            "--opts", "pc_type=pgx_mg"]
    d = _launch(2, args)
    d1 = _single(args)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["parallelism"].startswith("sharded: ONE 128x128 solve on 2 strips")
    assert d["config"]["newton_iterations_per_step"] == d1["config"]["newton_iterations_per_step"]
    assert d["config"]["proximal_iterations_per_step"] == d1["config"]["proximal_iterations_per_step"]
    assert d["value"] > 0
