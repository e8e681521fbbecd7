"""N>1 launch path on CPU: two processes over gloo exercise bench.py's rank aggregation (time = max over
ranks, work = sum over ranks) exactly as the torch.distributed.run launch does on GPUs."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench

    dt, newton, outer = bench.reduce_over_ranks(dist, 1.0 + 0.5 * rank, 22 + rank, 8, "cpu")
    dist.barrier()
    if rank == 0:
        out.put((dt, newton, outer))
    dist.destroy_process_group()


def test_two_rank_gloo_aggregation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == (1.5, 45, 16)  # max time, summed Newton and proximal iteration counts


def test_bench_cli_parses_and_flags_survive_torchrun_abbreviation_rules():
    """bench.py must import and parse; none of its long options may be an ambiguous PREFIX of a torch.distributed.run
    option (argparse abbreviation matching made `--n` fail under the launcher)."""
    import pathlib
    import re
    import subprocess
    import sys

    from torch.distributed.run import get_args_parser

    root = pathlib.Path(__file__).resolve().parents[1]
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    ours = set(re.findall(r"--[a-z][a-z0-9-]*", out.stdout)) - {"--help"}
    assert {"--gpus", "--steps", "--warmup"} <= ours
    theirs = [s for a in get_args_parser()._actions for s in a.option_strings if s.startswith("--")]
    for o in ours:
        clashes = [t for t in theirs if t.startswith(o)]
        assert not clashes, (o, clashes)


def test_bench_refuses_a_world_size_that_is_not_what_gpus_asks_for():
    """bench.py --gpus 4 under a launcher that started 2 ranks would print n_gpus = 2 for a run the driver files under 4: it must
    exit non-zero instead (before any GPU or torch work - this runs on the CPU-only container)."""
    import os
    import pathlib
    import subprocess
    import sys

    root = pathlib.Path(__file__).resolve().parents[1]
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()


def _partition_worker(rank, world, port, out):
    """Every rank resolves ITS strip through the C ABI (pgx_partition_rows: no GPU) and the ranks compare notes over gloo."""
    import ctypes as C

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from proximalgalerkin_amd import _lib

    lib = _lib.load()
    rows = []
    for ny, levels in ((64, 0), (128, 2), (2048, 3), (96, 1)):
        pt = _lib.pgx_partition(rank, world, ny, levels)
        r0, nr, o0, no = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        rc = lib.pgx_partition_rows(C.byref(pt), C.byref(r0), C.byref(nr), C.byref(o0), C.byref(no))
        rows.append((ny, rc, r0.value, nr.value, o0.value, no.value, pt.dist_levels))
    t = torch.tensor(rows, dtype=torch.int64)
    allr = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allr, t)
    if rank == 0:
        out.put([a.tolist() for a in allr])
    dist.barrier()
    dist.destroy_process_group()


def test_strip_ownership_tiles_the_vertex_rows_without_overlap():
    """Ownership arithmetic of the sharded path (include/pgx.h: pgx_partition_rows) across two REAL ranks, no GPU: the owned row
    ranges tile [0, ny] exactly once, every rank's local rows contain its owned rows plus the ghost rows the ABI promises
    (2^dist_levels times the ghost multiplier on interior sides, none on the domain boundary), neighbours' ghost rows are rows the
    other rank owns, and all ranks resolve the same dist_levels (reference counterpart: DOLFINx's index maps, owned + ghost dofs of
    a mesh created on MPI.COMM_WORLD, obstacle_pg.py:64)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_partition_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    allr = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ncase = len(allr[0])
    for k in range(ncase):
        per = [allr[r][k] for r in range(world)]
        ny = per[0][0]
        assert all(x[1] == 0 for x in per), per
        assert len({x[6] for x in per}) == 1  # same dist_levels everywhere
        cover = []
        for r, (_, _, row0, nrows, own0, nown, lv) in enumerate(per):
            assert row0 <= own0 and own0 + nown <= row0 + nrows  # owned rows are local rows
            assert (row0 == 0) == (r == 0) and (row0 + nrows == ny + 1) == (r == world - 1)  # no ghosts beyond the domain
            glo, ghi = own0 - row0, row0 + nrows - (own0 + nown)
            assert (glo > 0) == (r > 0) and (ghi > 0) == (r < world - 1)
            assert glo % (1 << lv) == 0 and (ghi - 1) % (1 << lv) == 0 or r == world - 1  # depth g below, g + 1 above
            cover += list(range(own0, own0 + nown))
        assert cover == list(range(ny + 1))  # every vertex row owned exactly once, in rank order
