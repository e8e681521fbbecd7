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
