"""The inter-process transport (pgx_comm_shm_init, include/pgx.h) between REAL processes, on host buffers (host_mode), so that
the protocol - rendezvous through torch.distributed(gloo), barrier, mailboxes, chunking, failure instead of hang - is covered
on machines without a GPU.  The same transport carries the sharded solver between processes in tests/test_gpu_multiprocess.py
(reference counterpart: MPI ghost updates and all-reduces, src/lvpp/problem.py:56-73, obstacle_pg.py:50)."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp(a):
    from proximalgalerkin_amd import _lib

    return _lib.dptr(a)


def _worker(rank, world, port, slot_bytes, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from proximalgalerkin_amd import _lib
    from proximalgalerkin_amd.comm import shm_from_torch_distributed

    lib = _lib.load()
    c = shm_from_torch_distributed(slot_bytes=slot_bytes, host_mode=True)
    res = {}
    # the start-of-run self-check (pgx_comm_selfcheck): verified halo pattern with both neighbours + packed all-reduce
    c.selfcheck(5.0, host_mode=True)
    res["selfcheck"] = True
    # all-reduce, longer than one mailbox (chunked)
    n = 3000
    a = np.arange(n, dtype=np.float64) * (rank + 1)
    assert lib.pgx_comm_allreduce(c._c, _dp(a), n) == 0
    res["allreduce"] = bool(np.array_equal(a, np.arange(n) * sum(r + 1.0 for r in range(world))))
    # halo: strips of 10 rows x 7, ghost depth 2 below / 3 above (asymmetric like the solver's g / g+1)
    sx, H, glo, ghi = 7, 10, (2 if rank > 0 else 0), (3 if rank + 1 < world else 0)
    rows = glo + H + ghi
    f0 = np.full(rows * sx, -1.0)
    f1 = np.full(rows * sx, -2.0)
    own = np.arange(H * sx, dtype=np.float64) + 1000.0 * rank
    f0[glo * sx:(glo + H) * sx] = own
    f1[glo * sx:(glo + H) * sx] = -own
    # to rank-1: my first 3 owned rows (its ghi); to rank+1: my last 2 owned rows (its glo)
    rc = lib.pgx_comm_halo(c._c, _dp(f0), _dp(f1), glo * sx, 3 * sx, 0, glo * sx, (glo + H - 2) * sx, 2 * sx, (glo + H) * sx, ghi * sx)
    assert rc == 0, lib.pgx_comm_last_error()
    ok = True
    if rank > 0:
        exp = (np.arange(H * sx) + 1000.0 * (rank - 1))[(H - 2) * sx:]
        ok &= bool(np.array_equal(f0[:glo * sx], exp) and np.array_equal(f1[:glo * sx], -exp))
    if rank + 1 < world:
        exp = (np.arange(H * sx) + 1000.0 * (rank + 1))[:3 * sx]
        ok &= bool(np.array_equal(f0[(glo + H) * sx:], exp) and np.array_equal(f1[(glo + H) * sx:], -exp))
    ok &= bool(np.array_equal(f0[glo * sx:(glo + H) * sx], own))
    res["halo"] = ok
    # gather0 / scatter0 (chunked)
    m = 1500
    send = np.full(m, float(rank)) + np.arange(m) * 1e-3
    recv0 = np.zeros(world * m)
    assert lib.pgx_comm_gather0(c._c, _dp(send), m, _dp(recv0)) == 0
    if rank == 0:
        res["gather0"] = all(np.array_equal(recv0[q * m:(q + 1) * m], np.full(m, float(q)) + np.arange(m) * 1e-3) for q in range(1, world))
    send0 = np.repeat(np.arange(world, dtype=np.float64), m) * 10.0 + np.tile(np.arange(m), world) * 1e-3
    recv = np.zeros(m)
    assert lib.pgx_comm_scatter0(c._c, _dp(send0), m, _dp(recv)) == 0
    if rank > 0:
        res["scatter0"] = bool(np.array_equal(recv, 10.0 * rank + np.arange(m) * 1e-3))
    # a rank that leaves the protocol makes the others FAIL (PGX_ECOMM) within the timeout instead of hanging
    if rank == 0:
        b = np.ones(4)
        res["peer_missing_rc"] = int(lib.pgx_comm_allreduce(c._c, _dp(b), 4))
        res["peer_missing_msg"] = lib.pgx_comm_last_error().decode()
        # ... and the start-of-run self-check NAMES the operation that could not complete (what a first-contact RCCL failure
        # must look like instead of a hang)
        try:
            c.selfcheck(2.0, host_mode=True)
            res["selfcheck_without_peer"] = "passed"
        except _lib.PgxError as e:
            res["selfcheck_without_peer"] = str(e)
    c.free()
    out.put((rank, res))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,slot_bytes", [(2, 4096), (3, 8192)])
def test_shm_transport_between_processes(world, slot_bytes):
    os.environ["PGX_COMM_TIMEOUT"] = "2"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, slot_bytes, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r]["selfcheck"] and got[r]["allreduce"] and got[r]["halo"], (r, got[r])
        if r > 0:
            assert got[r]["scatter0"]
    assert got[0]["gather0"]
    assert got[0]["peer_missing_rc"] == -6 and "did not arrive" in got[0]["peer_missing_msg"]
    msg = got[0]["selfcheck_without_peer"]
    assert "self-check" in msg and "halo exchange" in msg and "rank 0 of" in msg, msg
