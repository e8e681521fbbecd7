"""Communicators of the sharded path (include/pgx.h, "Sharded path").

The reference runs under `mpirun` and inherits its parallelism from DOLFINx/PETSc: meshes are created on
`MPI.COMM_WORLD` (/root/reference/examples/01_obstacle_problem/obstacle_pg.py:64), ghost values travel by
`Vec.ghostUpdate` (/root/reference/src/lvpp/problem.py:56-73) and scalars by `comm.allreduce`
(obstacle_pg.py:50).  Here a `Communicator` plays the role of that MPI communicator for ONE handle per GPU:

    rccl_from_torch_distributed(device)   one process per GPU (torch.distributed launch), RCCL over xGMI
    shm_from_torch_distributed()          one process per rank, host-staged through POSIX shared memory: the same launch on a
                                          box whose ranks share ONE GPU (RCCL refuses that) - rehearsal of the multi-GPU run
    local_group(n)                        n communicators for n host threads of one process (tests / one-GPU boxes)

`mesh.create_rectangle(points, n, comm=c)` then builds only this rank's strip of vertex rows.
"""
from __future__ import annotations

import ctypes as C

from . import _lib


class Communicator:
    def __init__(self, ptr, rank: int, size: int, kind: str):
        self._c, self.rank, self.size, self.kind = ptr, int(rank), int(size), kind

    def free(self):
        if self._c:
            _lib.load().pgx_comm_free(self._c)
            self._c = None

    def selfcheck(self, timeout_s: float = 10.0, host_mode: bool = False):
        """One verified halo exchange with both neighbours and one packed all-reduce, each bounded by `timeout_s` (include/pgx.h:
        pgx_comm_selfcheck).  Collective.  Raises PgxError naming the operation that failed or did not complete - call it before
        the first solve of a multi-rank run and END THE PROCESS on failure (a hung collective cannot be cancelled)."""
        lib = _lib.load()
        rc = lib.pgx_comm_selfcheck(self._c, int(host_mode), float(timeout_s))
        if rc:
            _comm_error(lib, "pgx_comm_selfcheck", rc)

    def __repr__(self):
        return f"Communicator({self.kind}, rank {self.rank} of {self.size})"


def _comm_error(lib, what, rc):
    msg = lib.pgx_comm_last_error()
    raise _lib.PgxError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def local_group(size: int):
    """`size` communicators that talk to each other inside this process: drive each handle from its own thread
    (ctypes releases the GIL during library calls).  All calls on sharded handles are collective."""
    lib = _lib.load()
    arr = (C.c_void_p * size)()
    rc = lib.pgx_comm_local_group(int(size), arr)
    if rc:
        _comm_error(lib, "pgx_comm_local_group", rc)
    return [Communicator(C.c_void_p(arr[r]), r, size, "local") for r in range(size)]


def rccl_from_torch_distributed(device: int):
    """RCCL communicator over the ranks of the initialised torch.distributed process group: rank 0 creates the
    ncclUniqueId, torch.distributed broadcasts its 128 bytes (any backend), every rank joins."""
    import torch.distributed as dist

    lib = _lib.load()
    rank, size = dist.get_rank(), dist.get_world_size()
    buf = C.create_string_buffer(128)
    if rank == 0:
        rc = lib.pgx_comm_rccl_unique_id(buf)
        if rc:
            _comm_error(lib, "pgx_comm_rccl_unique_id", rc)
    box = [bytes(buf.raw)]
    dist.broadcast_object_list(box, src=0)
    ptr = C.c_void_p()
    rc = lib.pgx_comm_rccl_init(box[0], rank, size, int(device), C.byref(ptr))
    if rc:
        _comm_error(lib, "pgx_comm_rccl_init", rc)
    c = Communicator(ptr, rank, size, "rccl")
    if size > 1:  # first contact between the ranks: a mis-wired launch must end with a message, not with a watchdog kill
        import os

        c.selfcheck(float(os.environ.get("PGX_COMM_SELFCHECK_TIMEOUT", "10")))
    return c


def rccl_single(device: int = 0):
    """A one-rank RCCL communicator (no peers): exercises library loading, communicator creation and the in-stream
    all-reduce on a single-GPU box."""
    lib = _lib.load()
    buf = C.create_string_buffer(128)
    rc = lib.pgx_comm_rccl_unique_id(buf)
    if rc:
        _comm_error(lib, "pgx_comm_rccl_unique_id", rc)
    ptr = C.c_void_p()
    rc = lib.pgx_comm_rccl_init(bytes(buf.raw), 0, 1, int(device), C.byref(ptr))
    if rc:
        _comm_error(lib, "pgx_comm_rccl_init", rc)
    return Communicator(ptr, 0, 1, "rccl")


def shm_init(name: str, rank: int, size: int, slot_bytes: int = 0, host_mode: bool = False):
    """Join the shared-memory group `name` ("/...") as `rank` of `size` (include/pgx.h: pgx_comm_shm_init)."""
    lib = _lib.load()
    ptr = C.c_void_p()
    rc = lib.pgx_comm_shm_init(name.encode(), int(rank), int(size), int(slot_bytes), int(host_mode), C.byref(ptr))
    if rc:
        _comm_error(lib, "pgx_comm_shm_init", rc)
    return Communicator(ptr, rank, size, "shm")


def shm_from_torch_distributed(slot_bytes: int = 0, host_mode: bool = False):
    """Shared-memory communicator over the ranks of the initialised torch.distributed process group (all on this host): rank 0
    picks the segment name, torch.distributed broadcasts it - the same bootstrap as rccl_from_torch_distributed."""
    import os
    import time

    import torch.distributed as dist

    rank, size = dist.get_rank(), dist.get_world_size()
    box = [f"/pgx_{os.getpid()}_{time.monotonic_ns() & 0xffffffff:x}" if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return shm_init(box[0], rank, size, slot_bytes, host_mode)
