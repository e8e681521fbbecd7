"""Sparse direct solver (geometric nested-dissection multifrontal LU on the GPU) - host mirror of include/pgx_nd.h.

Stands where the reference configures PETSc with ``"ksp_type": "preonly", "pc_type": "lu",
"pc_factor_mat_solver_type": "mumps"`` (examples/01_obstacle_problem/obstacle_pg.py:129-131,
examples/06_gradient_constraints/gradient_constraint_dolfinx.py:118-121).  No CPU numeric phase exists: `factor` and
`solve` raise without a GPU; `device=-1` builds the symbolic structure only (statistics, tests).
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np

from . import _lib as L


class DirectSolver:
    def __init__(self, indptr, indices, node_of_dof, node_coords, leaf_nodes: int = 0, device: int = 0, comm=None,
                 symbolic_rank: tuple | None = None):
        """comm: a proximalgalerkin_amd.comm.Communicator -> distributed factorisation (pgx_nd_create_dist): one subtree of
        the dissection tree per rank; every rank passes the same matrix / right-hand sides, all calls are collective."""
        self._lib = L.load()
        self._h = L._H()
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.node_of_dof = np.ascontiguousarray(node_of_dof, dtype=np.int32)
        self.node_coords = np.ascontiguousarray(node_coords, dtype=np.float64)
        self.n = len(self.indptr) - 1
        m = L.pgx_nd_matrix(self.n, L.iptr(self.indptr), L.iptr(self.indices), int(self.node_coords.shape[0]),
                            L.iptr(self.node_of_dof), int(self.node_coords.shape[1]), L.dptr(self.node_coords),
                            int(leaf_nodes))
        self._dist_size = 1
        if symbolic_rank is not None:  # (rank, size): symbolic view of one rank of a distributed factorisation (tests)
            self._dist_size = int(symbolic_rank[1])
            rc = self._lib.pgx_nd_create_symbolic_dist(C.byref(m), int(symbolic_rank[0]), int(symbolic_rank[1]), C.byref(self._h))
        elif comm is None:
            rc = self._lib.pgx_nd_create(C.byref(m), int(device), None, C.byref(self._h))
        else:
            self._comm = comm  # keep the communicator alive
            rc = self._lib.pgx_nd_create_dist(C.byref(m), comm._c, int(device), None, C.byref(self._h))
        if rc:
            msg = self._lib.pgx_nd_last_error(None)
            raise L.PgxError(f"pgx_nd_create failed (code {rc}): {msg.decode() if msg else ''}")

    def _check(self, rc, what):
        if rc:
            msg = self._lib.pgx_nd_last_error(self._h)
            raise L.PgxError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")

    def stats(self) -> dict:
        st = L.pgx_nd_stats()
        self._check(self._lib.pgx_nd_get_stats(self._h, C.byref(st)), "pgx_nd_get_stats")
        return {k: getattr(st, k) for k, _ in st._fields_}

    def factor(self, data):
        data = np.ascontiguousarray(data, dtype=np.float64)
        assert data.shape == (int(self.indptr[-1]),)
        self._check(self._lib.pgx_nd_factor(self._h, L.dptr(data), 0), "pgx_nd_factor")
        bad = self.stats()["perturbed_pivots"]
        if bad:  # static pivoting replaced (near-)zero pivots: the factorisation does not represent the matrix
            msg = self._lib.pgx_nd_last_error(self._h)
            warnings.warn(msg.decode() if msg else f"pgx_nd_factor: {bad} perturbed pivots", RuntimeWarning, stacklevel=2)

    def solve(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.empty_like(b)
        self._check(self._lib.pgx_nd_solve(self._h, L.dptr(b), L.dptr(x), 0), "pgx_nd_solve")
        return x

    def timing(self, enable=True):
        f, s = C.c_double(), C.c_double()
        self._check(self._lib.pgx_nd_timing(self._h, int(enable), C.byref(f), C.byref(s)), "pgx_nd_timing")
        return f.value, s.value

    def set_symmetric(self, on=True):
        """The matrix is symmetric (indefinite is fine): L D L^T in LU clothing, half the flops of the factorisation (include/pgx_nd.h:
        pgx_nd_set_symmetric).  Returns whether the handle honours the request (single rank, default schedule)."""
        self._check(self._lib.pgx_nd_set_symmetric(self._h, int(bool(on))), "pgx_nd_set_symmetric")
        return bool(self._lib.pgx_nd_is_symmetric(self._h))

    def depth_profile(self, enable=None):
        """Device time per TREE DEPTH (0 = root) of the factorisations / forward sweeps / backward sweeps since recording was switched
        on (include/pgx_nd.h: pgx_nd_depth_profile).  enable=True starts (and clears), False stops, None only reads.  Returns
        {"factor_ms": [...], "fwd_ms": [...], "bwd_ms": [...], "calls": (factorisations, forward sweeps, backward sweeps)}."""
        n = C.c_int32(0)
        self._check(self._lib.pgx_nd_depth_profile(self._h, -1, C.byref(n), None, None, None, None), "pgx_nd_depth_profile")
        nd = n.value
        f, fw, bw = (np.zeros(nd) for _ in range(3))
        calls = np.zeros(3, dtype=np.int32)
        n = C.c_int32(nd)
        flag = -1 if enable is None else int(bool(enable))
        self._check(self._lib.pgx_nd_depth_profile(self._h, flag, C.byref(n), f.ctypes.data_as(L.c_double_p), fw.ctypes.data_as(L.c_double_p),
                                                   bw.ctypes.data_as(L.c_double_p), calls.ctypes.data_as(L.c_int32_p)), "pgx_nd_depth_profile")
        return {"factor_ms": f, "fwd_ms": fw, "bwd_ms": bw, "calls": tuple(int(c) for c in calls)}

    def export_symbolic(self) -> dict:
        """The level/front/destination maps the device numeric phase uses (tests emulate it with numpy)."""
        lib, h = self._lib, self._h
        i64, i32 = np.int64, np.int32
        p64 = lambda a: a.ctypes.data_as(L.c_int64_p)  # noqa: E731
        nl = C.c_int64()
        self._check(lib.pgx_nd_export_levels(h, C.byref(nl), None, None, None, None, None), "export_levels")
        nl = nl.value
        lev_start, P, B, lev_off = np.zeros(nl + 1, i64), np.zeros(nl, i32), np.zeros(nl, i32), np.zeros(nl, i64)
        depth = np.zeros(nl, i32)
        n1 = C.c_int64()
        lib.pgx_nd_export_levels(h, C.byref(n1), p64(lev_start), L.iptr(P), L.iptr(B), p64(lev_off), L.iptr(depth))
        nf = C.c_int64()
        lib.pgx_nd_export_fronts(h, C.byref(nf), None, None, None, None, None, None, None, None)
        nf = nf.value
        fp, fb, par, s01 = (np.zeros(nf, i32) for _ in range(4))
        dof_ptr, rel_ptr = np.zeros(nf + 1, i64), np.zeros(nf + 1, i64)
        lib.pgx_nd_export_fronts(h, C.byref(n1), L.iptr(fp), L.iptr(fb), L.iptr(par), L.iptr(s01), p64(dof_ptr), None,
                                 p64(rel_ptr), None)
        own, rel = np.zeros(max(int(dof_ptr[-1]), 1), i32), np.zeros(max(int(rel_ptr[-1]), 1), i32)
        lib.pgx_nd_export_fronts(h, C.byref(n1), None, None, None, None, None, L.iptr(own), None, L.iptr(rel))
        nnz = C.c_int64()
        lib.pgx_nd_export_dest(h, C.byref(nnz), None)
        dest = np.zeros(nnz.value, i64)
        lib.pgx_nd_export_dest(h, C.byref(nnz), p64(dest))
        kd, kb, rs = C.c_int32(), C.c_int32(), C.c_int32()
        gs = np.full(max(self._dist_size, 1), -1, i32)
        lib.pgx_nd_export_dist(h, C.byref(kd), C.byref(kb), C.byref(rs), L.iptr(gs))
        dist = dict(kdist=kd.value, kbatch=kb.value, root_slot=rs.value, ghost_slot=gs)
        return dict(dist=dist, lev_start=lev_start, P=P, B=B, lev_off=lev_off, depth=depth, fp=fp, fb=fb, parent=par, slot01=s01,
                    dof_ptr=dof_ptr, own_dofs=own[: int(dof_ptr[-1])], rel_ptr=rel_ptr, rel=rel[: int(rel_ptr[-1])],
                    dest=dest)

    def close(self):
        if self._h:
            self._lib.pgx_nd_destroy(self._h)
            self._h = L._H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
