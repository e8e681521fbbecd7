"""ctypes binding of libpgx.so (C ABI: include/pgx.h).

The product path has NO CPU fallback: if the shared library is missing or no GPU is visible the
calls raise.  (The numpy oracle under oracle/ is test infrastructure and is never imported here.)
"""
from __future__ import annotations

import ctypes as C
import pathlib

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
import os as _os

LIB_PATH = pathlib.Path(_os.environ.get("PGX_LIB", _HERE / "libpgx.so"))  # PGX_LIB: A/B builds of the same library

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)


class PgxError(RuntimeError):
    pass


class pgx_mesh(C.Structure):
    _fields_ = [
        ("n_vertices", C.c_int32),
        ("n_cells", C.c_int32),
        ("coords", c_double_p),
        ("cells", c_int32_p),
        ("structured_nx", C.c_int32),
        ("structured_ny", C.c_int32),
        ("cell_dofs", c_int32_p),
        ("n_dofs", C.c_int32),
    ]


class pgx_problem(C.Structure):
    _fields_ = [
        ("degree", C.c_int32),
        ("nq", C.c_int32),
        ("qpts", c_double_p),
        ("qwts", c_double_p),
        ("phi_q", c_double_p),
        ("f", C.c_double),
        ("n_bc", C.c_int32),
        ("bc_dofs", c_int32_p),
        ("bc_vals", c_double_p),
    ]


class pgx_snes_opts(C.Structure):
    _fields_ = [
        ("snes_rtol", C.c_double),
        ("snes_atol", C.c_double),
        ("snes_stol", C.c_double),
        ("snes_divtol", C.c_double),
        ("snes_max_it", C.c_int32),
        ("ksp_rtol", C.c_double),
        ("ksp_max_it", C.c_int32),
        ("ksp_restart", C.c_int32),
        ("mg_nu", C.c_int32),
        ("mg_omega", C.c_double),
        ("monitor", C.c_int32),
        ("pc_type", C.c_int32),
        ("linesearch", C.c_int32),
    ]


class pgx_nd_matrix(C.Structure):  # include/pgx_nd.h
    _fields_ = [
        ("n", C.c_int64),
        ("rowptr", C.POINTER(C.c_int32)),
        ("col", C.POINTER(C.c_int32)),
        ("n_nodes", C.c_int32),
        ("node_of_dof", C.POINTER(C.c_int32)),
        ("dim", C.c_int32),
        ("node_coords", C.POINTER(C.c_double)),
        ("leaf_nodes", C.c_int32),
    ]


class pgx_nd_stats(C.Structure):
    _fields_ = [
        ("n_fronts", C.c_int64),
        ("n_levels", C.c_int64),
        ("max_front", C.c_int64),
        ("arena_doubles", C.c_int64),
        ("factor_nnz", C.c_int64),
        ("flops", C.c_double),
        ("flops_padded", C.c_double),
        ("perturbed_pivots", C.c_int64),
    ]


class pgx_gc_problem(C.Structure):  # include/pgx_gc.h
    _fields_ = [
        ("nq", C.c_int32),
        ("qpts", c_double_p),
        ("qwts", c_double_p),
        ("phi_dofs", c_double_p),
        ("f_dofs", c_double_p),
        ("n_bc", C.c_int32),
        ("bc_dofs", c_int32_p),
        ("bc_vals", c_double_p),
    ]


class pgx_gc_spaces(C.Structure):  # include/pgx_gc.h: general primal degree
    _fields_ = [
        ("n_vertices", C.c_int32),
        ("n_cells", C.c_int32),
        ("coords", c_double_p),
        ("cells", c_int32_p),
        ("nu", C.c_int32),
        ("np", C.c_int32),
        ("n_u", C.c_int32),
        ("n_p", C.c_int32),
        ("cell_dofs_u", c_int32_p),
        ("cell_dofs_p", c_int32_p),
        ("coords_u", c_double_p),
        ("coords_p", c_double_p),
        ("tab_Nu", c_double_p),
        ("tab_dNu", c_double_p),
        ("tab_Np", c_double_p),
    ]


class pgx_sg_mesh(C.Structure):  # include/pgx_sg.h
    _fields_ = [
        ("n_vertices", C.c_int32),
        ("n_cells", C.c_int32),
        ("coords", c_double_p),
        ("cells", c_int32_p),
        ("n_facets", C.c_int32),
        ("facets", c_int32_p),
        ("degree", C.c_int32),
        ("cell_type", C.c_int32),
    ]


class pgx_sg_problem(C.Structure):
    _fields_ = [
        ("E", C.c_double),
        ("nu", C.c_double),
        ("gap", C.c_double),
        ("nq", C.c_int32),
        ("qpts", c_double_p),
        ("qwts", c_double_p),
        ("n_bc", C.c_int32),
        ("bc_dofs", c_int32_p),
        ("bc_vals", c_double_p),
    ]


class pgx_sg_curved(C.Structure):  # include/pgx_sg.h: order-2 geometry of a mesh of 10-node tetrahedra
    _fields_ = [
        ("nq", C.c_int32),
        ("qpts", c_double_p),
        ("qwts", c_double_p),
        ("cell_geo", c_double_p),
        ("facet_geo", c_double_p),
    ]


class pgx_qvi_problem(C.Structure):  # include/pgx_qvi.h
    _fields_ = [
        ("nq", C.c_int32),
        ("qpts", c_double_p),
        ("qwts", c_double_p),
        ("beta", C.c_double),
        ("f", C.c_double),
        ("knee", C.c_double),
        ("eps_mod", C.c_double),
        ("n_bc", C.c_int32),
        ("bc_dofs", c_int32_p),
    ]


class pgx_ic_problem(C.Structure):  # include/pgx_ic.h
    _fields_ = [
        ("n_vertices", C.c_int32),
        ("x", c_double_p),
        ("nq", C.c_int32),
        ("qpts", c_double_p),
        ("qwts", c_double_p),
        ("phi0_q", c_double_p),
        ("phi_q", c_double_p),
        ("c", C.c_double),
        ("n_bc", C.c_int32),
        ("bc_dofs", c_int32_p),
    ]


class pgx_partition(C.Structure):
    _fields_ = [
        ("rank", C.c_int32),
        ("size", C.c_int32),
        ("global_ny", C.c_int32),
        ("dist_levels", C.c_int32),
    ]


# every symbol include/pgx.h declares: (name, restype, argtypes)
_H = C.c_void_p
_COMM = C.c_void_p
SYMBOLS = [
    ("pgx_default_opts", None, [C.POINTER(pgx_snes_opts)]),
    ("pgx_create", C.c_int, [C.POINTER(pgx_mesh), C.POINTER(pgx_problem), C.c_int, C.POINTER(_H)]),
    ("pgx_create_curved", C.c_int, [C.POINTER(pgx_mesh), C.POINTER(pgx_problem), c_double_p, C.c_int, C.POINTER(_H)]),
    ("pgx_destroy", None, [_H]),
    ("pgx_last_error", C.c_char_p, [_H]),
    ("pgx_num_dofs", C.c_int, [_H, C.POINTER(C.c_int64)]),
    ("pgx_set_state", C.c_int, [_H, c_double_p]),
    ("pgx_get_state", C.c_int, [_H, c_double_p]),
    ("pgx_set_prev", C.c_int, [_H, c_double_p]),
    ("pgx_get_prev", C.c_int, [_H, c_double_p]),
    ("pgx_advance_prev", C.c_int, [_H]),
    ("pgx_zero_state", C.c_int, [_H]),
    ("pgx_set_alpha", C.c_int, [_H, C.c_double]),
    ("pgx_residual", C.c_int, [_H, c_double_p, c_double_p, c_double_p]),
    ("pgx_jacobian_fill", C.c_int, [_H, c_double_p]),
    ("pgx_csr_export", C.c_int,
     [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int64), c_int32_p, c_int32_p, c_double_p, c_double_p, c_double_p]),
    ("pgx_spmv", C.c_int, [_H, c_double_p, c_double_p]),
    ("pgx_spmv_bench", C.c_int, [_H, C.c_int, c_double_p, c_double_p]),
    ("pgx_spmv_bench_cold", C.c_int, [_H, C.c_int, c_double_p, c_double_p]),
    ("pgx_tuning_set", C.c_int, [C.c_char_p, C.c_char_p]),
    ("pgx_smoother_bench", C.c_int, [_H, C.c_int, c_double_p, c_double_p]),
    ("pgx_vcycle_bench", C.c_int, [_H, C.c_int, C.c_int, c_double_p, C.POINTER(C.c_int)]),
    ("pgx_spmv_select", C.c_int, [_H, C.c_int, C.POINTER(C.c_int)]),
    ("pgx_p2_stencil_info", C.c_int, [_H, C.POINTER(C.c_int32)]),
    ("pgx_comm_counts", C.c_int, [_H, C.POINTER(C.c_int64), C.c_int]),
    ("pgx_newton_solve", C.c_int,
     [_H, C.POINTER(pgx_snes_opts), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("pgx_observables", C.c_int, [_H, c_double_p]),
    ("pgx_profile_enable", C.c_int, [_H, C.c_int]),
    ("pgx_profile_get", C.c_int, [_H, c_double_p, C.c_int]),
    # sharded path
    ("pgx_partition_rows", C.c_int,
     [C.POINTER(pgx_partition), c_int32_p, c_int32_p, c_int32_p, c_int32_p]),
    ("pgx_comm_rccl_unique_id", C.c_int, [C.c_char_p]),
    ("pgx_comm_rccl_init", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(_COMM)]),
    ("pgx_comm_local_group", C.c_int, [C.c_int, C.POINTER(_COMM)]),
    ("pgx_comm_shm_init", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(_COMM)]),
    ("pgx_comm_allreduce", C.c_int, [_COMM, c_double_p, C.c_uint64]),
    ("pgx_comm_halo", C.c_int, [_COMM, c_double_p, c_double_p] + [C.c_uint64] * 8),
    ("pgx_comm_gather0", C.c_int, [_COMM, c_double_p, C.c_uint64, c_double_p]),
    ("pgx_comm_scatter0", C.c_int, [_COMM, c_double_p, C.c_uint64, c_double_p]),
    ("pgx_comm_selfcheck", C.c_int, [_COMM, C.c_int, C.c_double]),
    ("pgx_comm_free", None, [_COMM]),
    ("pgx_comm_last_error", C.c_char_p, []),
    ("pgx_create_sharded", C.c_int,
     [C.POINTER(pgx_mesh), C.POINTER(pgx_problem), C.POINTER(pgx_partition), _COMM, C.c_int, C.POINTER(_H)]),
    ("pgx_create_lu_dist", C.c_int, [C.POINTER(pgx_mesh), C.POINTER(pgx_problem), _COMM, C.c_int, C.POINTER(_H)]),
    ("pgx_owned_range", C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("pgx_owned_edge_range", C.c_int, [_H, c_int64_p, c_int64_p]),
    ("pgx_sync_ghosts", C.c_int, [_H]),
    # sparse direct solver (include/pgx_nd.h)
    ("pgx_nd_create", C.c_int, [C.POINTER(pgx_nd_matrix), C.c_int, C.c_void_p, C.POINTER(_H)]),
    ("pgx_nd_create_dist", C.c_int, [C.POINTER(pgx_nd_matrix), _COMM, C.c_int, C.c_void_p, C.POINTER(_H)]),
    ("pgx_nd_create_symbolic_dist", C.c_int, [C.POINTER(pgx_nd_matrix), C.c_int, C.c_int, C.POINTER(_H)]),
    ("pgx_nd_export_dist", C.c_int, [_H, c_int32_p, c_int32_p, c_int32_p, c_int32_p]),
    ("pgx_nd_destroy", None, [_H]),
    ("pgx_nd_last_error", C.c_char_p, [_H]),
    ("pgx_nd_get_stats", C.c_int, [_H, C.POINTER(pgx_nd_stats)]),
    ("pgx_nd_factor", C.c_int, [_H, c_double_p, C.c_int]),
    ("pgx_nd_solve", C.c_int, [_H, c_double_p, c_double_p, C.c_int]),
    ("pgx_nd_timing", C.c_int, [_H, C.c_int, c_double_p, c_double_p]),
    ("pgx_nd_depth_profile", C.c_int, [_H, C.c_int, c_int32_p, c_double_p, c_double_p, c_double_p, c_int32_p]),
    ("pgx_nd_set_symmetric", C.c_int, [_H, C.c_int]),
    ("pgx_nd_is_symmetric", C.c_int, [_H]),
    ("pgx_nd_sym_tile_count", C.c_int, [C.c_int, C.c_int, C.c_int]),
    ("pgx_nd_sym_tile_at", None, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("pgx_nd_export_levels", C.c_int, [_H, c_int64_p, c_int64_p, c_int32_p, c_int32_p, c_int64_p, c_int32_p]),
    ("pgx_nd_export_fronts", C.c_int,
     [_H, c_int64_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_int64_p, c_int32_p, c_int64_p, c_int32_p]),
    ("pgx_nd_export_dest", C.c_int, [_H, c_int64_p, c_int64_p]),
    # example 06: gradient constraint, vector latent variable (include/pgx_gc.h)
    ("pgx_gc_create", C.c_int, [C.POINTER(pgx_mesh), C.POINTER(pgx_gc_problem), C.c_int, C.POINTER(_H)]),
    ("pgx_gc_create_general", C.c_int, [C.POINTER(pgx_gc_spaces), C.POINTER(pgx_gc_problem), C.c_int, C.POINTER(_H)]),
    ("pgx_gc_create_dist", C.c_int, [C.POINTER(pgx_mesh), C.POINTER(pgx_gc_problem), _COMM, C.c_int, C.POINTER(_H)]),
    ("pgx_gc_lu_stats", C.c_int, [_H, C.POINTER(pgx_nd_stats)]),
    ("pgx_gc_lu_is_symmetric", C.c_int, [_H]),
    ("pgx_gc_destroy", None, [_H]),
    ("pgx_gc_last_error", C.c_char_p, [_H]),
    ("pgx_gc_num_dofs", C.c_int, [_H, c_int64_p]),
    ("pgx_gc_set_state", C.c_int, [_H, c_double_p]),
    ("pgx_gc_get_state", C.c_int, [_H, c_double_p]),
    ("pgx_gc_set_prev", C.c_int, [_H, c_double_p]),
    ("pgx_gc_get_prev", C.c_int, [_H, c_double_p]),
    ("pgx_gc_advance_prev", C.c_int, [_H]),
    ("pgx_gc_set_alpha", C.c_int, [_H, C.c_double]),
    ("pgx_gc_residual", C.c_int, [_H, c_double_p, c_double_p, c_double_p]),
    ("pgx_gc_jacobian_fill", C.c_int, [_H, c_double_p]),
    ("pgx_gc_csr_export", C.c_int, [_H, c_int64_p, c_int64_p, c_int32_p, c_int32_p, c_double_p]),
    ("pgx_gc_spmv", C.c_int, [_H, c_double_p, c_double_p]),
    ("pgx_gc_newton_solve", C.c_int,
     [_H, C.POINTER(pgx_snes_opts), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("pgx_gc_l2_increment", C.c_int, [_H, c_double_p]),
    ("pgx_gc_profile", C.c_int, [_H, C.c_int, c_double_p]),
    # example 02: Signorini contact (include/pgx_sg.h)
    ("pgx_sg_create", C.c_int, [C.POINTER(pgx_sg_mesh), C.POINTER(pgx_sg_problem), C.c_int, C.POINTER(_H)]),
    ("pgx_sg_create_dist", C.c_int, [C.POINTER(pgx_sg_mesh), C.POINTER(pgx_sg_problem), _COMM, C.c_int, C.POINTER(_H)]),
    ("pgx_sg_create_curved", C.c_int, [C.POINTER(pgx_sg_mesh), C.POINTER(pgx_sg_problem), C.POINTER(pgx_sg_curved), C.c_int, C.POINTER(_H)]),
    ("pgx_sg_partition_info", C.c_int, [_H, c_int64_p, c_int64_p]),
    ("pgx_sg_lu_stats", C.c_int, [_H, C.POINTER(pgx_nd_stats)]),
    ("pgx_sg_lu_is_symmetric", C.c_int, [_H]),
    ("pgx_sg_destroy", None, [_H]),
    ("pgx_sg_last_error", C.c_char_p, [_H]),
    ("pgx_sg_num_dofs", C.c_int, [_H, c_int64_p, c_int64_p]),
    ("pgx_sg_contact_vertices", C.c_int, [_H, c_int32_p]),
    ("pgx_sg_set_state", C.c_int, [_H, c_double_p]),
    ("pgx_sg_get_state", C.c_int, [_H, c_double_p]),
    ("pgx_sg_set_prev", C.c_int, [_H, c_double_p]),
    ("pgx_sg_get_prev", C.c_int, [_H, c_double_p]),
    ("pgx_sg_advance_prev", C.c_int, [_H]),
    ("pgx_sg_set_alpha", C.c_int, [_H, C.c_double]),
    ("pgx_sg_residual", C.c_int, [_H, c_double_p, c_double_p, c_double_p]),
    ("pgx_sg_jacobian_fill", C.c_int, [_H, c_double_p]),
    ("pgx_sg_csr_export", C.c_int, [_H, c_int64_p, c_int64_p, c_int32_p, c_int32_p, c_double_p]),
    ("pgx_sg_spmv", C.c_int, [_H, c_double_p, c_double_p]),
    ("pgx_sg_newton_solve", C.c_int,
     [_H, C.POINTER(pgx_snes_opts), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("pgx_sg_u_increment", C.c_int, [_H, c_double_p]),
    ("pgx_sg_profile", C.c_int, [_H, C.c_int, c_double_p]),
    # example 05: thermoforming QVI (include/pgx_qvi.h)
    ("pgx_qvi_create", C.c_int, [C.POINTER(pgx_mesh), C.POINTER(pgx_qvi_problem), C.c_int, C.POINTER(_H)]),
    ("pgx_qvi_destroy", None, [_H]),
    ("pgx_qvi_last_error", C.c_char_p, [_H]),
    ("pgx_qvi_num_dofs", C.c_int, [_H, c_int64_p]),
    ("pgx_qvi_set_state", C.c_int, [_H, c_double_p]),
    ("pgx_qvi_get_state", C.c_int, [_H, c_double_p]),
    ("pgx_qvi_set_prev", C.c_int, [_H, c_double_p]),
    ("pgx_qvi_get_prev", C.c_int, [_H, c_double_p]),
    ("pgx_qvi_advance_prev", C.c_int, [_H]),
    ("pgx_qvi_set_alpha", C.c_int, [_H, C.c_double]),
    ("pgx_qvi_residual", C.c_int, [_H, c_double_p, c_double_p, c_double_p]),
    ("pgx_qvi_jacobian_fill", C.c_int, [_H, c_double_p]),
    ("pgx_qvi_csr_export", C.c_int, [_H, c_int64_p, c_int64_p, c_int32_p, c_int32_p, c_double_p]),
    ("pgx_qvi_spmv", C.c_int, [_H, c_double_p, c_double_p]),
    ("pgx_qvi_newton_solve", C.c_int,
     [_H, C.POINTER(pgx_snes_opts), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("pgx_qvi_h1_increment", C.c_int, [_H, c_double_p]),
    ("pgx_qvi_profile", C.c_int, [_H, C.c_int, c_double_p]),
    # example 08: intersecting constraints (include/pgx_ic.h)
    ("pgx_ic_create", C.c_int, [C.POINTER(pgx_ic_problem), C.c_int, C.POINTER(_H)]),
    ("pgx_ic_destroy", None, [_H]),
    ("pgx_ic_last_error", C.c_char_p, [_H]),
    ("pgx_ic_num_dofs", C.c_int, [_H, c_int64_p]),
    ("pgx_ic_set_state", C.c_int, [_H, c_double_p]),
    ("pgx_ic_get_state", C.c_int, [_H, c_double_p]),
    ("pgx_ic_set_prev", C.c_int, [_H, c_double_p]),
    ("pgx_ic_get_prev", C.c_int, [_H, c_double_p]),
    ("pgx_ic_advance_prev", C.c_int, [_H]),
    ("pgx_ic_set_alpha", C.c_int, [_H, C.c_double]),
    ("pgx_ic_set_phi", C.c_int, [_H, c_double_p]),
    ("pgx_ic_residual", C.c_int, [_H, c_double_p, c_double_p, c_double_p]),
    ("pgx_ic_jacobian_fill", C.c_int, [_H, c_double_p]),
    ("pgx_ic_csr_export", C.c_int, [_H, c_int64_p, c_int64_p, c_int32_p, c_int32_p, c_double_p]),
    ("pgx_ic_spmv", C.c_int, [_H, c_double_p, c_double_p]),
    ("pgx_ic_newton_solve", C.c_int,
     [_H, C.POINTER(pgx_snes_opts), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("pgx_ic_l2_increment", C.c_int, [_H, c_double_p]),
    ("pgx_ic_profile", C.c_int, [_H, C.c_int, c_double_p]),
]

_lib = None
_forwarded: dict = {}

# environment variables libpgx.so itself reads (documented run-time options, include/pgx.h); everything else named PGX_* that the
# library understands is a TUNING key and reaches it only through pgx_tuning_set
_RUNTIME_ENV = {"PGX_COMM_TIMEOUT", "PGX_ROCTX", "PGX_ND_THREADS", "PGX_LIB", "PGX_TUNING_FROM_ENV"}


def tuning_set(key: str, value=None):
    """Set (value=None: clear) one tuning / A-B / test switch of the library (include/pgx.h: pgx_tuning_set).  Switches are read
    when a handle is created (a few at every call), never from the environment."""
    lib = load()
    rc = lib.pgx_tuning_set(key.encode(), None if value is None else str(value).encode())
    if rc != 0:
        raise PgxError(f"pgx_tuning_set({key!r}) failed (code {rc})")


def _sync_tuning_from_env(lib):
    """Opt-in bridge for tools/ and tests/: with PGX_TUNING_FROM_ENV=1 the PGX_* variables of the environment are copied into the
    library's tuning table before each handle is created (and removed again when they disappear).  Without it - the default - the
    environment has no influence on the library's behaviour."""
    if _os.environ.get("PGX_TUNING_FROM_ENV") != "1":
        return
    now = {k: v for k, v in _os.environ.items() if k.startswith("PGX_") and k not in _RUNTIME_ENV}
    for k in list(_forwarded):
        if k not in now:
            lib.pgx_tuning_set(k.encode(), None)
            del _forwarded[k]
    for k, v in now.items():
        if _forwarded.get(k) != v:
            lib.pgx_tuning_set(k.encode(), v.encode())
            _forwarded[k] = v


def load():
    """Load libpgx.so (built by `make -C proximalgalerkin_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        _sync_tuning_from_env(_lib)
        return _lib
    if not LIB_PATH.exists():
        raise PgxError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)")
    lib = C.CDLL(str(LIB_PATH))
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    _sync_tuning_from_env(lib)
    return lib


def dptr(a: np.ndarray | None):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(c_double_p)


def iptr(a: np.ndarray | None):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(c_int32_p)


def check(lib, handle, rc: int, what: str):
    if rc != 0:
        msg = lib.pgx_last_error(handle)
        raise PgxError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
