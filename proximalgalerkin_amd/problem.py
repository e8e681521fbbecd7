"""Problem / solver classes with the call shapes of the reference, backed by libpgx.so.

    NonlinearProblem(F, u, bcs=, J=, petsc_options=, petsc_options_prefix=).solve()
        -> dolfinx.fem.petsc.NonlinearProblem as used at
           /root/reference/examples/01_obstacle_problem/obstacle_pg.py:140-142,190-192
    SNESProblem(F, u, J=None, bcs=None, ...) with .F(snes,x,F) / .J(snes,x,J,P);
    SNESSolver(problem, options).solve() -> (converged_reason, iterations)
        -> /root/reference/src/lvpp/problem.py:14-127

"UFL forms in" becomes a declarative form object (ObstacleResidual) because UFL cannot ship
(SURVEY.md section 8b); everything from `solve()` downward runs in hand-written HIP through the C ABI
of include/pgx.h.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import warnings
from dataclasses import dataclass

import numpy as np

from . import _lib
from .fem import Constant, DirichletBC, Function, QuadratureFunction

# PETSc SNESConvergedReason values
SNES_CONVERGED_FNORM_ABS = 2
SNES_CONVERGED_FNORM_RELATIVE = 3
SNES_CONVERGED_SNORM_RELATIVE = 4
SNES_DIVERGED_LINEAR_SOLVE = -3
SNES_DIVERGED_FNORM_NAN = -4
SNES_DIVERGED_MAX_IT = -5
SNES_DIVERGED_DTOL = -9


class ConvergenceError(RuntimeError):
    """Raised for snes_error_if_not_converged / ksp_error_if_not_converged (obstacle_pg.py:132,135)."""


@dataclass
class ObstacleResidual:
    """The residual form of obstacle_pg.py:116-124,

        F = alpha*inner(grad(u),grad(v))*dx + psi*v*dx + u*w*dx - exp(psi)*w*dx - phi*w*dx
            - alpha*f*v*dx - psi_k*v*dx,          dx with quadrature_degree (obstacle_pg.py:115)

    for sol=(u,psi), sol_k=(u_k,psi_k) in a mixed [P_k,P_k] space."""
    sol: Function
    sol_k: Function
    alpha: Constant
    f: Constant
    phi: QuadratureFunction
    quadrature_degree: int = 6


@dataclass
class Derivative:
    form: ObstacleResidual
    u: Function


def derivative(F, u, du=None):
    """ufl.derivative(F, sol) (obstacle_pg.py:125): the exact Jacobian [[alpha K, M],[M, -D(psi)]]."""
    return Derivative(F, u)


def _compile(F, u, J):
    """Symbolic forms (proximalgalerkin_amd.ufl) -> the family description the HIP path is built from; J must be the exact
    derivative of F with respect to u (obstacle_pg.py:125) or None (NonlinearProblem's default)."""
    from . import ufl

    if isinstance(F, ufl.Form):
        spec = ufl.compile_form(F, u, J)
        if not isinstance(spec, ObstacleResidual):
            raise NotImplementedError(f"this form is a {type(spec).__name__}: construct it with the family's own NonlinearProblem "
                                      "(proximalgalerkin_amd.thermoforming.NonlinearProblem)")
        return spec, (Derivative(spec, u) if J is not None else None)
    return F, J


_IGNORED_KEYS = {"pc_factor_mat_solver_type", "mat_mumps_icntl_14", "mat_mumps_icntl_24"}


def _parse_options(lib, options: dict | None):
    o = _lib.pgx_snes_opts()
    lib.pgx_default_opts(C.byref(o))
    flags = {"snes_error_if_not_converged": False, "ksp_error_if_not_converged": False}
    for key, v in (options or {}).items():
        if key in ("snes_rtol", "snes_atol", "snes_stol", "snes_divergence_tolerance"):
            setattr(o, "snes_divtol" if key == "snes_divergence_tolerance" else key, float(v))
        elif key == "snes_max_it":
            o.snes_max_it = int(v)
        elif key in ("ksp_rtol",):
            o.ksp_rtol = float(v)
        elif key == "ksp_max_it":
            o.ksp_max_it = int(v)
        elif key == "ksp_gmres_restart":
            o.ksp_restart = int(v)
        elif key in ("mg_nu", "pc_mg_smoothup", "pc_mg_smoothdown"):
            o.mg_nu = int(v)
        elif key == "mg_omega":
            o.mg_omega = float(v)
        elif key == "snes_type":
            if v != "newtonls":
                raise NotImplementedError(f"snes_type {v}")
        elif key == "snes_linesearch_type":
            if v not in ("none", "basic"):
                raise NotImplementedError(f"snes_linesearch_type {v}: only the full Newton step is implemented")
        elif key == "pc_type":
            # "lu" (what the reference passes, obstacle_pg.py:130) does NOT force the sparse direct solver: for P1 the
            # multigrid-preconditioned FGMRES reaches LU-level accuracy (ksp_rtol) an order of magnitude faster
            # (DESIGN.md); "pgx_lu" / "pgx_mg" select explicitly, anything else keeps the automatic choice
            o.pc_type = {"pgx_lu": 2, "pgx_mg": 1}.get(v, 0)
        elif key == "ksp_type":
            pass
        elif key in ("snes_monitor", "ksp_monitor"):
            o.monitor = max(o.monitor, 2 if key == "ksp_monitor" else 1)
        elif key in flags:
            flags[key] = bool(v) if v is not None else True
        elif key in _IGNORED_KEYS:
            warnings.warn(f"PETSc option {key!r} has no meaning for the HIP backend and is ignored", stacklevel=3)
        else:
            warnings.warn(f"unknown option {key!r} ignored", stacklevel=3)
    return o, flags


class _KSP:
    def __init__(self):
        self._reason, self._its = 0, 0

    def getConvergedReason(self):
        return self._reason

    def getIterationNumber(self):
        return self._its


class _SNES:
    """The handful of petsc4py.SNES methods the examples call on `problem.solver`."""

    def __init__(self, opts):
        self._o = opts
        self._reason, self._its = 0, 0
        self.ksp = _KSP()

    def getConvergedReason(self):
        return self._reason

    def getIterationNumber(self):
        return self._its

    def getLinearSolveIterations(self):
        return self.ksp._its

    def setTolerances(self, rtol=None, atol=None, stol=None, max_it=None):
        if rtol is not None:
            self._o.snes_rtol = float(rtol)
        if atol is not None:
            self._o.snes_atol = float(atol)
        if stol is not None:
            self._o.snes_stol = float(stol)
        if max_it is not None:
            self._o.snes_max_it = int(max_it)


class NonlinearProblem:
    def __init__(self, F, u: Function, bcs=None, J=None, petsc_options=None, petsc_options_prefix="", device=0, lu_comm=None):
        """lu_comm: a comm.Communicator -> this handle is one of several REPLICAS (whole mesh on every rank) whose sparse-LU
        preconditioner is distributed over the ranks (include/pgx.h: pgx_create_lu_dist); every call is collective."""
        F, J = _compile(F, u, J)
        if not isinstance(F, ObstacleResidual):
            raise TypeError("F must be a form (proximalgalerkin_amd.ufl) or an ObstacleResidual form description")
        if J is not None and not (isinstance(J, Derivative) and J.form is F):
            raise NotImplementedError("only J = derivative(F, u) (the exact Jacobian) is supported")
        if u is not F.sol:
            raise ValueError("u must be the unknown the form was written for")
        self.F_form, self.u, self.bcs = F, u, list(bcs or [])
        self._lib = lib = _lib.load()
        self._opts, self._flags = _parse_options(lib, petsc_options)
        self.solver = _SNES(self._opts)
        V = u.function_space
        mesh = V.mesh
        if V.ncomp != 2 or F.sol_k.function_space != V:
            raise ValueError("sol and sol_k must live in the same mixed [P_k,P_k] space")
        if F.phi.degree != F.quadrature_degree:
            raise ValueError("phi must be interpolated at the quadrature degree of the measure")
        bc_dofs = np.zeros(0, dtype=np.int32)
        bc_vals = np.zeros(0)
        for bc in self.bcs:
            if not isinstance(bc, DirichletBC) or bc.sub != 0:
                raise NotImplementedError("Dirichlet conditions are supported on sub(0) (u) only")
            bc_dofs = np.concatenate([bc_dofs, bc.dofs])
            bc_vals = np.concatenate([bc_vals, bc.values])
        self._keep = (mesh.geometry, mesh.cells, F.phi.points, F.phi.weights,
                      np.ascontiguousarray(F.phi.values), np.ascontiguousarray(bc_dofs, dtype=np.int32),
                      np.ascontiguousarray(bc_vals, dtype=np.float64))
        cdofs = V.cell_dofs() if V.degree == 2 else None
        self._keep_cd = cdofs
        pm = _lib.pgx_mesh(mesh.num_vertices, mesh.num_cells, _lib.dptr(self._keep[0]), _lib.iptr(self._keep[1]),
                           *(mesh.structured or (0, 0)), _lib.iptr(cdofs), V.block_size if V.degree == 2 else 0)
        pp = _lib.pgx_problem(V.degree, len(self._keep[3]), _lib.dptr(self._keep[2]), _lib.dptr(self._keep[3]),
                              _lib.dptr(self._keep[4]), F.f.value, len(self._keep[5]), _lib.iptr(self._keep[5]),
                              _lib.dptr(self._keep[6]))
        h = C.c_void_p()
        self.partition = part = getattr(mesh, "partition", None)
        if part is None and lu_comm is not None:
            if getattr(mesh, "curved", False):
                raise NotImplementedError("order-2 geometry runs on a single handle (pgx_create_curved); mesh.flattened() gives the affine cells")
            self._lu_comm = lu_comm
            rc = lib.pgx_create_lu_dist(C.byref(pm), C.byref(pp), lu_comm._c, int(device), C.byref(h))
            _lib.check(lib, None, rc, "pgx_create_lu_dist")
        elif part is None and getattr(mesh, "curved", False):
            # order-2 geometry (round 5): degree-2 fields = isoparametric P2, degree-1 fields = hat functions on the quadratic cells
            # (the reference's default run on its own meshes).  Weights and inverse Jacobians of the quadratic cell map at every
            # quadrature point (fem.Mesh.geometry_at); phi was interpolated at the curved cells' quadrature points
            self._keep_geo = mesh.geometry_at(self._keep[2])[1]
            rc = lib.pgx_create_curved(C.byref(pm), C.byref(pp), _lib.dptr(self._keep_geo), int(device), C.byref(h))
            _lib.check(lib, None, rc, "pgx_create_curved")
        elif part is None:
            rc = lib.pgx_create(C.byref(pm), C.byref(pp), int(device), C.byref(h))
            _lib.check(lib, None, rc, "pgx_create")
        else:  # this mesh is one rank's strip: every call below is collective over part.comm (include/pgx.h)
            pt = _lib.pgx_partition(part.rank, part.size, part.global_ny, part.dist_levels)
            rc = lib.pgx_create_sharded(C.byref(pm), C.byref(pp), C.byref(pt), part.comm._c, int(device), C.byref(h))
            _lib.check(lib, None, rc, "pgx_create_sharded")
        self._h = h
        self.ndofs = V.num_dofs
        F.sol.x._binding = (self, "state")
        F.sol_k.x._binding = (self, "prev")
        F.sol.x._dev_valid = F.sol_k.x._dev_valid = False

    # -- host <-> device synchronisation of the two bound Functions ---------------------------------
    def _pull(self, slot, out):
        fn = self._lib.pgx_get_state if slot == "state" else self._lib.pgx_get_prev
        _lib.check(self._lib, self._h, fn(self._h, _lib.dptr(out)), "pgx_get_" + slot)

    def _push(self, slot, vec):
        if not vec._dev_valid:
            fn = self._lib.pgx_set_state if slot == "state" else self._lib.pgx_set_prev
            _lib.check(self._lib, self._h, fn(self._h, _lib.dptr(vec._a)), "pgx_set_" + slot)
            vec._dev_valid = True

    def _advance_prev(self):
        _lib.check(self._lib, self._h, self._lib.pgx_advance_prev(self._h), "pgx_advance_prev")

    def zero_state(self):
        """sol = sol_k = 0 on the device, no PCIe traffic (obstacle_pg.py:157-158)."""
        _lib.check(self._lib, self._h, self._lib.pgx_zero_state(self._h), "pgx_zero_state")
        for v in (self.F_form.sol.x, self.F_form.sol_k.x):
            v._dev_valid, v._host_valid = True, False

    def _sync_inputs(self):
        self._push("state", self.F_form.sol.x)
        self._push("prev", self.F_form.sol_k.x)
        if self.partition is not None:
            # the owners' values win over whatever sits in the ghost entries (Vec.ghostUpdate(INSERT, FORWARD),
            # lvpp/problem.py:56).  UNCONDITIONAL, like the reference's: the exchange is collective, and whether a host
            # array was touched is a per-rank fact (`if rank == 0: print(sol.x.array...)`) - a rank-local test would
            # leave the other ranks out of the exchange and hang the run.  One 16-33 KB exchange per call.
            _lib.check(self._lib, self._h, self._lib.pgx_sync_ghosts(self._h), "pgx_sync_ghosts")
        _lib.check(self._lib, self._h, self._lib.pgx_set_alpha(self._h, float(self.F_form.alpha.value)),
                   "pgx_set_alpha")

    # -- the call the examples make once per proximal step (obstacle_pg.py:190) ---------------------
    def solve(self):
        self._sync_inputs()
        reason, its, lin = C.c_int(0), C.c_int(0), C.c_int(0)
        rc = self._lib.pgx_newton_solve(self._h, C.byref(self._opts), C.byref(reason), C.byref(its), C.byref(lin))
        _lib.check(self._lib, self._h, rc, "pgx_newton_solve")
        s = self.solver
        s._reason, s._its = reason.value, its.value
        s.ksp._its = lin.value
        s.ksp._reason = -3 if reason.value == SNES_DIVERGED_LINEAR_SOLVE else 2
        if reason.value > 0:
            self.u.x._host_valid = False  # device holds the new iterate; pulled on next `.x.array`
            self.u.x._dev_valid = True
        if reason.value == SNES_DIVERGED_LINEAR_SOLVE and self._flags["ksp_error_if_not_converged"]:
            raise ConvergenceError("KSP did not converge (DIVERGED_LINEAR_SOLVE)")
        if reason.value <= 0 and self._flags["snes_error_if_not_converged"]:
            raise ConvergenceError(f"SNES did not converge: reason {reason.value} after {its.value} iterations")
        return self.u

    # -- fine-grained probes (tests, parity checks) -----------------------------------------------
    def residual(self, x=None):
        """F(x) with the callback contract of lvpp/problem.py:54-67; x=None uses the current `sol`."""
        self._sync_inputs()
        out = np.empty(self.ndofs)
        nrm = C.c_double(0)
        xx = None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        rc = self._lib.pgx_residual(self._h, _lib.dptr(xx), _lib.dptr(out), C.byref(nrm))
        _lib.check(self._lib, self._h, rc, "pgx_residual")
        return out, nrm.value

    def assemble_jacobian(self, x=None):
        self._sync_inputs()
        xx = None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        _lib.check(self._lib, self._h, self._lib.pgx_jacobian_fill(self._h, _lib.dptr(xx)), "pgx_jacobian_fill")

    def export_blocks(self, with_D=True):
        """(rowptr, col, K, M, D) of the scalar pattern shared by the four Jacobian blocks."""
        nrows, nnz = C.c_int64(0), C.c_int64(0)
        lib = self._lib
        _lib.check(lib, self._h, lib.pgx_csr_export(self._h, C.byref(nrows), C.byref(nnz), None, None, None, None,
                                                    None), "pgx_csr_export")
        rowptr = np.empty(nrows.value + 1, dtype=np.int32)
        col = np.empty(nnz.value, dtype=np.int32)
        K, M = np.empty(nnz.value), np.empty(nnz.value)
        D = np.empty(nnz.value) if with_D else None
        _lib.check(lib, self._h, lib.pgx_csr_export(self._h, None, None, _lib.iptr(rowptr), _lib.iptr(col),
                                                    _lib.dptr(K), _lib.dptr(M), _lib.dptr(D)), "pgx_csr_export")
        return rowptr, col, K, M, D

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        _lib.check(self._lib, self._h, self._lib.pgx_spmv(self._h, _lib.dptr(x), _lib.dptr(y)), "pgx_spmv")
        return y

    def spmv_bench(self, reps=20):
        ms, by = C.c_double(0), C.c_double(0)
        _lib.check(self._lib, self._h, self._lib.pgx_spmv_bench(self._h, int(reps), C.byref(ms), C.byref(by)),
                   "pgx_spmv_bench")
        return ms.value, by.value

    def spmv_bench_cold(self, reps=20):
        """Like spmv_bench, every apply after a 512 MB sweep (operands from HBM, not the Infinity Cache): pgx_spmv_bench_cold."""
        ms, by = C.c_double(0), C.c_double(0)
        _lib.check(self._lib, self._h, self._lib.pgx_spmv_bench_cold(self._h, int(reps), C.byref(ms), C.byref(by)),
                   "pgx_spmv_bench_cold")
        return ms.value, by.value

    def spmv_select(self, kind=-1):
        """Select / query the operator-apply kernel of the Krylov solver (include/pgx.h: pgx_spmv_select): 1 matrix-free stencil
        (default on structured P1 meshes), 0 block-CSR stream, 2 generic stencil kernel; returns the kind that will run."""
        act = C.c_int(0)
        _lib.check(self._lib, self._h, self._lib.pgx_spmv_select(self._h, int(kind), C.byref(act)), "pgx_spmv_select")
        return act.value

    def p2_stencil_info(self):
        """(state, i0, ni, j0, nj) of the structured P2 operator apply (include/pgx.h: pgx_p2_stencil_info)."""
        out = (C.c_int32 * 5)()
        _lib.check(self._lib, self._h, self._lib.pgx_p2_stencil_info(self._h, out), "pgx_p2_stencil_info")
        return tuple(int(v) for v in out)

    def comm_counts(self, reset=False):
        """Collectives this rank issued since the last reset (include/pgx.h: pgx_comm_counts)."""
        out = (C.c_int64 * 4)()
        _lib.check(self._lib, self._h, self._lib.pgx_comm_counts(self._h, out, int(reset)), "pgx_comm_counts")
        return dict(zip(("halo_exchanges", "allreduces", "vcycles", "krylov_iterations"), (int(v) for v in out)))

    def smoother_bench(self, reps=20):
        """(avg ms, algorithmic bytes) of the finest level's fused smoother launch (include/pgx.h: pgx_smoother_bench)."""
        ms, by = C.c_double(0), C.c_double(0)
        _lib.check(self._lib, self._h, self._lib.pgx_smoother_bench(self._h, int(reps), C.byref(ms), C.byref(by)),
                   "pgx_smoother_bench")
        return ms.value, by.value

    def vcycle_bench(self, level, reps=50):
        """(avg ms, vertices of that level): the part of one V-cycle from multigrid level `level` down and back up, launches back
        to back (include/pgx.h: pgx_vcycle_bench; level -1 = the fused tail launch)."""
        ms, nl = C.c_double(0), C.c_int(0)
        _lib.check(self._lib, self._h, self._lib.pgx_vcycle_bench(self._h, int(level), int(reps), C.byref(ms), C.byref(nl)),
                   "pgx_vcycle_bench")
        return ms.value, nl.value

    def observables(self):
        """[energy, |complementarity|, feasibility, dual feasibility, H1 increment, latent L2 increment]
        of obstacle_pg.py:145-152,196-201 in one device pass."""
        self._sync_inputs()
        out = np.empty(6)
        _lib.check(self._lib, self._h, self._lib.pgx_observables(self._h, _lib.dptr(out)), "pgx_observables")
        return out

    def owned_edge_range(self):
        """(offset, count): owned EDGE dofs within each field block of a P2 local vector (include/pgx.h: pgx_owned_edge_range)."""
        off, cnt = C.c_int64(0), C.c_int64(0)
        _lib.check(self._lib, self._h, self._lib.pgx_owned_edge_range(self._h, C.byref(off), C.byref(cnt)),
                   "pgx_owned_edge_range")
        return off.value, cnt.value

    def owned_range(self):
        """(offset, count): owned VERTEX dofs of each field block of a local vector (all vertices for an unsharded mesh).  Degree 2:
        the field block is [vertex dofs | edge dofs] and this is the vertex part only - combine it with owned_edge_range()."""
        off, cnt = C.c_int64(0), C.c_int64(0)
        _lib.check(self._lib, self._h, self._lib.pgx_owned_range(self._h, C.byref(off), C.byref(cnt)),
                   "pgx_owned_range")
        return off.value, cnt.value

    def profile(self, enable=None, reset=False):
        if enable is not None:
            self._lib.pgx_profile_enable(self._h, int(enable))
        ms = np.empty(8)
        self._lib.pgx_profile_get(self._h, _lib.dptr(ms), int(reset))
        keys = ["residual", "jacobian_fill", "mg_setup", "spmv", "vcycle", "orthogonalisation", "observables",
                "newton_total"]
        return dict(zip(keys, ms.tolist()))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pgx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# lvpp twins (src/lvpp/problem.py)
# ------------------------------------------------------------------------------------------------
class SNESProblem:
    """lvpp.SNESProblem(F, u, J=None, bcs=None, form_compiler_options=None, jit_options=None)
    (src/lvpp/problem.py:14-52).  `.F` / `.J` keep the SNES callback signatures (problem.py:54,69);
    `x`, `F` are numpy arrays (or objects with `.array`)."""

    def __init__(self, F, u, J=None, bcs=None, form_compiler_options=None, jit_options=None, device=0):
        self.L = F
        self.a = J if J is not None else derivative(F, u)  # problem.py:37-49
        self.bcs = bcs
        self.u = u
        self._backend = NonlinearProblem(F, u, bcs=bcs, J=self.a, device=device)

    @staticmethod
    def _arr(v):
        return v.array if hasattr(v, "array") else v

    def F(self, snes, x, F):
        """Assemble the residual at x into F (problem.py:54-67)."""
        out, _ = self._backend.residual(self._arr(x))
        self._arr(F)[:] = out

    def J(self, snes, x, J, P):
        """Assemble the Jacobian at x into the backend's fixed pattern (problem.py:69-77)."""
        self._backend.assemble_jacobian(self._arr(x))


class SNESSolver:
    """lvpp.SNESSolver(problem, options).solve() -> (converged_reason, iterations)
    (src/lvpp/problem.py:80-127)."""

    def __init__(self, problem: SNESProblem, options: dict):
        self.problem = problem
        self.options = options
        b = problem._backend
        b._opts, b._flags = _parse_options(b._lib, options)
        b.solver = _SNES(b._opts)
        self._snes = b.solver

    def solve(self):
        b = self.problem._backend
        b.solve()  # copies back only if reason>0 (problem.py:121-123), inside pgx_newton_solve
        return self._snes.getConvergedReason(), self._snes.getIterationNumber()
