"""Example 06 - gradient constraint |grad u| <= phi with a vector latent variable - on the HIP backend.

Host-side mirror of /root/reference/examples/06_gradient_constraints/gradient_constraint_dolfinx.py: `solve_problem`
keeps the reference's signature (:18-33) and loop (:168-205); `GradientConstraintProblem` stands where the script builds
`dolfinx.fem.petsc.NonlinearProblem(F, u=sol, bcs=bcs, petsc_options=...)` (:108-131) and exposes the calls the script
makes on it: `.solve()`, `.solver.getIterationNumber()`, `.solver.getConvergedReason()`,
`.solver.ksp.getConvergedReason()` (:179-183).  Everything below `.solve()` runs in libpgx.so (include/pgx_gc.h):
hand-written HIP assembly + the sparse direct solver of include/pgx_nd.h.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Callable, Literal

import numpy as np

from . import _lib, fem
from .problem import ConvergenceError, _SNES

AlphaScheme = Literal["constant", "linear", "doubling"]

# the reference's SNES options (:116-131); MUMPS-specific keys have no meaning here
PETSC_OPTIONS = {
    "snes_type": "newtonls", "ksp_type": "preonly", "pc_type": "lu", "snes_atol": 1e-9, "snes_rtol": 1e-9,
    "snes_stol": 1e-9, "snes_max_it": 20, "snes_error_if_not_converged": True, "snes_linesearch_type": "none",
}


def phi_default(x):
    return 0.1 + 0.2 * x[0] + x[1] * 0.4  # :291-292


def f_default(x):
    return 15 * np.sin(np.pi * x[0]) * np.sin(np.pi * x[0])  # :296-297


class GradientConstraintProblem:
    """Mixed space [P2, (P1)^2] on `mesh`; state layout x = [u (P2 dofs: vertices | edges) | psi_x | psi_y]."""

    def __init__(self, mesh: fem.Mesh, phi_func: Callable, f_func: Callable, petsc_options: dict | None = None,
                 quadrature_degree: int = 10, device: int = 0, comm=None):
        self._lib = lib = _lib.load()
        self.mesh = mesh
        U = fem.FunctionSpace(mesh, 2, 1)  # primal space (collapsed sub(0), :54)
        self.U = U
        self.n2, self.nv = U.block_size, mesh.num_vertices
        self.ndofs = self.n2 + 2 * self.nv
        xd = U.dof_coordinates()
        pts, wts = fem.quadrature_rule("triangle", quadrature_degree)  # :53
        phi = np.ascontiguousarray(phi_func(xd.T.copy()), dtype=np.float64)  # phi.interpolate, :55-56
        f = np.ascontiguousarray(f_func(xd.T.copy()), dtype=np.float64)  # :60-61
        bc = np.ascontiguousarray(mesh.exterior_dofs(2), dtype=np.int32)  # :63-69
        cd = U.cell_dofs()
        self._keep = (mesh.geometry, mesh.cells, cd, pts, wts, phi, f, bc)
        pm = _lib.pgx_mesh(mesh.num_vertices, mesh.num_cells, _lib.dptr(mesh.geometry), _lib.iptr(mesh.cells), 0, 0,
                           _lib.iptr(cd), self.n2)
        pp = _lib.pgx_gc_problem(len(wts), _lib.dptr(pts), _lib.dptr(wts), _lib.dptr(phi), _lib.dptr(f), len(bc),
                                 _lib.iptr(bc), None)
        self._h = C.c_void_p()
        if comm is None:
            rc = lib.pgx_gc_create(C.byref(pm), C.byref(pp), int(device), C.byref(self._h))
        else:  # one handle per GPU: replicated iterate, distributed sparse LU; every call below is collective
            self._comm = comm
            rc = lib.pgx_gc_create_dist(C.byref(pm), C.byref(pp), comm._c, int(device), C.byref(self._h))
        if rc:
            msg = lib.pgx_gc_last_error(None)
            raise _lib.PgxError(f"pgx_gc_create failed (code {rc}): {msg.decode() if msg else ''}")
        self._opts = _lib.pgx_snes_opts()
        lib.pgx_default_opts(C.byref(self._opts))
        self._flags = {"snes_error_if_not_converged": False}
        for k, v in (petsc_options or PETSC_OPTIONS).items():
            if k in ("snes_rtol", "snes_atol", "snes_stol"):
                setattr(self._opts, k, float(v))
            elif k == "snes_max_it":
                self._opts.snes_max_it = int(v)
            elif k == "ksp_rtol":
                self._opts.ksp_rtol = float(v)
            elif k == "snes_monitor":
                self._opts.monitor = max(self._opts.monitor, 1)
            elif k == "ksp_monitor":
                self._opts.monitor = 2
            elif k == "snes_error_if_not_converged":
                self._flags[k] = bool(v) if v is not None else True
            elif k == "snes_linesearch_type" and v not in ("none", "basic"):
                raise NotImplementedError(f"snes_linesearch_type {v}")
            elif k == "snes_type" and v != "newtonls":
                raise NotImplementedError(f"snes_type {v}")
        self._opts.ksp_max_it = 6
        self.solver = _SNES(self._opts)
        self.alpha = 1.0

    def _check(self, rc, what):
        if rc:
            msg = self._lib.pgx_gc_last_error(self._h)
            raise _lib.PgxError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")

    # -- state -------------------------------------------------------------------------------------------------
    def get_state(self):
        x = np.empty(self.ndofs)
        self._check(self._lib.pgx_gc_get_state(self._h, _lib.dptr(x)), "pgx_gc_get_state")
        return x

    def set_state(self, x):
        self._check(self._lib.pgx_gc_set_state(self._h, _lib.dptr(np.ascontiguousarray(x, dtype=np.float64))), "set_state")

    def set_prev(self, x):
        self._check(self._lib.pgx_gc_set_prev(self._h, _lib.dptr(np.ascontiguousarray(x, dtype=np.float64))), "set_prev")

    def advance_prev(self):
        """w0.x.array[:] = sol.x.array (:205), on the device"""
        self._check(self._lib.pgx_gc_advance_prev(self._h), "pgx_gc_advance_prev")

    def set_alpha(self, a):
        self.alpha = float(a)
        self._check(self._lib.pgx_gc_set_alpha(self._h, float(a)), "pgx_gc_set_alpha")

    # -- the call the script makes once per proximal step (:179) ------------------------------------------------------
    def solve(self):
        reason, its, lin = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self._lib.pgx_gc_newton_solve(self._h, C.byref(self._opts), C.byref(reason), C.byref(its),
                                                  C.byref(lin)), "pgx_gc_newton_solve")
        s = self.solver
        s._reason, s._its = reason.value, its.value
        s.ksp._its, s.ksp._reason = lin.value, (-3 if reason.value == -3 else 4)
        if reason.value <= 0 and self._flags["snes_error_if_not_converged"]:
            raise ConvergenceError(f"SNES did not converge: reason {reason.value} after {its.value} iterations")
        return reason.value, its.value

    def l2_increment(self):
        """sqrt(assemble_scalar(dot(u - u0, u - u0) dx)) (:164-166,184-186)"""
        out = C.c_double(0)
        self._check(self._lib.pgx_gc_l2_increment(self._h, C.byref(out)), "pgx_gc_l2_increment")
        return out.value

    def warm_start(self, device: int = 0):
        """The reference's --warm_start (:72-96): u <- solution of the Poisson problem (grad p, grad q) = (f, q), u = 0 on the
        boundary; psi <- 0.  At alpha = 1, psi = psi0 = 0 the u-block of the assembled Newton matrix IS that stiffness matrix
        (Dirichlet rows = identity) and the u-residual at u = 0 is -(f, q), so the pre-solve is one sparse-LU solve
        (include/pgx_nd.h) of that block with one refinement step - the reference's LinearProblem with pc_type lu."""
        from .direct import DirectSolver

        n2 = self.n2
        x0 = np.zeros(self.ndofs)
        alpha = self.alpha
        self.set_alpha(1.0)
        self.set_prev(x0)
        F, _ = self.residual(x0)
        K = self.jacobian(x0)[:n2, :n2].tocsr()
        K.sort_indices()
        self.set_alpha(alpha)
        lu = DirectSolver(K.indptr, K.indices, np.arange(n2, dtype=np.int32), self.U.dof_coordinates(), device=device)
        lu.factor(K.data)
        b = -F[:n2]
        u = lu.solve(b)
        u += lu.solve(b - K @ u)
        lu.close()
        x0[:n2] = u
        self.set_state(x0)
        return x0

    # -- fine-grained probes (tests) -----------------------------------------------------------------------------------
    def residual(self, x=None):
        out = np.empty(self.ndofs)
        nrm = C.c_double(0)
        xx = None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        self._check(self._lib.pgx_gc_residual(self._h, _lib.dptr(xx), _lib.dptr(out), C.byref(nrm)), "pgx_gc_residual")
        return out, nrm.value

    def jacobian(self, x=None):
        import scipy.sparse as sp

        xx = None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        self._check(self._lib.pgx_gc_jacobian_fill(self._h, _lib.dptr(xx)), "pgx_gc_jacobian_fill")
        nr, nnz = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.pgx_gc_csr_export(self._h, C.byref(nr), C.byref(nnz), None, None, None), "csr_export")
        rp, col, val = np.empty(nr.value + 1, np.int32), np.empty(nnz.value, np.int32), np.empty(nnz.value)
        self._check(self._lib.pgx_gc_csr_export(self._h, None, None, _lib.iptr(rp), _lib.iptr(col), _lib.dptr(val)),
                    "csr_export")
        return sp.csr_matrix((val, col, rp), shape=(nr.value, nr.value))

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        self._check(self._lib.pgx_gc_spmv(self._h, _lib.dptr(x), _lib.dptr(y)), "pgx_gc_spmv")
        return y

    def lu_stats(self) -> dict:
        st = _lib.pgx_nd_stats()
        self._check(self._lib.pgx_gc_lu_stats(self._h, C.byref(st)), "pgx_gc_lu_stats")
        return {k: getattr(st, k) for k, _ in st._fields_}

    def profile(self, enable=True):
        ms = (C.c_double * 6)()
        self._check(self._lib.pgx_gc_profile(self._h, int(enable), ms), "pgx_gc_profile")
        return dict(zip(("residual", "jacobian", "lu_factor", "lu_solve", "spmv", "newton_total"), ms))

    def close(self):
        if self._h:
            self._lib.pgx_gc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_problem(N: int, M: int, primal_space: str = "Lagrange", primal_degree: int = 2, cell_type: str = "triangle",
                  alpha_scheme: AlphaScheme = "doubling", alpha_0: float = 1.0, alpha_c: float = 1.0,
                  max_iterations: int = 25, stopping_tol: float = 1e-8, result_dir: Path | None = None,
                  phi_func: Callable = phi_default, f_func: Callable = f_default, warm_start: bool = False,
                  verbose: bool = True, return_solution: bool = False, device: int = 0, comm=None):
    """gradient_constraint_dolfinx.solve_problem (:18-205): returns (newton_iterations, L2_diff) [, final state]."""
    if primal_space not in ("Lagrange", "P", "CG") or primal_degree != 2 or cell_type != "triangle":
        raise NotImplementedError("HIP backend: primal Lagrange degree 2 on triangles (the reference's defaults)")
    mesh = fem.create_unit_square(N, M)  # :36
    problem = GradientConstraintProblem(mesh, phi_func, f_func, device=device, comm=comm)
    if warm_start:  # :72-96
        if comm is not None:
            raise NotImplementedError("warm_start with a distributed LU")
        problem.warm_start(device=device)
    if verbose:
        print(f"Number of dofs: {problem.n2}")  # :112
    newton_iterations = np.zeros(max_iterations, dtype=np.int32)
    L2_diff = np.zeros(max_iterations, dtype=np.float64)
    i = -1
    for i in range(max_iterations):
        if alpha_scheme == "constant":
            alpha = alpha_0
        elif alpha_scheme == "linear":
            alpha = alpha_0 + alpha_c * i
        elif alpha_scheme == "doubling":
            alpha = alpha_0 * 2**i
        else:
            raise ValueError(alpha_scheme)
        problem.set_alpha(alpha)
        problem.solve()  # :179
        num_newton_iterations = problem.solver.getIterationNumber()
        newton_iterations[i] = num_newton_iterations
        converged = problem.solver.getConvergedReason()
        ksp_converged = problem.solver.ksp.getConvergedReason()
        global_diff = problem.l2_increment()
        L2_diff[i] = global_diff
        if verbose:
            print(f"Iteration {i + 1}: {converged=} {num_newton_iterations=} {ksp_converged=}", f"|delta u |= {global_diff}")
        if global_diff < stopping_tol:
            break
        problem.advance_prev()  # :205
    out = (newton_iterations[: i + 1], L2_diff[: i + 1])
    if result_dir is not None:  # CSV fingerprint in place of the reference's VTX/XDMF output (I/O is out of scope)
        result_dir = Path(result_dir)
        result_dir.mkdir(parents=True, exist_ok=True)
        np.savetxt(result_dir / "lvpp_history.csv", np.stack(out, axis=1), delimiter=",", header="newton,L2_diff")
        from .io import write_vtu  # u (P2) for ParaView - the reference writes u.bp / grad_u.bp with VTXWriter (:145-158)

        xs = problem.get_state()
        write_vtu(result_dir / "u.vtu", problem.U.dof_coordinates(), problem.U.cell_dofs(), {"u": xs[: problem.n2]})
        nv, n2 = problem.nv, problem.n2
        write_vtu(result_dir / "psi.vtu", mesh.geometry, mesh.cells,
                  {"psi": np.stack([xs[n2: n2 + nv], xs[n2 + nv:]], axis=1)})
    if return_solution:
        x = problem.get_state()
        problem.close()
        return out + (x,)
    problem.close()
    return out
