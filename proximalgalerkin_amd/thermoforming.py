"""Example 05 - thermoforming quasi-variational inequality (membrane u, mould temperature T, latent psi) - on the HIP
backend.  Host-side mirror of /root/reference/examples/05_obstacle_type_qvi/thermoforming_dolfinx.py, a flat script:
`solve_problem` runs its LVPP loop (:117-158) and `ThermoformingProblem` stands where the script builds
`dolfinx.fem.petsc.NonlinearProblem(F, u=s, bcs=[bc], J=J, petsc_options=sp)` (:114-116) with the MODIFIED Jacobian
(:69-71) and the `bt` line search of order 2 (:100-113).  Everything below `.solve()` runs in libpgx.so (include/pgx_qvi.h).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib, fem
from .problem import ConvergenceError, _SNES

# the reference's solver parameters (:102-113)
SP = {"snes_type": "newtonls", "snes_linesearch_type": "bt", "pc_type": "lu", "snes_atol": 1e-5, "snes_rtol": 1e-5,
      "snes_stol": 10 * np.finfo(np.float64).eps, "snes_linesearch_order": 2}


class ThermoformingProblem:
    """x = [u | T | psi], each P1 on `mesh`."""

    def __init__(self, mesh: fem.Mesh, petsc_options: dict | None = None, beta=1.0, f=25.0, knee=0.01, eps_mod=1.0e-10,
                 quadrature_degree=6, device=0):
        self._lib = lib = _lib.load()
        self.mesh = mesh
        self.nv = mesh.num_vertices
        self.ndofs = 3 * self.nv
        pts, wts = fem.quadrature_rule("triangle", quadrature_degree)
        bc = np.ascontiguousarray(mesh.exterior_dofs(1), dtype=np.int32)  # :73-79
        self._keep = (mesh.geometry, mesh.cells, pts, wts, bc)
        pm = _lib.pgx_mesh(self.nv, mesh.num_cells, _lib.dptr(mesh.geometry), _lib.iptr(mesh.cells), 0, 0, None, 0)
        pp = _lib.pgx_qvi_problem(len(wts), _lib.dptr(pts), _lib.dptr(wts), float(beta), float(f), float(knee), float(eps_mod),
                                  len(bc), _lib.iptr(bc))
        self._h = C.c_void_p()
        rc = lib.pgx_qvi_create(C.byref(pm), C.byref(pp), int(device), C.byref(self._h))
        if rc:
            msg = lib.pgx_qvi_last_error(None)
            raise _lib.PgxError(f"pgx_qvi_create failed (code {rc}): {msg.decode() if msg else ''}")
        self._opts = _lib.pgx_snes_opts()
        lib.pgx_default_opts(C.byref(self._opts))
        for k, v in (petsc_options or SP).items():
            if k in ("snes_rtol", "snes_atol", "snes_stol"):
                setattr(self._opts, k, float(v))
            elif k == "snes_max_it":
                self._opts.snes_max_it = int(v)
            elif k == "snes_linesearch_type":
                if v not in ("bt", "none", "basic"):
                    raise NotImplementedError(f"snes_linesearch_type {v}")
                self._opts.linesearch = 1 if v == "bt" else 0
            elif k == "snes_linesearch_order" and int(v) != 2:
                raise NotImplementedError("bt line search: order 2 (quadratic) only")
            elif k == "snes_monitor":
                self._opts.monitor = max(self._opts.monitor, 1)
        self._opts.ksp_max_it = 6
        self.solver = _SNES(self._opts)

    def _check(self, rc, what):
        if rc:
            msg = self._lib.pgx_qvi_last_error(self._h)
            raise _lib.PgxError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")

    def get_state(self):
        x = np.empty(self.ndofs)
        self._check(self._lib.pgx_qvi_get_state(self._h, _lib.dptr(x)), "pgx_qvi_get_state")
        return x

    def set_state(self, x):
        self._check(self._lib.pgx_qvi_set_state(self._h, _lib.dptr(np.ascontiguousarray(x, dtype=np.float64))), "set_state")

    def set_prev(self, x):
        self._check(self._lib.pgx_qvi_set_prev(self._h, _lib.dptr(np.ascontiguousarray(x, dtype=np.float64))), "set_prev")

    def advance_prev(self):
        self._check(self._lib.pgx_qvi_advance_prev(self._h), "pgx_qvi_advance_prev")

    def set_alpha(self, a):
        self._check(self._lib.pgx_qvi_set_alpha(self._h, float(a)), "pgx_qvi_set_alpha")

    def solve(self):
        reason, its, lin = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self._lib.pgx_qvi_newton_solve(self._h, C.byref(self._opts), C.byref(reason), C.byref(its),
                                                   C.byref(lin)), "pgx_qvi_newton_solve")
        s = self.solver
        s._reason, s._its = reason.value, its.value
        s.ksp._its, s.ksp._reason = lin.value, (-3 if reason.value == -3 else 4)
        return reason.value, its.value

    def h1_increment(self):
        out = C.c_double(0)
        self._check(self._lib.pgx_qvi_h1_increment(self._h, C.byref(out)), "pgx_qvi_h1_increment")
        return out.value

    def residual(self, x=None):
        out = np.empty(self.ndofs)
        nrm = C.c_double(0)
        xx = None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        self._check(self._lib.pgx_qvi_residual(self._h, _lib.dptr(xx), _lib.dptr(out), C.byref(nrm)), "pgx_qvi_residual")
        return out, nrm.value

    def jacobian(self, x=None):
        import scipy.sparse as sp

        xx = None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        self._check(self._lib.pgx_qvi_jacobian_fill(self._h, _lib.dptr(xx)), "pgx_qvi_jacobian_fill")
        nr, nnz = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.pgx_qvi_csr_export(self._h, C.byref(nr), C.byref(nnz), None, None, None), "csr_export")
        rp, col, val = np.empty(nr.value + 1, np.int32), np.empty(nnz.value, np.int32), np.empty(nnz.value)
        self._check(self._lib.pgx_qvi_csr_export(self._h, None, None, _lib.iptr(rp), _lib.iptr(col), _lib.dptr(val)),
                    "csr_export")
        return sp.csr_matrix((val, col, rp), shape=(nr.value, nr.value))

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        self._check(self._lib.pgx_qvi_spmv(self._h, _lib.dptr(x), _lib.dptr(y)), "pgx_qvi_spmv")
        return y

    def profile(self, enable=True):
        ms = (C.c_double * 6)()
        self._check(self._lib.pgx_qvi_profile(self._h, int(enable), ms), "pgx_qvi_profile")
        return dict(zip(("residual", "jacobian", "lu_factor", "lu_solve", "spmv", "newton_total"), ms))

    def close(self):
        if self._h:
            self._lib.pgx_qvi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_problem(M: int = 150, alpha_0: float = 2.0**-6, alpha_max: float = 2.0**14, termination_tol: float = 1e-9,
                  max_lvpp_iterations: int = 100, verbose: bool = True, return_solution: bool = False, device: int = 0):
    """The script's body (:23-160): returns (num_iterations, H1 increments) [, final state]."""
    mesh = fem.create_unit_square(M, M)  # :24-25
    problem = ThermoformingProblem(mesh, device=device)
    x0 = np.zeros(problem.ndofs)
    x0[problem.nv:2 * problem.nv] = 1.0  # :119 initial guess for T
    problem.set_state(x0)
    alpha = alpha_0
    num_iterations, diffs = [], []
    i = 0
    for i in range(1, max_lvpp_iterations + 1):
        if verbose:
            print(f"LVPP iteration: {i} Alpha: {alpha}", flush=True)
        problem.set_alpha(alpha)
        problem.solve()  # :124
        num_its = problem.solver.getIterationNumber()
        converged_reason = problem.solver.getConvergedReason()
        if converged_reason <= 0:  # :127-128 (an assert in the reference)
            raise ConvergenceError(f"Solver did not converge with {converged_reason}")
        normed_diff = problem.h1_increment()  # :138-140
        if verbose:
            print(f"LVPP iteration {i}, Converged reason {converged_reason}",
                  f" Newton iterations {num_its} ||u-u_prev||_L2={normed_diff}", flush=True)
        num_iterations.append(num_its)
        diffs.append(normed_diff)
        if normed_diff < termination_tol:
            if verbose:
                print(f"Solver converged after {i} iterations", flush=True)
            break
        problem.advance_prev()  # :156
        alpha = min(alpha_max, alpha * 4)  # :157-158
    if verbose:
        print(f"Total number of LVPP iterations: {i}", flush=True)
        print(f"Total number of Newton iterations: {sum(num_iterations)}", flush=True)
    if return_solution:
        x = problem.get_state()
        problem.close()
        return num_iterations, diffs, x
    problem.close()
    return num_iterations, diffs
