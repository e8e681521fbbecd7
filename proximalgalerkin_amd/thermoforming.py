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
        if getattr(mesh, "curved", False):  # order-2 geometry is built for examples 01 and 02 only
            raise NotImplementedError("example 05 integrates on affine cells: pass mesh.flattened()")
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


class NonlinearProblem:
    """`dolfinx.fem.petsc.NonlinearProblem(F, u=s, bcs=[bc], J=J, petsc_options=sp)` (:114-116) for FORMS of the thermoforming
    family: the UFL-subset front end (ufl.py) recognises the residual (:62-67) and the modified Jacobian form (:69-71), reads
    beta, f, the knees of g and eps off the forms' Constants and selects the HIP kernels of include/pgx_qvi.h.  `alpha` and the
    two Functions stay live: every `.solve()` takes their current values (:119,156-158)."""

    def __init__(self, F, u: fem.Function, bcs=None, J=None, petsc_options=None, petsc_options_prefix="", device=0):
        from . import ufl

        spec = ufl.compile_form(F, u, J)
        if not isinstance(spec, ufl.ThermoformingSpec):
            raise NotImplementedError(f"this form is a {type(spec).__name__}, not the thermoforming QVI")
        V = u.function_space
        if V.degree != 1 or spec.s_prev.function_space != V:
            raise NotImplementedError("the thermoforming kernels are written for the mixed [P1, P1, P1] space (:28-33)")
        if spec.bound0.value != 0.0 or not spec.bound1.value > 0.0:
            raise NotImplementedError("g is implemented with knees 0 < bound1 (:36-48)")
        ext = np.sort(V.mesh.exterior_dofs(1))
        bcs = list(bcs or [])
        ok = (len(bcs) == 1 and bcs[0].sub == 0 and np.array_equal(np.sort(bcs[0].dofs), ext) and not np.any(bcs[0].values))
        if not ok:
            raise NotImplementedError("boundary conditions: u = 0 on the whole boundary (:73-79)")
        self.spec, self.u = spec, u
        self._p = ThermoformingProblem(V.mesh, petsc_options, beta=spec.beta.value, f=spec.f.value, knee=spec.bound1.value,
                                       eps_mod=spec.eps.value if spec.eps is not None else 0.0,
                                       quadrature_degree=spec.quadrature_degree or 6, device=device)
        self.solver = self._p.solver

    def solve(self):
        p, sp = self._p, self.spec
        p.set_alpha(sp.alpha.value)
        p.set_prev(sp.s_prev.x.array)
        p.set_state(self.u.x.array)
        p.solve()
        self.u.x.array[:] = p.get_state()
        return self.u

    def h1_increment(self):
        """sqrt(assemble_scalar(inner(u-u_prev, u-u_prev)*dx + inner(grad(u-u_prev), grad(u-u_prev))*dx)) (:81-83,139-140)
        for the Functions' current values."""
        self._p.set_prev(self.spec.s_prev.x.array)
        self._p.set_state(self.u.x.array)
        return self._p.h1_increment()

    def close(self):
        self._p.close()


def solve_problem_forms(M: int = 150, alpha_0: float = 2.0**-6, alpha_max: float = 2.0**14, termination_tol: float = 1e-9,
                        max_lvpp_iterations: int = 100, verbose: bool = False, device: int = 0):
    """The reference script with its problem stated as forms (:23-160), through the front end.  Returns
    (num_iterations, final state)."""
    from . import ufl
    from .ufl import conditional, dx, exp, grad, inner, lt, max_value, pi, sin

    mesh = fem.create_unit_square(M, M)  # :24-25
    V = fem.functionspace(mesh, ("Lagrange", 1), ncomp=3)  # :28-30
    s = fem.Function(V)
    u, T, psi = ufl.split(s)  # :31
    v, q, w = ufl.TestFunctions(V)  # :33
    bound0, bound1 = fem.Constant(mesh, 0.0), fem.Constant(mesh, 0.01)  # :36-39

    def g(t):  # :42-48
        return conditional(lt(t, bound0), 1, conditional(lt(t, bound1), 1 - t / bound1, 0))

    x, y = ufl.SpatialCoordinate(mesh)  # :51
    s_prev = fem.Function(V)
    u_prev, _, psi_prev = ufl.split(s_prev)
    beta, alpha, f = fem.Constant(mesh, 1.0), fem.Constant(mesh, alpha_0), fem.Constant(mesh, 25)  # :55-57
    Phi0 = 1 - 2 * max_value(abs(x - 0.5), abs(y - 0.5))  # :58
    xi = sin(pi * x) * sin(pi * y)  # :59
    F = alpha * inner(grad(u), grad(v)) * dx + inner(psi, v) * dx  # :62-67
    F += -alpha * inner(f, v) * dx - inner(psi_prev, v) * dx
    F += inner(grad(T), grad(q)) * dx + beta * inner(T, q) * dx
    F += -inner(g(exp(-psi)), q) * dx
    F += inner(u, w) * dx + inner(exp(-psi), w) * dx
    F += -inner(Phi0 + xi * T, w) * dx
    eps = fem.Constant(mesh, 1.0e-10)  # :70
    J = ufl.derivative(F - eps / alpha * inner(grad(psi), grad(w)) * dx, s)  # :71
    bc = fem.dirichletbc(0.0, mesh.exterior_dofs(1), V.sub(0))  # :73-79
    problem = NonlinearProblem(F, u=s, bcs=[bc], J=J, petsc_options=dict(SP), petsc_options_prefix="snes_", device=device)
    n = V.block_size
    s.x.array[n:2 * n] = 1.0  # :119
    num_iterations = []
    for i in range(1, max_lvpp_iterations + 1):
        problem.solve()  # :124
        num_its = problem.solver.getIterationNumber()
        converged_reason = problem.solver.getConvergedReason()
        if converged_reason <= 0:  # :127-128
            raise ConvergenceError(f"Solver did not converge with {converged_reason}")
        normed_diff = problem.h1_increment()  # :138-140
        if verbose:
            print(f"LVPP iteration {i}, Converged reason {converged_reason}",
                  f" Newton iterations {num_its} ||u-u_prev||_L2={normed_diff}", flush=True)
        num_iterations.append(num_its)
        if normed_diff < termination_tol:
            break
        s_prev.x.array[:] = s.x.array  # :156
        alpha.value = min(alpha_max, alpha.value * 4)  # :157-158
    x_final = s.x.array.copy()
    problem.close()
    return num_iterations, x_final
