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
    """Mixed space [P_k, (P_(k-1))^2] on `mesh` (k = `degree`, 2 by default as in the reference, up to 8); state layout
    x = [u (primal dofs) | psi_x | psi_y].  k = 2 runs the specialised kernels of include/pgx_gc.h, k >= 3 (or `general=True`) the
    table-driven ones (pgx_gc_create_general) with the Lagrange tables of proximalgalerkin_amd/lagrange.py."""

    def __init__(self, mesh: fem.Mesh, phi_func: Callable, f_func: Callable, petsc_options: dict | None = None,
                 quadrature_degree: int = 10, device: int = 0, comm=None, degree: int = 2, general: bool = False):
        if getattr(mesh, "curved", False):  # order-2 geometry is built for examples 01 and 02 only
            raise NotImplementedError("example 06 integrates on affine cells: pass mesh.flattened()")
        self._lib = lib = _lib.load()
        self.mesh = mesh
        self.degree = k = int(degree)
        if not 2 <= k <= 8:
            raise NotImplementedError("primal degree 2..8 (gradient_constraint_dolfinx.py:245-250)")
        quad = mesh.cell_name() == "quadrilateral"  # --cell_type quadrilateral (:229-236): Q_k / (Q_(k-1))^2 on the grid of rectangles
        pts, wts = fem.quadrature_rule(mesh.cell_name(), quadrature_degree)  # :53
        self._h = C.c_void_p()
        if quad:
            if comm is not None:
                raise NotImplementedError("quadrilateral cells with a distributed LU")
            from . import lagrange

            general = True
            self.U = None
            self.n2, cd, xd = lagrange.numbering_quad(mesh, k)
            self.nv, cdp, xdp = lagrange.numbering_quad(mesh, k - 1)
            bc = lagrange.exterior_dofs_quad(mesh, k)
            Nu, dNu = lagrange.tabulate_quad(k, pts)
            Npl, _ = lagrange.tabulate_quad(k - 1, pts)
            corners = mesh.affine_corners
        elif k == 2 and not general:
            U = fem.FunctionSpace(mesh, 2, 1)  # primal space (collapsed sub(0), :54)
            self.U = U
            self.n2, self.nv = U.block_size, mesh.num_vertices
            xd = U.dof_coordinates()
            cd = U.cell_dofs()
            bc = np.ascontiguousarray(mesh.exterior_dofs(2), dtype=np.int32)  # :63-69
        else:
            if comm is not None:
                raise NotImplementedError("general degree with a distributed LU")
            from . import lagrange

            self.U = None
            self.n2, cd, xd = lagrange.numbering(mesh, k)
            self.nv, cdp, xdp = lagrange.numbering(mesh, k - 1)
            bc = lagrange.exterior_dofs(mesh, k, cd)
            Nu, dNu = lagrange.tabulate(k, pts)
            Npl, _ = lagrange.tabulate(k - 1, pts)
            corners = mesh.cells
        self.dof_coords = xd
        self.ndofs = self.n2 + 2 * self.nv
        # phi.interpolate(phi_func), f.interpolate(f_func) (:55-61); arrays of nodal values are taken as they are (forms front end)
        phi = np.ascontiguousarray(phi_func(xd.T.copy()) if callable(phi_func) else phi_func, dtype=np.float64)
        f = np.ascontiguousarray(f_func(xd.T.copy()) if callable(f_func) else f_func, dtype=np.float64)
        if phi.shape != (self.n2,) or f.shape != (self.n2,):
            raise ValueError("phi and f must be given in the collapsed primal space (one value per primal dof)")
        pp = _lib.pgx_gc_problem(len(wts), _lib.dptr(pts), _lib.dptr(wts), _lib.dptr(phi), _lib.dptr(f), len(bc),
                                 _lib.iptr(bc), None)
        if k == 2 and not general:
            self._keep = (mesh.geometry, mesh.cells, cd, pts, wts, phi, f, bc)
            pm = _lib.pgx_mesh(mesh.num_vertices, mesh.num_cells, _lib.dptr(mesh.geometry), _lib.iptr(mesh.cells), 0, 0,
                               _lib.iptr(cd), self.n2)
            if comm is None:
                rc = lib.pgx_gc_create(C.byref(pm), C.byref(pp), int(device), C.byref(self._h))
            else:  # one handle per GPU: replicated iterate, distributed sparse LU; every call below is collective
                self._comm = comm
                rc = lib.pgx_gc_create_dist(C.byref(pm), C.byref(pp), comm._c, int(device), C.byref(self._h))
        else:
            self._keep = (mesh.geometry, corners, cd, cdp, xd, xdp, Nu, dNu, Npl, pts, wts, phi, f, bc)
            sp = _lib.pgx_gc_spaces(mesh.num_vertices, mesh.num_cells, _lib.dptr(mesh.geometry), _lib.iptr(corners), cd.shape[1],
                                    cdp.shape[1], self.n2, self.nv, _lib.iptr(cd), _lib.iptr(cdp), _lib.dptr(xd), _lib.dptr(xdp),
                                    _lib.dptr(Nu), _lib.dptr(dNu), _lib.dptr(Npl))
            rc = lib.pgx_gc_create_general(C.byref(sp), C.byref(pp), int(device), C.byref(self._h))
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
        lu = DirectSolver(K.indptr, K.indices, np.arange(n2, dtype=np.int32), self.dof_coords, device=device)
        lu.set_symmetric(True)  # the stiffness matrix with identity Dirichlet rows and columns
        lu.factor(K.data)
        b = -F[:n2]
        u = lu.solve(b)
        u += lu.solve(b - K @ u)
        lu.close()
        rel = np.linalg.norm(b - K @ u) / max(np.linalg.norm(b), 1e-300)
        if not rel <= 1e-10:  # the reference's LinearProblem runs with ksp_error_if_not_converged
            raise RuntimeError(f"warm start: the Poisson pre-solve left a relative residual of {rel:.2e}")
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
        out = {k: getattr(st, k) for k, _ in st._fields_}
        out["symmetric"] = bool(self._lib.pgx_gc_lu_is_symmetric(self._h))  # L D L^T in LU clothing: about half of `flops` executed
        return out

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
    if primal_space not in ("Lagrange", "P", "CG") or not 2 <= primal_degree <= 8 or cell_type not in ("triangle", "quadrilateral"):
        raise NotImplementedError("HIP backend: primal Lagrange degree 2..8 on triangles or quadrilaterals (:229-250)")
    mesh = fem.create_unit_square(N, M, cell_type)  # :36
    problem = GradientConstraintProblem(mesh, phi_func, f_func, device=device, comm=comm, degree=primal_degree)
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
        nv, n2, nvert = problem.nv, problem.n2, mesh.num_vertices
        if mesh.cell_name() == "quadrilateral":  # vertex values: every k-th point of the Q_k lattice, every (k-1)-th of the latent one
            nx, ny = mesh.structured
            k = problem.degree

            def at_vertices(vals, d):
                return vals.reshape(d * ny + 1, d * nx + 1)[::d, ::d].ravel()

            tri = mesh.triangulated()
            write_vtu(result_dir / "u.vtu", mesh.geometry, tri, {"u": at_vertices(xs[:n2], k)})
            write_vtu(result_dir / "psi.vtu", mesh.geometry, tri,
                      {"psi": np.stack([at_vertices(xs[n2: n2 + nv], k - 1), at_vertices(xs[n2 + nv:], k - 1)], axis=1)})
        elif problem.U is not None:
            write_vtu(result_dir / "u.vtu", problem.U.dof_coordinates(), problem.U.cell_dofs(), {"u": xs[: problem.n2]})
        else:  # general degree: the vertex values (the first dofs of every Lagrange space here)
            write_vtu(result_dir / "u.vtu", mesh.geometry, mesh.cells, {"u": xs[:nvert]})
        if mesh.cell_name() != "quadrilateral":
            write_vtu(result_dir / "psi.vtu", mesh.geometry, mesh.cells,
                      {"psi": np.stack([xs[n2: n2 + nvert], xs[n2 + nv: n2 + nv + nvert]], axis=1)})
    if return_solution:
        x = problem.get_state()
        problem.close()
        return out + (x,)
    problem.close()
    return out


class NonlinearProblem:
    """dolfinx.fem.petsc.NonlinearProblem(F, u=sol, bcs=bcs, petsc_options=..., petsc_options_prefix="pg_") as
    gradient_constraint_dolfinx.py:113-132 builds it, for the residual FORM of :100-107 stated in proximalgalerkin_amd.ufl: the
    front end recognises the gradient-constraint family (vector latent variable) and reads alpha, phi, f and the previous
    iterate off the form; `.solve()` and `.solver` behave like the reference's.  sol / w0 are host Functions, synchronised with
    the device state around every solve (this is the reference's own usage: `w0.x.array[:] = sol.x.array`, :205)."""

    def __init__(self, F, u: fem.Function, bcs=None, J=None, petsc_options=None, petsc_options_prefix="", device=0):
        from . import ufl

        spec = ufl.compile_form(F, u, J)
        if not isinstance(spec, ufl.GradientConstraintSpec):
            raise NotImplementedError(f"this form is a {type(spec).__name__}, not the gradient-constraint family")
        V = u.function_space
        els = V.elements
        if not (els[0].degree == 2 and els[1].degree == 1 and els[1].shape == (2,)):
            raise NotImplementedError("HIP backend: primal Lagrange degree 2 with a vector P1 latent variable (the reference's defaults)")
        bc_dofs = np.zeros(0, dtype=np.int64)
        for bc in bcs or []:
            if bc.sub != 0 or np.any(bc.values != 0.0):
                raise NotImplementedError("homogeneous Dirichlet data on sub(0) (gradient_constraint_dolfinx.py:109-110)")
            bc_dofs = np.union1d(bc_dofs, bc.dofs)
        if not np.array_equal(bc_dofs, np.sort(V.mesh.exterior_dofs(2))):
            raise NotImplementedError("the Dirichlet dofs must be the exterior dofs of the primal space (:63-69)")
        self.spec, self.u = spec, u
        self._p = GradientConstraintProblem(V.mesh, spec.phi.x.array.copy(), spec.f.x.array.copy(), petsc_options=petsc_options,
                                            quadrature_degree=spec.quadrature_degree, device=device)
        self.solver = self._p.solver

    def solve(self):
        p, sp = self._p, self.spec
        p.set_alpha(sp.alpha.value)
        p.set_state(sp.sol.x.array)
        p.set_prev(sp.w0.x.array)
        reason, its = p.solve()
        if reason > 0:  # NonlinearProblem.solve copies back; SNESSolver keeps the old iterate otherwise (lvpp/problem.py:121-123)
            sp.sol.x.array[:] = p.get_state()
        return self.u

    def l2_increment(self):
        self._p.set_state(self.spec.sol.x.array)
        self._p.set_prev(self.spec.w0.x.array)
        return self._p.l2_increment()

    def close(self):
        self._p.close()


def solve_problem_forms(N: int, M: int, alpha_scheme: AlphaScheme = "doubling", alpha_0: float = 1.0, alpha_c: float = 1.0,
                        max_iterations: int = 25, stopping_tol: float = 1e-8, phi_func: Callable = phi_default,
                        f_func: Callable = f_default, device: int = 0):
    """gradient_constraint_dolfinx.solve_problem (:18-205) with the problem stated as the reference states it - elements, spaces,
    coefficient Functions and the residual FORM - through the UFL-subset front end.  Returns (newton_iterations, L2_diff, sol)."""
    from . import ufl

    mesh = fem.create_unit_square(N, M)  # :36
    el_0 = fem.element("Lagrange", mesh.cell_name(), 2)  # :38
    el_1 = fem.element("Lagrange", mesh.cell_name(), 1, shape=(2,))  # :40-42
    V_trial = fem.functionspace(mesh, fem.mixed_element([el_0, el_1]))  # :44-45
    sol = fem.Function(V_trial)
    u, psi = ufl.split(sol)  # :48-49
    v, w = ufl.TestFunctions(V_trial)  # :51
    dx = ufl.Measure("dx", domain=mesh, metadata={"quadrature_degree": 10})  # :53
    alpha = fem.Constant(mesh, alpha_0)
    U, U_to_W = V_trial.sub(0).collapse()  # :54
    phi = fem.Function(U)
    phi.interpolate(phi_func)
    w0 = fem.Function(V_trial)
    f = fem.Function(U)
    f.interpolate(f_func)
    boundary_dofs = mesh.exterior_dofs(2)  # :63-69
    _, psi0 = ufl.split(w0)  # :98
    F = alpha * ufl.inner(ufl.grad(u), ufl.grad(v)) * dx  # :100-107
    F += ufl.inner(psi, ufl.grad(v)) * dx
    F -= alpha * ufl.inner(f, v) * dx
    F -= ufl.inner(psi0, ufl.grad(v)) * dx
    F += ufl.inner(ufl.grad(u), w) * dx
    non_lin_term = 1 / (ufl.sqrt(1 + ufl.dot(psi, psi)))
    F -= phi * non_lin_term * ufl.dot(psi, w) * dx
    bcs = [fem.dirichletbc(0.0, boundary_dofs, V_trial.sub(0))]  # :109-110
    problem = NonlinearProblem(F, u=sol, bcs=bcs, petsc_options_prefix="pg_", petsc_options=dict(PETSC_OPTIONS), device=device)
    newton_iterations = np.zeros(max_iterations, dtype=np.int32)
    L2_diff = np.zeros(max_iterations)
    i = -1
    for i in range(max_iterations):  # :171-205
        alpha.value = alpha_0 if alpha_scheme == "constant" else alpha_0 + alpha_c * i if alpha_scheme == "linear" else alpha_0 * 2**i
        problem.solve()
        newton_iterations[i] = problem.solver.getIterationNumber()
        L2_diff[i] = problem.l2_increment()
        if L2_diff[i] < stopping_tol:
            break
        w0.x.array[:] = sol.x.array  # :205
    problem.close()
    return newton_iterations[: i + 1], L2_diff[: i + 1], sol
