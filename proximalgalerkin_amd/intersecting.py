"""Example 08 - intersecting constraints (an obstacle u >= phi0 AND a gradient bound |u'| <= phi on one primal field, two latent
variables) - on the HIP backend.  Host-side mirror of
/root/reference/examples/08_intersecting_constraints/intersecting_constraints_dolfinx.py, a flat script: `solve_problem` runs its
continuation in `phic` with the adaptive-alpha LVPP loop inside (:112-175), `IntersectingProblem` stands where the script builds
`dolfinx.fem.petsc.NonlinearProblem(F, z, bcs=bcs, petsc_options=sp)` (:75-77,124-126) with the `l2` line search (:66-79), and
`NonlinearProblem` takes the script's FORM (:47-58) through the UFL-subset front end - SURVEY.md section 8(f)3's acceptance test:
the residual is recognised as the composition of example 01's and example 06's latent rows, its two coordinate expressions phi0
and phi are sampled at the quadrature points, and everything below `.solve()` runs in libpgx.so (include/pgx_ic.h).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib, fem
from .problem import _SNES

# the reference's solver parameters (:66-79)
SP = {"snes_linesearch_type": "l2", "snes_linesearch_maxlambda": 1, "snes_atol": 1.0e-6, "snes_rtol": 1.0e-6, "snes_stol": 1e-14,
      "ksp_type": "preonly", "pc_type": "lu", "pc_factor_mat_solver_type": "mumps"}
PHICS = (3, 2, 1, 0.5, 0.1, 0.01)  # :114
NFAIL_MAX = 50  # :113


class NotConvergedError(Exception):  # :10-11
    pass


def phi0_bump(x, l=0.2, r=0.8):
    """:39-42 as an interpolation callable: x of shape (1, npts)"""
    x = np.asarray(x, dtype=np.float64)[0]
    inside = (x > l) & (x < r)
    xs = np.where(inside, x, 0.5)
    bump = np.exp(-1.0 / (10.0 * (xs - l) * (r - xs))) / np.exp(-1.0 / (10.0 * (0.5 - l) * (r - 0.5)))
    return np.where(inside, bump, 0.0)


def phi_bound(phic):
    """:44-45"""
    return lambda x: np.where(x[0] <= 0.2, float(phic), np.where(x[0] > 0.8, float(phic), 100.0))


class IntersectingProblem:
    """x = [u | psi0 | psi], each P1 on the interval `mesh`; phi0 / phi: callables of x (shape (1, npts))."""

    def __init__(self, mesh: fem.IntervalMesh, phi0, phi, c=0.0, petsc_options: dict | None = None, quadrature_degree=6,
                 bc_dofs=None, device=0):
        if not isinstance(mesh, fem.IntervalMesh):
            raise NotImplementedError("the kernels of example 08 are written for interval meshes (:13)")
        self._lib = lib = _lib.load()
        self.mesh = mesh
        self.nv = mesh.num_vertices
        self.ndofs = 3 * self.nv
        self.qpts, self.qwts = fem.interval_quadrature(quadrature_degree)
        x = np.ascontiguousarray(mesh.geometry[:, 0])
        self.xq = np.ascontiguousarray(x[:-1, None] + np.diff(x)[:, None] * self.qpts[None])  # (nc, nq)
        bc = np.ascontiguousarray(mesh.exterior_vertices() if bc_dofs is None else bc_dofs, dtype=np.int32)  # :60-63
        phi0_q, phi_q = self._sample(phi0), self._sample(phi)
        self._keep = (x, bc, phi0_q, phi_q)
        pp = _lib.pgx_ic_problem(self.nv, _lib.dptr(x), len(self.qwts), _lib.dptr(self.qpts), _lib.dptr(self.qwts),
                                 _lib.dptr(phi0_q), _lib.dptr(phi_q), float(c), len(bc), _lib.iptr(bc))
        self._h = C.c_void_p()
        rc = lib.pgx_ic_create(C.byref(pp), int(device), C.byref(self._h))
        if rc:
            msg = lib.pgx_ic_last_error(None)
            raise _lib.PgxError(f"pgx_ic_create failed (code {rc}): {msg.decode() if msg else ''}")
        self._phi_q = phi_q
        self._opts = _lib.pgx_snes_opts()
        lib.pgx_default_opts(C.byref(self._opts))
        for k, v in (SP if petsc_options is None else petsc_options).items():
            if k in ("snes_rtol", "snes_atol", "snes_stol"):
                setattr(self._opts, k, float(v))
            elif k == "snes_max_it":
                self._opts.snes_max_it = int(v)
            elif k == "snes_linesearch_type":
                if v not in ("l2", "bt", "none", "basic"):
                    raise NotImplementedError(f"snes_linesearch_type {v}")
                self._opts.linesearch = {"l2": 2, "bt": 1}.get(v, 0)
            elif k == "snes_linesearch_maxlambda" and float(v) != 1.0:
                raise NotImplementedError("l2 line search: maxlambda 1 (:69)")
            elif k == "snes_monitor":
                self._opts.monitor = max(self._opts.monitor, 1)
            elif k == "snes_linesearch_monitor":
                self._opts.monitor = 2
        self._opts.ksp_max_it = 6
        self.solver = _SNES(self._opts)

    def _sample(self, fn):
        v = np.asarray(fn(self.xq.reshape(1, -1)), dtype=np.float64)
        return np.ascontiguousarray(np.broadcast_to(v, (self.xq.size,)).reshape(self.xq.shape))

    def _check(self, rc, what):
        if rc:
            msg = self._lib.pgx_ic_last_error(self._h)
            raise _lib.PgxError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")

    def get_state(self):
        x = np.empty(self.ndofs)
        self._check(self._lib.pgx_ic_get_state(self._h, _lib.dptr(x)), "pgx_ic_get_state")
        return x

    def set_state(self, x):
        self._check(self._lib.pgx_ic_set_state(self._h, _lib.dptr(np.ascontiguousarray(x, dtype=np.float64))), "pgx_ic_set_state")

    def get_prev(self):
        x = np.empty(self.ndofs)
        self._check(self._lib.pgx_ic_get_prev(self._h, _lib.dptr(x)), "pgx_ic_get_prev")
        return x

    def set_prev(self, x):
        self._check(self._lib.pgx_ic_set_prev(self._h, _lib.dptr(np.ascontiguousarray(x, dtype=np.float64))), "pgx_ic_set_prev")

    def advance_prev(self):
        self._check(self._lib.pgx_ic_advance_prev(self._h), "pgx_ic_advance_prev")

    def set_alpha(self, a):
        self._check(self._lib.pgx_ic_set_alpha(self._h, float(a)), "pgx_ic_set_alpha")

    def set_phi(self, phi):
        """the gradient bound changed (`phic.value = phi_`, :116): re-sample it; uploaded only if a value moved"""
        q = self._sample(phi)
        if not np.array_equal(q, self._phi_q):
            self._check(self._lib.pgx_ic_set_phi(self._h, _lib.dptr(q)), "pgx_ic_set_phi")
            self._phi_q = q

    def solve(self):
        reason, its, lin = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self._lib.pgx_ic_newton_solve(self._h, C.byref(self._opts), C.byref(reason), C.byref(its), C.byref(lin)),
                    "pgx_ic_newton_solve")
        s = self.solver
        s._reason, s._its = reason.value, its.value
        s.ksp._its, s.ksp._reason = lin.value, (-3 if reason.value == -3 else 4)
        return reason.value, its.value

    def l2_increment(self):
        out = C.c_double(0)
        self._check(self._lib.pgx_ic_l2_increment(self._h, C.byref(out)), "pgx_ic_l2_increment")
        return out.value

    def residual(self, x=None):
        out = np.empty(self.ndofs)
        nrm = C.c_double(0)
        xx = None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        self._check(self._lib.pgx_ic_residual(self._h, _lib.dptr(xx), _lib.dptr(out), C.byref(nrm)), "pgx_ic_residual")
        return out, nrm.value

    def jacobian(self, x=None):
        import scipy.sparse as sp

        xx = None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        self._check(self._lib.pgx_ic_jacobian_fill(self._h, _lib.dptr(xx)), "pgx_ic_jacobian_fill")
        nr, nnz = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.pgx_ic_csr_export(self._h, C.byref(nr), C.byref(nnz), None, None, None), "pgx_ic_csr_export")
        rp, col, val = np.empty(nr.value + 1, np.int32), np.empty(nnz.value, np.int32), np.empty(nnz.value)
        self._check(self._lib.pgx_ic_csr_export(self._h, None, None, _lib.iptr(rp), _lib.iptr(col), _lib.dptr(val)),
                    "pgx_ic_csr_export")
        return sp.csr_matrix((val, col, rp), shape=(nr.value, nr.value))

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        self._check(self._lib.pgx_ic_spmv(self._h, _lib.dptr(x), _lib.dptr(y)), "pgx_ic_spmv")
        return y

    def profile(self, enable=True):
        ms = (C.c_double * 6)()
        self._check(self._lib.pgx_ic_profile(self._h, int(enable), ms), "pgx_ic_profile")
        return dict(zip(("residual", "jacobian", "lu_factor", "lu_solve", "spmv", "newton_total"), ms))

    def close(self):
        if self._h:
            self._lib.pgx_ic_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_problem(n: int = 1001, phis=PHICS, tol: float = 1.0e-4, nfail_max: int = NFAIL_MAX, verbose: bool = True, device: int = 0):
    """The script's body (:13-186) on the handle directly.  Returns (num_lvpp_iterations, num_newton_iterations, final state, log);
    log rows are (phic, k, alpha, Newton iterations, converged reason, increment or None for a rejected solve)."""
    mesh = fem.create_unit_interval(n)  # :13
    problem = IntersectingProblem(mesh, phi0_bump, phi_bound(100.0), device=device)
    z_prev = np.zeros(problem.ndofs)  # :24, never updated
    num_newton, num_lvpp, log = np.zeros(len(phis), dtype=np.int32), np.zeros(len(phis), dtype=np.int32), []
    for i, phi_ in enumerate(phis):
        problem.set_phi(phi_bound(phi_))  # :116
        if verbose:
            print(f"Solving for phi = {float(phi_)}", flush=True)
        alpha = 1.0  # :118
        problem.advance_prev()  # z_iter.interpolate(z) :119
        k, r, nfail = 1, 2, 0
        while nfail <= nfail_max:
            try:
                if verbose:
                    print(f"Attempting k={k} alpha={alpha}", flush=True)
                problem.set_alpha(alpha)
                problem.solve()  # :127
                num_iterations = problem.solver.getIterationNumber()
                converged_reason = problem.solver.getConvergedReason()
                num_newton[i] += num_iterations
                if num_iterations == 0 and converged_reason > 0:  # :131-135
                    raise NotConvergedError("Not converged")
                if converged_reason < 0:
                    raise NotConvergedError("Not converged")
            except NotConvergedError:
                nfail += 1
                log.append((phi_, k, alpha, num_iterations, converged_reason, None))
                if verbose:
                    print(f"Failed to converge, k={k} alpha={alpha}", flush=True)
                alpha /= 2
                problem.set_state(z_prev if k == 1 else problem.get_prev())  # :145-148
                if nfail >= nfail_max:
                    if verbose:
                        print(f"Giving up. phic={phi_} alpha={alpha} k={k}", flush=True)
                    break
                continue
            nrm = problem.l2_increment()  # :156
            log.append((phi_, k, alpha, num_iterations, converged_reason, nrm))
            if verbose:
                print(f"Solved k={k} phi={phi_} alpha={alpha} ||u_{k} - u_{k - 1}|| = {nrm}", flush=True)
            num_lvpp[i] += 1
            if nrm < tol:  # :163
                break
            if num_iterations <= 4:  # :166-169
                alpha *= r
            elif num_iterations >= 10:
                alpha /= r
            problem.advance_prev()  # :171
            k += 1
    z = problem.get_state()
    problem.close()
    return num_lvpp, num_newton, z, log


class NonlinearProblem:
    """`dolfinx.fem.petsc.NonlinearProblem(F, z, bcs=bcs, petsc_options=sp, petsc_options_prefix="snes_")` (:75-77,124-126) for
    FORMS of example 08's family: the front end (ufl.compile_form) lifts the two coordinate expressions out of the form, matches what
    remains against the composition of the exp and Hellinger latent rows, and this class samples the expressions at the quadrature
    points and drives the HIP kernels of include/pgx_ic.h.  alpha, the Constants inside phi (`phic`) and the two Functions stay
    live: every `.solve()` takes their current values.  The handle is kept in a cache on the unknown, because the script
    constructs a new NonlinearProblem for every attempt (:124-126)."""

    def __init__(self, F, u: fem.Function, bcs=None, J=None, petsc_options=None, petsc_options_prefix="", device=0):
        from . import ufl

        spec = ufl.compile_form(F, u, J)
        if not isinstance(spec, ufl.IntersectingSpec):
            raise NotImplementedError(f"this form is a {type(spec).__name__}, not the intersecting-constraints problem")
        V = u.function_space
        mesh = V.mesh
        if not isinstance(mesh, fem.IntervalMesh):
            raise NotImplementedError("the form is example 08's composition of the obstacle and gradient-bound rows, but its HIP "
                                      "kernels are written for interval meshes (intersecting_constraints_dolfinx.py:13)")
        if any(e.degree != 1 for e in V.elements) or spec.z_iter.function_space != V:
            raise NotImplementedError("example 08's kernels are written for the mixed [P1, P1, (P1)^1] space (:15-19)")
        bcs = list(bcs or [])
        ext = np.sort(mesh.exterior_vertices())
        if not (len(bcs) == 1 and bcs[0].sub == 0 and np.array_equal(np.sort(bcs[0].dofs), ext) and not np.any(bcs[0].values)):
            raise NotImplementedError("boundary conditions: u = 0 at both ends (:60-63)")
        self.spec, self.u = spec, u
        key = (id(spec.z_iter), id(spec.alpha), spec.phi0.expr, spec.phi.expr, tuple(sorted((petsc_options or {}).items(), key=str)))
        cache = u.__dict__.setdefault("_pgx_ic_cache", {})
        p = cache.get("problem") if cache.get("key") == key else None
        if p is None:
            if cache.get("problem") is not None:
                cache["problem"].close()
            p = IntersectingProblem(mesh, spec.phi0, spec.phi, c=spec.c.value if spec.c is not None else 0.0,
                                    petsc_options=petsc_options, quadrature_degree=spec.quadrature_degree or 6, device=device)
            cache["key"], cache["problem"] = key, p
        self._p = p
        self.solver = p.solver

    def solve(self):
        p, sp = self._p, self.spec
        p.set_alpha(sp.alpha.value)
        p.set_phi(sp.phi)
        p.set_prev(sp.z_iter.x.array)
        p.set_state(self.u.x.array)
        p.solve()
        # SNES solves in place: the Function holds the last iterate only after a converged solve here (lvpp/problem.py:121-123
        # keeps the old one); the script restores z itself after a failure (:145-148), so both conventions give the same run
        self.u.x.array[:] = p.get_state()
        return self.u

    def l2_increment(self):
        """sqrt(assemble_scalar(L2_u)) (:81,156) for the Functions' current values"""
        self._p.set_prev(self.spec.z_iter.x.array)
        self._p.set_state(self.u.x.array)
        return self._p.l2_increment()

    def close(self):
        cache = self.u.__dict__.get("_pgx_ic_cache", {})
        if cache.get("problem") is self._p:
            cache.clear()
        self._p.close()


def build_forms(n: int = 1001):
    """The script's problem statement (:13-63) in this package's UFL subset, verbatim.  -> dict(mesh, Z, z, z_prev, z_iter, F, bcs,
    alpha, phic, phi0, phi)"""
    from . import ufl

    mesh = fem.create_unit_interval(n)  # :13
    p = 1
    el_s = fem.element("Lagrange", mesh.cell_name(), p)
    el_v = fem.element("Lagrange", mesh.cell_name(), p, shape=(mesh.geometry.shape[1],))
    Z = fem.functionspace(mesh, fem.mixed_element([el_s, el_s, el_v]))  # :15-19
    z = fem.Function(Z, name="Solution")
    (u, psi0, psi) = ufl.split(z)
    z_test = ufl.TestFunction(Z)
    (v, w0, w) = ufl.split(z_test)
    z_prev = fem.Function(Z, name="PreviousContinuationSolution")
    z_iter = fem.Function(Z, name="PreviousLVPPSolution")
    (u_iter, psi0_iter, psi_iter) = ufl.split(z_iter)
    c = fem.Constant(mesh, 0.0)
    dx = ufl.dx(domain=mesh)
    E = 0.5 * ufl.inner(ufl.grad(u), ufl.grad(u)) * dx + c * u * dx  # :32-33
    x = ufl.SpatialCoordinate(mesh)[0]
    (l, r) = (0.2, 0.8)
    bump = ufl.exp(-1 / (10 * (x - l) * (r - x))) / ufl.exp(-1 / (10 * (0.5 - l) * (r - 0.5)))
    phi0 = ufl.conditional(ufl.le(x, l), 0, ufl.conditional(ufl.ge(x, r), 0, bump))  # :39-42
    phic = fem.Constant(mesh, 100.0)
    phi = ufl.conditional(ufl.le(x, 0.2), phic, ufl.conditional(ufl.gt(x, 0.8), phic, 100))  # :44-45
    alpha = fem.Constant(mesh, 1.0)
    F = (
        alpha * ufl.derivative(E, z, z_test)
        + ufl.inner(psi0, v) * dx
        + ufl.inner(psi, ufl.grad(v)) * dx
        - ufl.inner(psi0_iter, v) * dx
        - ufl.inner(psi_iter, ufl.grad(v)) * dx
        + ufl.inner(u, w0) * dx
        - ufl.inner(ufl.exp(psi0), w0) * dx
        - ufl.inner(phi0, w0) * dx
        + ufl.inner(ufl.grad(u), w) * dx
        - ufl.inner(phi * psi / ufl.sqrt(1 + ufl.dot(psi, psi)), w) * dx
    )  # :47-58
    bcs = [fem.dirichletbc(0.0, mesh.exterior_vertices(), Z.sub(0))]  # :60-63
    return dict(mesh=mesh, Z=Z, z=z, z_prev=z_prev, z_iter=z_iter, F=F, bcs=bcs, alpha=alpha, phic=phic, phi0=phi0, phi=phi)


def solve_problem_forms(n: int = 1001, phis=PHICS, tol: float = 1.0e-4, nfail_max: int = NFAIL_MAX, verbose: bool = False,
                        device: int = 0):
    """The reference script with its problem stated as forms (:13-175), through the front end.  Returns
    (num_lvpp_iterations, num_newton_iterations, final state)."""
    P = build_forms(n)
    z, z_prev, z_iter, F, bcs, alpha, phic = (P[k] for k in ("z", "z_prev", "z_iter", "F", "bcs", "alpha", "phic"))
    sp = dict(SP)
    num_newton, num_lvpp = np.zeros(len(phis), dtype=np.int32), np.zeros(len(phis), dtype=np.int32)
    problem = None
    for i, phi_ in enumerate(phis):
        phic.value = float(phi_)
        alpha.value = 1.0
        z_iter.x.array[:] = z.x.array
        k, rr, nfail = 1, 2, 0
        while nfail <= nfail_max:
            try:
                problem = NonlinearProblem(F, z, bcs=bcs, petsc_options=sp, petsc_options_prefix="snes_", device=device)
                problem.solve()
                num_iterations = problem.solver.getIterationNumber()
                converged_reason = problem.solver.getConvergedReason()
                num_newton[i] += num_iterations
                if num_iterations == 0 and converged_reason > 0:
                    raise NotConvergedError("Not converged")
                if converged_reason < 0:
                    raise NotConvergedError("Not converged")
            except NotConvergedError:
                nfail += 1
                alpha.value /= 2
                z.x.array[:] = z_prev.x.array if k == 1 else z_iter.x.array
                if nfail >= nfail_max:
                    break
                continue
            nrm = problem.l2_increment()
            if verbose:
                print(f"Solved k={k} phi={phi_} alpha={alpha.value} ||u_{k} - u_{k - 1}|| = {nrm}", flush=True)
            num_lvpp[i] += 1
            if nrm < tol:
                break
            if num_iterations <= 4:
                alpha.value *= rr
            elif num_iterations >= 10:
                alpha.value /= rr
            z_iter.x.array[:] = z.x.array
            k += 1
    zf = z.x.array.copy()
    if problem is not None:
        problem.close()
    return num_lvpp, num_newton, zf
