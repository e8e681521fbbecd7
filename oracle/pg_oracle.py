"""CPU oracle for the LVPP / proximal-Galerkin Newton loop of the obstacle problem (example 01).

TEST INFRASTRUCTURE ONLY.  Nothing under ``proximalgalerkin_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and there
only as the checker / CPU baseline.

PARITY UNPINNED.  The reference's arithmetic for this path lives in un-vendored third-party
packages (DOLFINx v0.10.0.post1, Basix v0.10.0, UFL 2025.2.0, FFCx v0.10.0, PETSc+MUMPS; pinned at
/root/reference/docker/Dockerfile:1,60-63) that are absent from this machine, and the reference
holds no tests or golden vectors for it (/root/reference/pyproject.toml:24-26 names a tests/ dir
that does not exist).  This file therefore *restates* the algorithm from the reference's own call
sites; it is pinned by mathematics (tests/test_oracle_known_answers.py), not by reference output.

What is restated, with the reference lines each piece follows
--------------------------------------------------------------
* mesh      : right-diagonal N x M triangulation == dolfinx.mesh.create_unit_square default
              (same call as /root/reference/examples/06_gradient_constraints/
              gradient_constraint_dolfinx.py:36); ex 01 itself reads a gmsh disk
              (obstacle_pg.py:64-65).  Domain [-1,1]^2 follows obstacle_finite_difference.jl:46.
* spaces    : equal-order mixed [P1,P1], u=0 on the whole boundary, no BC on psi
              (obstacle_pg.py:68-83).
* obstacle  : phi_set, evaluated at the physical quadrature points of a degree-6 rule
              (obstacle_pg.py:92-111).
* residual  : obstacle_pg.py:116-124.   Jacobian: derivative(F, sol), obstacle_pg.py:125; block
              form as written out in obstacle_finite_difference.jl:37-43.
* BCs       : callback contract of /root/reference/src/lvpp/problem.py:54-77 (lifting with
              scale -1, set_bc with x and -1; Jacobian rows/cols zeroed, unit diagonal).
* Newton    : SNES newtonls + linesearch none + rtol 1e-6 + max_it 100, direct LU
              (obstacle_pg.py:128-139); SNESSolver.solve's "copy back only if converged"
              (problem.py:114-124).
* outer loop: alpha schedules, six observables, stopping test, sol_k <- sol
              (obstacle_pg.py:145-227).

All arithmetic is IEEE fp64.  DOF layout: x = [u_0..u_{n-1}, psi_0..psi_{n-1}] with n = number of
mesh vertices and vertex v = j*(N+1)+i.
"""
from __future__ import annotations

import json
import pathlib
import time
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

_TABLES = pathlib.Path(__file__).resolve().parents[1] / "proximalgalerkin_amd" / "tables" / "quadrature.json"


def load_quadrature(name: str = "tri_deg6_12"):
    """Read the shared table (data file, not product code). Returns (points (nq,2), weights (nq,))."""
    t = json.loads(_TABLES.read_text())[name]
    return np.asarray(t["points"], dtype=np.float64), np.asarray(t["weights"], dtype=np.float64)


# ----------------------------------------------------------------------------------------------
# mesh
# ----------------------------------------------------------------------------------------------
def create_rectangle(nx: int, ny: int, p0=(-1.0, -1.0), p1=(1.0, 1.0)):
    """Right-diagonal triangulation: vertex v=j*(nx+1)+i, square -> [v0,v1,v3],[v0,v2,v3]
    with v1=v0+1, v2=v0+nx+1, v3=v2+1."""
    xs = np.linspace(p0[0], p1[0], nx + 1)
    ys = np.linspace(p0[1], p1[1], ny + 1)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    coords = np.stack([X.ravel(), Y.ravel()], axis=1)
    i, j = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    v0 = (j * (nx + 1) + i).ravel()
    v1, v2 = v0 + 1, v0 + nx + 1
    v3 = v2 + 1
    cells = np.empty((2 * nx * ny, 3), dtype=np.int32)
    cells[0::2] = np.stack([v0, v1, v3], axis=1)
    cells[1::2] = np.stack([v0, v2, v3], axis=1)
    return coords, cells


def boundary_vertices_rectangle(nx: int, ny: int):
    i, j = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), indexing="xy")
    on = (i == 0) | (i == nx) | (j == 0) | (j == ny)
    return np.flatnonzero(on.ravel()).astype(np.int32)


def phi_set(x):
    """Obstacle of obstacle_pg.py:92-104 (x has shape (2, npts))."""
    r = np.sqrt(x[0] ** 2 + x[1] ** 2)
    r0 = 0.5
    beta = 0.9
    b = r0 * beta
    tmp = np.sqrt(r0**2 - b**2)
    B = tmp + b * b / tmp
    C = -b / tmp
    cond_true = B + r * C
    with np.errstate(invalid="ignore"):
        cond_false = np.sqrt(r0**2 - r**2)
    true_indices = np.flatnonzero(r > b)
    cond_false[true_indices] = cond_true[true_indices]
    return cond_false


# ----------------------------------------------------------------------------------------------
# P1 obstacle problem: element kernels + assembly into a fixed CSR pattern
# ----------------------------------------------------------------------------------------------
class ObstacleP1:
    """Discrete problem A.1/A.2 of SURVEY.md for P1 elements."""

    def __init__(self, coords, cells, bc_dofs, phi=phi_set, f: float = 0.0, quadrature="tri_deg6_12",
                 g_bc: float = 0.0):
        self.coords = np.ascontiguousarray(coords, dtype=np.float64)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.n = self.coords.shape[0]
        self.nc = self.cells.shape[0]
        self.bc = np.asarray(bc_dofs, dtype=np.int32)
        self.g_bc = float(g_bc)
        self.f = float(f)
        self.Xq, self.wq = load_quadrature(quadrature)
        X, Y = self.Xq[:, 0], self.Xq[:, 1]
        self.Nq = np.stack([1.0 - X - Y, X, Y], axis=1)  # (nq,3) P1 basis at quadrature points

        x = self.coords[self.cells]  # (nc,3,2)
        J = np.stack([x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]], axis=2)  # J[:, :, k] = column k
        det = J[:, 0, 0] * J[:, 1, 1] - J[:, 0, 1] * J[:, 1, 0]
        self.detJ = np.abs(det)
        invJ = np.empty_like(J)
        invJ[:, 0, 0] = J[:, 1, 1] / det
        invJ[:, 0, 1] = -J[:, 0, 1] / det
        invJ[:, 1, 0] = -J[:, 1, 0] / det
        invJ[:, 1, 1] = J[:, 0, 0] / det
        gref = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])
        self.G = np.einsum("ak,ckd->cad", gref, invJ)  # physical gradients (nc,3,2)
        self.Ke = 0.5 * self.detJ[:, None, None] * np.einsum("cad,cbd->cab", self.G, self.G)
        Mref = np.einsum("q,qa,qb->ab", self.wq, self.Nq, self.Nq)
        self.Me = self.detJ[:, None, None] * Mref[None]
        self.me = self.detJ[:, None] * np.einsum("q,qa->a", self.wq, self.Nq)[None]  # int N_a

        # obstacle at physical quadrature points (obstacle_pg.py:107-111)
        xq = np.einsum("qa,cad->cqd", self.Nq, x)  # (nc,nq,2)
        self.phi_q = phi(xq.reshape(-1, 2).T.copy()).reshape(self.nc, -1)
        bphi_e = self.detJ[:, None] * np.einsum("q,cq,qa->ca", self.wq, self.phi_q, self.Nq)
        self.b_phi = np.bincount(self.cells.ravel(), weights=bphi_e.ravel(), minlength=self.n)

        self._build_pattern()
        self.K = self._scalar_csr(self.Ke)
        self.M = self._scalar_csr(self.Me)
        self.m_l = np.bincount(self.cells.ravel(), weights=self.me.ravel(), minlength=self.n)

    # -- fixed sparsity pattern (dolfinx create_matrix analogue, problem.py:110) -----------------
    def _build_pattern(self):
        n = self.n
        r = np.repeat(self.cells, 3, axis=1).ravel().astype(np.int64)
        c = np.tile(self.cells, (1, 3)).ravel().astype(np.int64)
        key = r * n + c
        order = np.argsort(key, kind="stable")
        ks = key[order]
        starts = np.flatnonzero(np.concatenate(([True], ks[1:] != ks[:-1])))
        ukey = ks[starts]
        self._order, self._starts = order, starts
        self.indices_s = (ukey % n).astype(np.int32)
        rows = (ukey // n).astype(np.int64)
        self.indptr_s = np.zeros(n + 1, dtype=np.int64)
        np.add.at(self.indptr_s, rows + 1, 1)
        self.indptr_s = np.cumsum(self.indptr_s)
        self.nnz_s = len(ukey)
        self._rows_s = rows
        # mixed pattern: row i<n -> [cols_s(i), n+cols_s(i)], row n+i likewise
        ln = np.diff(self.indptr_s)
        off = np.arange(self.nnz_s) - self.indptr_s[rows]
        base = 2 * self.indptr_s[rows]
        self._pos_uu = base + off
        self._pos_up = base + ln[rows] + off
        self._pos_pu = 2 * self.nnz_s + base + off
        self._pos_pp = 2 * self.nnz_s + base + ln[rows] + off
        indptr = np.concatenate([2 * self.indptr_s[:-1], 2 * self.nnz_s + 2 * self.indptr_s])
        indices = np.empty(4 * self.nnz_s, dtype=np.int32)
        indices[self._pos_uu] = self.indices_s
        indices[self._pos_up] = self.indices_s + n
        indices[self._pos_pu] = self.indices_s
        indices[self._pos_pp] = self.indices_s + n
        self.indptr_J, self.indices_J = indptr, indices
        isbc = np.zeros(n, dtype=bool)
        isbc[self.bc] = True
        self.isbc = isbc
        rbc, cbc = isbc[rows], isbc[self.indices_s]
        self._keep_uu = ~(rbc | cbc)
        self._diag_bc = rbc & (rows == self.indices_s)
        self._keep_up = ~rbc
        self._keep_pu = ~cbc

    def _scalar_vals(self, Ae):
        """Sum element matrices (nc,3,3) into the scalar CSR value array."""
        return np.add.reduceat(Ae.reshape(-1)[self._order], self._starts)

    def _scalar_csr(self, Ae):
        return sp.csr_matrix((self._scalar_vals(Ae), self.indices_s, self.indptr_s), shape=(self.n, self.n))

    # -- element-level latent terms ----------------------------------------------------------------
    def exp_terms(self, psi):
        """E_q=exp(psi_h(X_q)); returns (b_exp element vectors (nc,3), D_e (nc,3,3))."""
        psi_q = psi[self.cells] @ self.Nq.T  # (nc,nq)
        with np.errstate(under="ignore"):
            Eq = np.exp(psi_q)
        wE = self.detJ[:, None] * self.wq[None] * Eq
        b = wE @ self.Nq
        D = np.einsum("cq,qa,qb->cab", wE, self.Nq, self.Nq)
        return b, D

    # -- residual: obstacle_pg.py:116-124 + problem.py:54-67 -----------------------------------------
    def residual(self, x, xk, alpha):
        n = self.n
        u, psi = x[:n], x[n:]
        psik = xk[n:]
        ut = u.copy()
        ut[self.bc] = self.g_bc  # lifting with scale -1  ==  raw residual at u with BC values imposed
        b_exp_e, _ = self.exp_terms(psi)
        b_exp = np.bincount(self.cells.ravel(), weights=b_exp_e.ravel(), minlength=n)
        Fu = alpha * (self.K @ ut) + self.M @ (psi - psik) - alpha * self.f * self.m_l
        Fp = self.M @ ut - b_exp - self.b_phi
        Fu[self.bc] = u[self.bc] - self.g_bc  # set_bc(F, bcs, x, -1)
        return np.concatenate([Fu, Fp])

    # -- Jacobian: obstacle_pg.py:125 + problem.py:69-77 ----------------------------------------------
    def jacobian_blocks(self, x):
        _, De = self.exp_terms(x[self.n:])
        return self._scalar_vals(De)

    def jacobian(self, x, alpha):
        n = self.n
        Dv = self.jacobian_blocks(x)
        data = np.zeros(4 * self.nnz_s)
        data[self._pos_uu] = np.where(self._keep_uu, alpha * self.K.data, 0.0) + self._diag_bc
        data[self._pos_up] = np.where(self._keep_up, self.M.data, 0.0)
        data[self._pos_pu] = np.where(self._keep_pu, self.M.data, 0.0)
        data[self._pos_pp] = -Dv
        return sp.csr_matrix((data, self.indices_J, self.indptr_J), shape=(2 * n, 2 * n))

    # -- observables: obstacle_pg.py:145-152,196-201 --------------------------------------------------
    def observables(self, x, xk, alpha):
        n = self.n
        u, psi, uk, psik = x[:n], x[n:], xk[:n], xk[n:]
        ce = self.cells
        wdet = self.detJ[:, None] * self.wq[None]  # (nc,nq)
        uq, pq = u[ce] @ self.Nq.T, psi[ce] @ self.Nq.T
        ukq, pkq = uk[ce] @ self.Nq.T, psik[ce] @ self.Nq.T
        gu = np.einsum("ca,cad->cd", u[ce], self.G)
        gd = np.einsum("ca,cad->cd", (u - uk)[ce], self.G)
        area = 0.5 * self.detJ
        energy = 0.5 * np.sum(area * np.sum(gu * gu, axis=1)) - self.f * np.sum(wdet * uq)
        compl = abs(np.sum(wdet * (pkq - pq) / alpha * uq))
        feas = np.sum(wdet * np.where(uq < 0, -uq, 0.0))
        dual = np.sum(wdet * np.where(pkq < pq, (pq - pkq) / alpha, 0.0))
        h1 = np.sqrt(np.sum(area * np.sum(gd * gd, axis=1)) + np.sum(wdet * (uq - ukq) ** 2))
        with np.errstate(under="ignore"):
            l2 = np.sqrt(np.sum(wdet * (np.exp(pq) - np.exp(pkq)) ** 2))
        return np.array([energy, compl, feas, dual, h1, l2])


# ----------------------------------------------------------------------------------------------
# SNES newtonls / linesearch none mirror (options obstacle_pg.py:128-139)
# ----------------------------------------------------------------------------------------------
SNES_CONVERGED_FNORM_ABS = 2
SNES_CONVERGED_FNORM_RELATIVE = 3
SNES_CONVERGED_SNORM_RELATIVE = 4
SNES_DIVERGED_LINEAR_SOLVE = -3
SNES_DIVERGED_FNORM_NAN = -4
SNES_DIVERGED_MAX_IT = -5
SNES_DIVERGED_DTOL = -9


@dataclass
class SnesOptions:
    rtol: float = 1e-8
    atol: float = 1e-50
    stol: float = 1e-8
    max_it: int = 50
    divtol: float = 1e4


@dataclass
class NewtonLog:
    fnorms: list = field(default_factory=list)
    t_residual: float = 0.0
    t_jacobian: float = 0.0
    t_factor: float = 0.0
    t_solve: float = 0.0


def newton_solve(prob: ObstacleP1, x0, xk, alpha, opts: SnesOptions, linear_solve=None, log: NewtonLog | None = None):
    """Returns (x, reason, its).  x is the *last iterate*; callers apply the
    "copy back only if reason>0" rule of problem.py:121-123 themselves."""
    x = x0.copy()
    t = time.perf_counter()
    F = prob.residual(x, xk, alpha)
    if log is not None:
        log.t_residual += time.perf_counter() - t
    fnorm = float(np.linalg.norm(F))
    fnorm0 = fnorm
    if log is not None:
        log.fnorms.append(fnorm)
    if not np.isfinite(fnorm):
        return x, SNES_DIVERGED_FNORM_NAN, 0
    if fnorm < opts.atol:
        return x, SNES_CONVERGED_FNORM_ABS, 0
    ttol = fnorm * opts.rtol
    for it in range(1, opts.max_it + 1):
        t = time.perf_counter()
        J = prob.jacobian(x, alpha)
        if log is not None:
            log.t_jacobian += time.perf_counter() - t
        if linear_solve is None:
            t = time.perf_counter()
            lu = spla.splu(J.tocsc())  # COLAMD: MMD_AT_PLUS_A explodes (100x fill) once partial pivoting leaves the diagonal
            t1 = time.perf_counter()
            dx = lu.solve(-F)
            if log is not None:
                log.t_factor += t1 - t
                log.t_solve += time.perf_counter() - t1
        else:
            dx = linear_solve(J, -F)
        if not np.all(np.isfinite(dx)):
            return x, SNES_DIVERGED_LINEAR_SOLVE, it
        x = x + dx
        t = time.perf_counter()
        F = prob.residual(x, xk, alpha)
        if log is not None:
            log.t_residual += time.perf_counter() - t
        fnorm = float(np.linalg.norm(F))
        if log is not None:
            log.fnorms.append(fnorm)
        if not np.isfinite(fnorm):
            return x, SNES_DIVERGED_FNORM_NAN, it
        if fnorm < opts.atol:
            return x, SNES_CONVERGED_FNORM_ABS, it
        if fnorm <= ttol:
            return x, SNES_CONVERGED_FNORM_RELATIVE, it
        if float(np.linalg.norm(dx)) < opts.stol * float(np.linalg.norm(x)):
            return x, SNES_CONVERGED_SNORM_RELATIVE, it
        if fnorm > opts.divtol * fnorm0:
            return x, SNES_DIVERGED_DTOL, it
    return x, SNES_DIVERGED_MAX_IT, opts.max_it


# ----------------------------------------------------------------------------------------------
# outer proximal loop: obstacle_pg.py:154-227
# ----------------------------------------------------------------------------------------------
class AlphaSchedule:
    """Step-size rules of obstacle_pg.py:173-186 (C=1, r=q=1.5 at :161-163)."""

    def __init__(self, rule: str, alpha_max: float, C=1.0, r=1.5, q=1.5):
        self.rule, self.alpha_max, self.C, self.r, self.q = rule, alpha_max, C, r, q
        self.alpha_k = 1
        self.value = 1.0

    def update(self, k: int) -> float:
        if self.rule == "constant":
            self.value = self.C
        elif self.rule == "double_exponential":
            try:
                self.value = max(self.C * self.r ** (self.q**k) - self.alpha_k, self.C)
            except OverflowError:
                pass
            self.alpha_k = self.value
            self.value = min(self.value, self.alpha_max)
        else:  # "geometric" (obstacle_pg.py:184-186 falls through for any other string)
            self.value = self.C * self.r**k
        return self.value


COLUMNS = ["Energy", "Complementarity", "Feasibility", "Dual Feasibility", "Newton steps", "Step sizes",
           "Primal increments", "Latent increments"]


def solve_problem(prob: ObstacleP1, max_outer: int, alpha_scheme: str, alpha_max: float, tol_exit: float,
                  snes: SnesOptions | None = None, linear_solve=None, log: NewtonLog | None = None,
                  verbose=False, iterates: list | None = None):
    """Mirror of obstacle_pg.solve_problem's loop (:154-227). Returns (x, history dict)."""
    snes = snes or SnesOptions(rtol=1e-6, max_it=100)  # obstacle_pg.py:137-138
    n2 = 2 * prob.n
    x = np.zeros(n2)
    xk = x.copy()
    sched = AlphaSchedule(alpha_scheme, alpha_max)
    hist = {c: [] for c in COLUMNS}
    hist["reasons"] = []
    for k in range(max_outer):
        alpha = sched.update(k)
        xn, reason, its = newton_solve(prob, x, xk, alpha, snes, linear_solve, log)
        if reason <= 0:  # snes_error_if_not_converged: obstacle_pg.py:135
            raise RuntimeError(f"SNES diverged at outer step {k}: reason {reason} after {its} its")
        x = xn
        obs = prob.observables(x, xk, alpha)
        for name, v in zip(COLUMNS[:4], obs[:4]):
            hist[name].append(float(v))
        hist["Newton steps"].append(its)
        hist["Step sizes"].append(float(alpha))
        hist["Primal increments"].append(float(obs[4]))
        hist["Latent increments"].append(float(obs[5]))
        hist["reasons"].append(reason)
        if iterates is not None:
            iterates.append(x.copy())
        if verbose:
            print(f"OUTER {k + 1} alpha {alpha:.6g} newton {its} reason {reason} incr {obs[4]:.3e}")
        if obs[4] < tol_exit:
            break
        xk = x.copy()
    return x, hist


# ----------------------------------------------------------------------------------------------
# General Lagrange degree (1 or 2) on affine triangles: obstacle_pg.py:68-70,288 (`-p {1,2}`)
# ----------------------------------------------------------------------------------------------
def build_edges(cells, nv):
    """Unique edges as sorted vertex pairs; returns (edges (ne,2), cell_edges (nc,3)) with local edge i
    OPPOSITE local vertex i (Basix/DOLFINx reference-triangle convention: SURVEY.md App. A.2)."""
    c = cells.astype(np.int64)
    pairs = np.stack([c[:, [1, 2]], c[:, [0, 2]], c[:, [0, 1]]], axis=1)  # (nc,3,2)
    pairs.sort(axis=2)
    key = pairs[:, :, 0] * nv + pairs[:, :, 1]
    uk, inv = np.unique(key.ravel(), return_inverse=True)
    edges = np.stack([uk // nv, uk % nv], axis=1).astype(np.int32)
    return edges, inv.reshape(-1, 3).astype(np.int32)


def lagrange_tabulate(degree, X, Y):
    """Basis values (nq,nd) and reference gradients (nq,nd,2) at reference points.
    P1: vertex functions. P2: 3 vertex functions l(2l-1) then 3 edge functions 4 l_j l_k (edge i opposite vertex i)."""
    l = np.stack([1.0 - X - Y, X, Y], axis=1)  # barycentric
    dl = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])
    if degree == 1:
        return l, np.broadcast_to(dl[None], (len(X), 3, 2)).copy()
    N = np.empty((len(X), 6))
    dN = np.empty((len(X), 6, 2))
    for i in range(3):
        N[:, i] = l[:, i] * (2 * l[:, i] - 1)
        dN[:, i, :] = (4 * l[:, i] - 1)[:, None] * dl[i][None]
    for i, (j, k) in enumerate(((1, 2), (0, 2), (0, 1))):
        N[:, 3 + i] = 4 * l[:, j] * l[:, k]
        dN[:, 3 + i, :] = 4 * (l[:, j][:, None] * dl[k][None] + l[:, k][:, None] * dl[j][None])
    return N, dN


class ObstacleLagrange:
    """Same discrete problem as ObstacleP1 for Lagrange degree 1 or 2 (equal order for u and psi).
    Dofs per field: vertices [0,nv) then (degree 2) edges [nv, nv+ne). x = [u | psi]."""

    def __init__(self, coords, cells, degree=1, phi=phi_set, f=0.0, quadrature="tri_deg6_12", g_bc=0.0, midside=None):
        """midside (ne, 2), optional: coordinates of the geometry's mid-side node on every edge (edge numbering of build_edges) -
        ORDER-2 GEOMETRY as the reference's own meshes have it (generate_mesh_gmsh.py:30-33 `Mesh.ElementOrder 2`,
        lvpp/mesh_generation.py:88,158): the cell map is x(xi) = sum_a X_a N2_a(xi) with the six P2 shape functions, Jacobians
        and physical gradients are evaluated per quadrature point (what DOLFINx / FFCx do for a P2 coordinate element)."""
        self.coords = np.ascontiguousarray(coords, dtype=np.float64)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.degree = int(degree)
        self.nv, self.nc = len(self.coords), len(self.cells)
        self.f, self.g_bc = float(f), float(g_bc)
        self.Xq, self.wq = load_quadrature(quadrature)
        self.Nq, self.dNq = lagrange_tabulate(self.degree, self.Xq[:, 0], self.Xq[:, 1])
        self.edges, self.cell_edges = build_edges(self.cells, self.nv)
        if self.degree == 1:
            self.cell_dofs = self.cells.copy()
            self.n = self.nv
            self.dof_coords = self.coords
        else:
            self.cell_dofs = np.concatenate([self.cells, self.nv + self.cell_edges], axis=1).astype(np.int32)
            self.n = self.nv + len(self.edges)
            self.dof_coords = np.concatenate([self.coords, 0.5 * (self.coords[self.edges[:, 0]] + self.coords[self.edges[:, 1]])])
        self.nd = self.cell_dofs.shape[1]
        # boundary dofs: vertices of exterior edges (+ the exterior edges themselves for P2)
        cnt = np.bincount(self.cell_edges.ravel(), minlength=len(self.edges))
        bedge = np.flatnonzero(cnt == 1)
        bv = np.unique(self.edges[bedge].ravel())
        self.bc = (bv if self.degree == 1 else np.concatenate([bv, self.nv + bedge])).astype(np.int32)
        self.isbc = np.zeros(self.n, dtype=bool)
        self.isbc[self.bc] = True
        x = self.coords[self.cells]
        if midside is None:
            J = np.stack([x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]], axis=2)
            det = J[:, 0, 0] * J[:, 1, 1] - J[:, 0, 1] * J[:, 1, 0]
            self.detJ = np.abs(det)
            invJ = np.empty_like(J)
            invJ[:, 0, 0], invJ[:, 0, 1] = J[:, 1, 1] / det, -J[:, 0, 1] / det
            invJ[:, 1, 0], invJ[:, 1, 1] = -J[:, 1, 0] / det, J[:, 0, 0] / det
            self.invJ = invJ
            # physical gradients at quadrature points: (nc,nq,nd,2)
            self.Gq = np.einsum("qak,ckd->cqad", self.dNq, invJ)
            wdet = self.detJ[:, None] * self.wq[None]
            Nlin = np.stack([1.0 - self.Xq[:, 0] - self.Xq[:, 1], self.Xq[:, 0], self.Xq[:, 1]], axis=1)
            xq = np.einsum("qa,cad->cqd", Nlin, x)
        else:
            # isoparametric map of degree 2: geometry nodes = the 3 vertices + the mid-side node of local edge i (opposite vertex i)
            self.midside = np.ascontiguousarray(midside, dtype=np.float64)
            X6 = np.concatenate([x, self.midside[self.cell_edges]], axis=1)  # (nc,6,2)
            N2, dN2 = lagrange_tabulate(2, self.Xq[:, 0], self.Xq[:, 1])  # (nq,6), (nq,6,2)
            J = np.einsum("cad,qak->cqdk", X6, dN2)  # J[c,q,d,k] = d x_d / d xi_k
            det = J[..., 0, 0] * J[..., 1, 1] - J[..., 0, 1] * J[..., 1, 0]
            if np.any(det.min(axis=1) * det.max(axis=1) <= 0):  # (a cell may be negatively oriented as a whole: |det J| is what enters)
                raise ValueError("order-2 geometry: the cell map is not orientation preserving at every quadrature point")
            invJ = np.empty_like(J)  # invJ[c,q,k,d] = d xi_k / d x_d
            invJ[..., 0, 0], invJ[..., 0, 1] = J[..., 1, 1] / det, -J[..., 0, 1] / det
            invJ[..., 1, 0], invJ[..., 1, 1] = -J[..., 1, 0] / det, J[..., 0, 0] / det
            self.invJq = invJ
            self.Gq = np.einsum("qak,cqkd->cqad", self.dNq, invJ)
            wdet = np.abs(det) * self.wq[None]
            self.detJ = wdet.sum(axis=1) / self.wq.sum()  # (mean |det|: 2 x cell area; diagnostics only)
            xq = np.einsum("qa,cad->cqd", N2, X6)
            if self.degree == 2:
                self.dof_coords = np.concatenate([self.coords, self.midside])
        self.wdet = wdet
        self.Ke = np.einsum("cq,cqad,cqbd->cab", wdet, self.Gq, self.Gq)
        self.Me = np.einsum("cq,qa,qb->cab", wdet, self.Nq, self.Nq)
        self.me = wdet @ self.Nq
        self.phi_q = phi(xq.reshape(-1, 2).T.copy()).reshape(self.nc, -1)
        self.b_phi = np.bincount(self.cell_dofs.ravel(), weights=((wdet * self.phi_q) @ self.Nq).ravel(), minlength=self.n)
        self._build_pattern()
        self.K = self._scalar_csr(self.Ke)
        self.M = self._scalar_csr(self.Me)
        self.m_l = np.bincount(self.cell_dofs.ravel(), weights=self.me.ravel(), minlength=self.n)

    def _build_pattern(self):
        n, nd = self.n, self.nd
        r = np.repeat(self.cell_dofs, nd, axis=1).ravel().astype(np.int64)
        c = np.tile(self.cell_dofs, (1, nd)).ravel().astype(np.int64)
        key = r * n + c
        order = np.argsort(key, kind="stable")
        ks = key[order]
        starts = np.flatnonzero(np.concatenate(([True], ks[1:] != ks[:-1])))
        ukey = ks[starts]
        self._order, self._starts = order, starts
        self.indices_s = (ukey % n).astype(np.int32)
        rows = (ukey // n).astype(np.int64)
        self.indptr_s = np.concatenate(([0], np.cumsum(np.bincount(rows, minlength=n)))).astype(np.int64)
        self.nnz_s = len(ukey)
        self._rows_s = rows
        rbc, cbc = self.isbc[rows], self.isbc[self.indices_s]
        self._keep_uu = ~(rbc | cbc)
        self._diag_bc = rbc & (rows == self.indices_s)
        self._keep_up = ~rbc
        self._keep_pu = ~cbc

    def _scalar_vals(self, Ae):
        return np.add.reduceat(Ae.reshape(-1)[self._order], self._starts)

    def _scalar_csr(self, Ae):
        return sp.csr_matrix((self._scalar_vals(Ae), self.indices_s, self.indptr_s), shape=(self.n, self.n))

    def exp_terms(self, psi):
        psi_q = psi[self.cell_dofs] @ self.Nq.T
        with np.errstate(under="ignore"):
            wE = self.wdet * np.exp(psi_q)
        return wE @ self.Nq, np.einsum("cq,qa,qb->cab", wE, self.Nq, self.Nq)

    def residual(self, x, xk, alpha):
        n = self.n
        u, psi, psik = x[:n], x[n:], xk[n:]
        ut = u.copy()
        ut[self.bc] = self.g_bc
        b_exp_e, _ = self.exp_terms(psi)
        b_exp = np.bincount(self.cell_dofs.ravel(), weights=b_exp_e.ravel(), minlength=n)
        Fu = alpha * (self.K @ ut) + self.M @ (psi - psik) - alpha * self.f * self.m_l
        Fp = self.M @ ut - b_exp - self.b_phi
        Fu[self.bc] = u[self.bc] - self.g_bc
        return np.concatenate([Fu, Fp])

    def jacobian_blocks(self, x):
        return self._scalar_vals(self.exp_terms(x[self.n:])[1])

    def jacobian(self, x, alpha):
        n = self.n
        Dv = self.jacobian_blocks(x)
        mk = lambda v: sp.csr_matrix((v, self.indices_s, self.indptr_s), shape=(n, n))  # noqa: E731
        A = mk(np.where(self._keep_uu, alpha * self.K.data, 0.0) + self._diag_bc)
        return sp.bmat([[A, mk(np.where(self._keep_up, self.M.data, 0.0))],
                        [mk(np.where(self._keep_pu, self.M.data, 0.0)), mk(-Dv)]], format="csr")

    def observables(self, x, xk, alpha):
        n = self.n
        cd = self.cell_dofs
        u, psi, uk, psik = x[:n], x[n:], xk[:n], xk[n:]
        wd = self.wdet
        uq, pq, ukq, pkq = (v[cd] @ self.Nq.T for v in (u, psi, uk, psik))
        gu = np.einsum("ca,cqad->cqd", u[cd], self.Gq)
        gd = np.einsum("ca,cqad->cqd", (u - uk)[cd], self.Gq)
        energy = 0.5 * np.sum(wd * np.sum(gu * gu, axis=2)) - self.f * np.sum(wd * uq)
        compl = abs(np.sum(wd * (pkq - pq) / alpha * uq))
        feas = np.sum(wd * np.where(uq < 0, -uq, 0.0))
        dual = np.sum(wd * np.where(pkq < pq, (pq - pkq) / alpha, 0.0))
        h1 = np.sqrt(np.sum(wd * np.sum(gd * gd, axis=2)) + np.sum(wd * (uq - ukq) ** 2))
        with np.errstate(under="ignore"):
            l2 = np.sqrt(np.sum(wd * (np.exp(pq) - np.exp(pkq)) ** 2))
        return np.array([energy, compl, feas, dual, h1, l2])
