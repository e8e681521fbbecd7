"""Example 02 - Signorini contact of a linear-elastic body with a rigid plane - on the HIP backend.

Host-side mirror of /root/reference/examples/02_signorini/signorini_dolfinx.py: `solve_contact_problem` keeps the
reference's signature (:156-174) and loop (:317-358); `SignoriniProblem` stands where the script builds the blocked
`dolfinx.fem.petsc.NonlinearProblem(F, [u, psi], bcs=bcs, entity_maps=..., petsc_options=...)` (:281-291) and exposes
`.solve()`, `.solver.setTolerances(atol=, rtol=)`, `.solver.getIterationNumber()`, `.solver.getConvergedReason()`
(:331-335).  Everything below `.solve()` runs in libpgx.so (include/pgx_sg.h).  No CPU fallback.

Degrees 1 (BASELINE.json config 5) and 2 (the reference's default, :68-73) on tetrahedra (`TetMesh`) and on the reference's native
hexahedral box grid (`HexMesh`, :376-383; Q1 / Q2 elements).  The forms-driven `NonlinearProblem` below takes both degrees.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from pathlib import Path
from typing import Literal

import numpy as np

from . import _lib, fem
from .problem import ConvergenceError, _SNES

AlphaScheme = Literal["constant", "linear", "doubling"]


_TET_EDGES = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))
_TRI_EDGES = ((0, 1), (0, 2), (1, 2))


@dataclass
class TetMesh:
    geometry: np.ndarray  # (nv,3)
    cells: np.ndarray  # (nc,4)
    # ORDER-2 GEOMETRY (round 5): the geometry's mid-edge node of every edge, edges ordered by their (min, max) vertex pair - the
    # numbering of `edges()` and of p2_nodes - as a mesh of 10-node tetrahedra carries them (the reference's half sphere:
    # lvpp/mesh_generation.py:88,158).  None, or every node at its edge's midpoint: affine cells.
    midside: np.ndarray | None = None

    def __post_init__(self):
        if self.midside is not None:
            self.midside = np.ascontiguousarray(self.midside, dtype=np.float64)
            e = self.edges()
            if self.midside.shape != (len(e), 3):
                raise ValueError(f"midside must be (n_edges, 3) = {(len(e), 3)}")
            straight = 0.5 * (self.geometry[e[:, 0]] + self.geometry[e[:, 1]])
            length = np.linalg.norm(self.geometry[e[:, 0]] - self.geometry[e[:, 1]], axis=1)
            if not np.any(np.linalg.norm(self.midside - straight, axis=1) > 1e-13 * length):
                self.midside = None

    @property
    def curved(self):
        return self.midside is not None

    def flattened(self):
        """the same mesh with affine cells (what a degree-1 run uses)"""
        return TetMesh(self.geometry, self.cells) if self.curved else self

    def edges(self):
        """(ne, 2) sorted unique (min, max) vertex pairs - the edge numbering of p2_nodes and of `midside`"""
        c = self.cells.astype(np.int64)
        nv = self.geometry.shape[0]
        pairs = np.sort(np.concatenate([c[:, list(e)] for e in _TET_EDGES]), axis=1)
        ukey = np.unique(pairs[:, 0] * nv + pairs[:, 1])
        return np.stack([ukey // nv, ukey % nv], axis=1)

    def facets_where(self, pred):
        """Exterior triangles (vertex triples) whose three vertices satisfy pred(x) (locate_entities_boundary, :369-373)."""
        on = pred(self.geometry.T)
        c = self.cells
        faces = np.concatenate([c[:, [1, 2, 3]], c[:, [0, 2, 3]], c[:, [0, 1, 3]], c[:, [0, 1, 2]]])
        sel = faces[on[faces].all(axis=1)]
        key = np.sort(sel, axis=1)
        _, idx, cnt = np.unique(key, axis=0, return_index=True, return_counts=True)
        idx = idx[cnt == 1]  # exterior: the face belongs to exactly one cell
        return np.ascontiguousarray(sel[np.sort(idx)], dtype=np.int32)


class MeshTags:
    """facet_tag of the reference (:384-385): tag -> facets (vertex triples); `.find(tag)` as dolfinx.mesh.MeshTags."""

    def __init__(self, tagged: dict):
        self._t = {int(k): np.ascontiguousarray(v, dtype=np.int32) for k, v in tagged.items()}

    def find(self, tag):
        return self._t.get(int(tag), np.zeros((0, 3), dtype=np.int32))




def p2_nodes(mesh: TetMesh, *facet_sets):
    """Degree-2 node numbering of include/pgx_sg.h: the mesh vertices, then one node per edge (edges ordered by their (min, max)
    vertex pair), at the edge midpoint (a curved mesh: at the geometry's mid-edge node).  Returns (node_coords, cells10, [facets6 for every facet set])."""
    cells = mesh.cells.astype(np.int64)
    nv = mesh.geometry.shape[0]
    pairs = np.sort(np.concatenate([cells[:, list(e)] for e in _TET_EDGES]), axis=1)
    key = pairs[:, 0] * nv + pairs[:, 1]
    ukey, inv = np.unique(key, return_inverse=True)
    nc = len(cells)
    cells10 = np.ascontiguousarray(np.concatenate([cells, nv + inv.reshape(6, nc).T], axis=1), dtype=np.int32)
    e0, e1 = ukey // nv, ukey % nv
    mid = mesh.midside if getattr(mesh, "midside", None) is not None else 0.5 * (mesh.geometry[e0] + mesh.geometry[e1])
    coords = np.ascontiguousarray(np.concatenate([mesh.geometry, mid]))  # order-2 geometry: an edge node sits on the mid-edge node
    out = []
    for f in facet_sets:
        f = np.asarray(f, dtype=np.int64).reshape(-1, 3)
        fp = np.sort(np.concatenate([f[:, list(e)] for e in _TRI_EDGES]), axis=1) if len(f) else np.zeros((0, 2), dtype=np.int64)
        pos = np.searchsorted(ukey, fp[:, 0] * nv + fp[:, 1])
        if len(f) and not np.array_equal(ukey[pos], fp[:, 0] * nv + fp[:, 1]):
            raise ValueError("a facet edge is not an edge of the mesh")
        out.append(np.ascontiguousarray(np.concatenate([f, nv + pos.reshape(3, len(f)).T], axis=1), dtype=np.int32))
    return coords, cells10, out


@dataclass
class HexMesh:
    """dolfinx.mesh.create_unit_cube(comm, nx, ny, nz, CellType.hexahedron) - the reference's NATIVE mesh (signorini_dolfinx.py:376-383):
    a structured grid of boxes.  `geometry` / `cells` are the vertex lattice (nv, 3) and the 8 corner vertices per cell, numbered
    lexicographically (x fastest); the Q_d nodes of include/pgx_sg.h are the lattice refined d times (`lattice(d)`)."""
    nx: int
    ny: int
    nz: int

    def _lat(self, d):
        Nx, Ny, Nz = d * self.nx + 1, d * self.ny + 1, d * self.nz + 1
        Z, Y, X = np.meshgrid(np.linspace(0, 1, Nz), np.linspace(0, 1, Ny), np.linspace(0, 1, Nx), indexing="ij")
        return np.ascontiguousarray(np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)), (Nx, Ny, Nz)

    @property
    def geometry(self):
        return self._lat(1)[0]

    @property
    def cells(self):
        return self.lattice(1)[1]

    def lattice(self, d):
        """(node coordinates, cells [nc][(d+1)^3]) of the Q_d discretisation, local nodes lexicographic"""
        coords, (Nx, Ny, Nz) = self._lat(d)
        cz, cy, cx = np.meshgrid(np.arange(self.nz), np.arange(self.ny), np.arange(self.nx), indexing="ij")
        cx, cy, cz = cx.ravel(), cy.ravel(), cz.ravel()
        cells = np.stack([((d * cz + iz) * Ny + d * cy + iy) * Nx + d * cx + ix
                          for iz in range(d + 1) for iy in range(d + 1) for ix in range(d + 1)], axis=1)
        return coords, np.ascontiguousarray(cells, dtype=np.int32)

    def facets_where(self, pred):
        """Exterior faces (4 corner vertices of the vertex lattice, lexicographic within the face) whose corners satisfy pred(x)."""
        nx, ny, nz = self.nx, self.ny, self.nz
        Nx, Ny = nx + 1, ny + 1
        vid = lambda gx, gy, gz: (gz * Ny + gy) * Nx + gx  # noqa: E731
        out = []
        j, i = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
        for gz in (0, nz):
            out.append(np.stack([vid(i, j, gz), vid(i + 1, j, gz), vid(i, j + 1, gz), vid(i + 1, j + 1, gz)], axis=-1).reshape(-1, 4))
        k, i = np.meshgrid(np.arange(nz), np.arange(nx), indexing="ij")
        for gy in (0, ny):
            out.append(np.stack([vid(i, gy, k), vid(i + 1, gy, k), vid(i, gy, k + 1), vid(i + 1, gy, k + 1)], axis=-1).reshape(-1, 4))
        k, j = np.meshgrid(np.arange(nz), np.arange(ny), indexing="ij")
        for gx in (0, nx):
            out.append(np.stack([vid(gx, j, k), vid(gx, j + 1, k), vid(gx, j, k + 1), vid(gx, j + 1, k + 1)], axis=-1).reshape(-1, 4))
        faces = np.concatenate(out)
        on = pred(self.geometry.T)
        return np.ascontiguousarray(faces[on[faces].all(axis=1)], dtype=np.int32)

    def facet_nodes(self, faces, d):
        """Q_d nodes (lexicographic within the face) of faces given by their 4 corner vertices"""
        faces = np.asarray(faces, dtype=np.int64).reshape(-1, 4)
        Nx1, Ny1 = self.nx + 1, self.ny + 1
        Nx, Ny = d * self.nx + 1, d * self.ny + 1

        def ijk(v):
            return np.stack([v % Nx1, (v // Nx1) % Ny1, v // (Nx1 * Ny1)], axis=-1)

        c0, e1, e2 = ijk(faces[:, 0]), ijk(faces[:, 1]) - ijk(faces[:, 0]), ijk(faces[:, 2]) - ijk(faces[:, 0])
        nodes = []
        for iy in range(d + 1):
            for ix in range(d + 1):
                g = d * c0 + ix * e1 + iy * e2
                nodes.append((g[:, 2] * Ny + g[:, 1]) * Nx + g[:, 0])
        return np.ascontiguousarray(np.stack(nodes, axis=1), dtype=np.int32)


def create_unit_cube_hex(nx, ny, nz) -> HexMesh:
    return HexMesh(int(nx), int(ny), int(nz))


def create_unit_cube(nx, ny, nz) -> TetMesh:
    """nx x ny x nz cubes, six tetrahedra each around the diagonal v0-v7 (dolfinx.mesh.create_unit_cube with
    CellType.tetrahedron [split pattern recalled, not verifiable offline])."""
    xs, ys, zs = np.linspace(0, 1, nx + 1), np.linspace(0, 1, ny + 1), np.linspace(0, 1, nz + 1)
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    coords = np.ascontiguousarray(np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1))
    iz, iy, ix = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    v0 = (iz * (ny + 1) * (nx + 1) + iy * (nx + 1) + ix).ravel()
    v1, v2 = v0 + 1, v0 + (nx + 1)
    v3 = v1 + (nx + 1)
    off = (nx + 1) * (ny + 1)
    v4, v5, v6, v7 = v0 + off, v1 + off, v2 + off, v3 + off
    tets = [(v0, v1, v3, v7), (v0, v1, v7, v5), (v0, v5, v7, v4), (v0, v3, v2, v7), (v0, v6, v4, v7), (v0, v2, v6, v7)]
    cells = np.stack([np.stack(t, axis=1) for t in tets], axis=1).reshape(-1, 4)
    return TetMesh(coords, np.ascontiguousarray(cells, dtype=np.int32))


def native_tags(mesh: TetMesh) -> tuple[MeshTags, dict]:
    """The `native` branch of the reference's __main__ (:365-386): top (z = 1) tagged 1, bottom (z = 0) tagged 2."""
    top = mesh.facets_where(lambda x: np.isclose(x[2], 1.0))
    bottom = mesh.facets_where(lambda x: np.isclose(x[2], 0.0))
    return MeshTags({1: top, 2: bottom}), {"contact": (2,), "displacement": (1,)}


def curved_tables(node_coords, cells10, facets6, cell_qpts, facet_qpts):
    """The two geometry tables of include/pgx_sg.h `pgx_sg_curved` for a mesh of 10-node tetrahedra: per cell and cell quadrature
    point |det J| and J^-1 (row-major, d xi_k / d x_d) of x(xi) = sum_a X_a N2_a(xi); per contact facet and facet quadrature point the
    surface element |x_xi x x_eta| of the 6-node triangle and its z coordinate.  Node orders of p2_nodes."""
    gref3 = np.array([[-1.0, -1.0, -1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
    L = np.concatenate([1.0 - cell_qpts.sum(axis=1, keepdims=True), cell_qpts], axis=1)
    dN = np.empty((len(L), 10, 3))
    for a in range(4):
        dN[:, a] = (4 * L[:, a] - 1)[:, None] * gref3[a][None]
    for k, (a, b) in enumerate(_TET_EDGES):
        dN[:, 4 + k] = 4 * (L[:, a][:, None] * gref3[b][None] + L[:, b][:, None] * gref3[a][None])
    J = np.einsum("cad,qak->cqdk", node_coords[cells10], dN)
    det = np.linalg.det(J)
    if np.any(det.min(axis=1) * det.max(axis=1) <= 0):  # (a cell may be negatively oriented as a whole: |det J| is what enters)
        raise ValueError("order-2 geometry: the cell map is not orientation preserving at every quadrature point")
    cgeo = np.ascontiguousarray(np.concatenate([np.abs(det)[..., None], np.linalg.inv(J).reshape(len(cells10), len(L), 9)], axis=2))
    gref2 = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])
    L3 = np.stack([1.0 - facet_qpts[:, 0] - facet_qpts[:, 1], facet_qpts[:, 0], facet_qpts[:, 1]], axis=1)
    N6 = np.concatenate([L3 * (2 * L3 - 1), np.stack([4 * L3[:, a] * L3[:, b] for a, b in _TRI_EDGES], axis=1)], axis=1)
    dT = np.empty((len(L3), 6, 2))
    for a in range(3):
        dT[:, a] = (4 * L3[:, a] - 1)[:, None] * gref2[a][None]
    for k, (a, b) in enumerate(_TRI_EDGES):
        dT[:, 3 + k] = 4 * (L3[:, a][:, None] * gref2[b][None] + L3[:, b][:, None] * gref2[a][None])
    X6 = node_coords[facets6]
    t = np.einsum("fad,qak->fqdk", X6, dT)
    ds = np.linalg.norm(np.cross(t[..., 0], t[..., 1]), axis=2)
    zq = np.einsum("qa,fa->fq", N6, X6[:, :, 2])
    return cgeo, np.ascontiguousarray(np.stack([ds, zq], axis=2))


class SignoriniProblem:
    """x = [u_x | u_y | u_z | psi (contact vertices ordered by vertex id)]."""

    def __init__(self, mesh: TetMesh, contact_facets, bc_vertices, E, nu, gap, disp, quadrature_degree=4, device=0, comm=None, degree=1,
                 bc_facets=None, cell_quadrature_degree=None):
        """degree 2 (the reference's default): the Dirichlet NODES come from `bc_vertices` if given (node ids of p2_nodes), else from
        `bc_facets` (the displacement facets: vertices and edge nodes).  State layout [u_x | u_y | u_z | psi] over the P2 nodes.
        A CURVED TetMesh (order-2 geometry, `mesh.midside`) is integrated on its quadratic cells and facets (pgx_sg_create_curved):
        degree 2 isoparametrically, degree 1 with P1 fields on the curved cells; the cell integrals then use a Gauss-Jacobi rule of
        degree `cell_quadrature_degree` - the reference leaves that degree to UFL's estimator: 2 (degree - 1) for the integrand + 3 for
        det J of a quadratic tetrahedron, the default here."""
        self._lib = lib = _lib.load()
        self.mesh = mesh
        self.degree = int(degree)
        pts, wts = fem.quadrature_rule("triangle", quadrature_degree)
        self.cell_type = 1 if isinstance(mesh, HexMesh) else 0
        if self.cell_type == 1:  # the reference's native mesh: Q_d on the structured box grid
            if self.degree not in (1, 2):
                raise NotImplementedError("HIP backend: degrees 1 and 2")
            if bc_facets is None:
                raise ValueError("hexahedral meshes need the displacement FACETS")
            coords, cells = mesh.lattice(self.degree)
            facets = mesh.facet_nodes(contact_facets, self.degree)
            bv = np.unique(mesh.facet_nodes(bc_facets, self.degree).ravel()).astype(np.int64)
            g, w1 = np.polynomial.legendre.leggauss(3)  # 3 x 3 Gauss-Legendre on the unit square (degree 5 >= quadrature_degree 4)
            g, w1 = 0.5 * (g + 1.0), 0.5 * w1
            pts = np.ascontiguousarray([(g[a], g[b]) for b in range(3) for a in range(3)])
            wts = np.ascontiguousarray([w1[a] * w1[b] for b in range(3) for a in range(3)])
        elif self.degree == 2:
            if bc_facets is None and bc_vertices is None:
                raise ValueError("degree 2 needs the Dirichlet nodes or the displacement FACETS (their edge nodes are constrained as well)")
            coords, cells, (facets, bf6) = p2_nodes(mesh, contact_facets, bc_facets if bc_vertices is None else np.zeros((0, 3), np.int32))
            bv = (np.unique(bf6.ravel()) if bc_vertices is None else np.asarray(bc_vertices)).astype(np.int64)
        elif self.degree == 1:
            coords, cells = mesh.geometry, mesh.cells
            facets = np.ascontiguousarray(contact_facets, dtype=np.int32)
            bv = np.asarray(bc_vertices, dtype=np.int64)
        else:
            raise NotImplementedError("HIP backend: degrees 1 and 2")
        self.node_coords = coords
        nv = coords.shape[0]
        bc = np.ascontiguousarray(np.concatenate([bv, nv + bv, 2 * nv + bv]), dtype=np.int32)  # all components (:267)
        vals = np.ascontiguousarray(np.concatenate([np.zeros(len(bv)), np.zeros(len(bv)), np.full(len(bv), float(disp))]))
        self._keep = (coords, cells, facets, pts, wts, bc, vals)
        pm = _lib.pgx_sg_mesh(nv, cells.shape[0], _lib.dptr(coords), _lib.iptr(cells), facets.shape[0], _lib.iptr(facets), self.degree,
                              self.cell_type)
        pp = _lib.pgx_sg_problem(float(E), float(nu), float(gap), len(wts), _lib.dptr(pts), _lib.dptr(wts), len(bc),
                                 _lib.iptr(bc), _lib.dptr(vals))
        self._h = C.c_void_p()
        curved = self.cell_type == 0 and getattr(mesh, "curved", False)
        if curved and comm is not None:
            raise NotImplementedError("order-2 geometry: single handle only (include/pgx_sg.h pgx_sg_create_curved)")
        if curved:
            # isoparametric P2 on the 10-node tetrahedra: the library takes the geometry as tables over the quadrature points
            qdeg = 2 * (self.degree - 1) + 3 if cell_quadrature_degree is None else int(cell_quadrature_degree)
            qp3, qw3 = fem.quadrature_rule("tetrahedron", qdeg)
            if self.degree == 2:
                cgeo, fgeo = curved_tables(coords, cells, facets, qp3, pts)
            else:  # P1 fields: the geometry tables come from the 10-node cells / 6-node facets, the mesh arrays stay the vertices'
                gcoords, g10, (g6,) = p2_nodes(mesh, facets)
                cgeo, fgeo = curved_tables(gcoords, g10, g6, qp3, pts)
            self._keep_curved = (qp3, qw3, cgeo, fgeo)
            cv = _lib.pgx_sg_curved(len(qw3), _lib.dptr(qp3), _lib.dptr(qw3), _lib.dptr(cgeo), _lib.dptr(fgeo))
            rc = lib.pgx_sg_create_curved(C.byref(pm), C.byref(pp), C.byref(cv), int(device), C.byref(self._h))
        elif comm is None:
            rc = lib.pgx_sg_create(C.byref(pm), C.byref(pp), int(device), C.byref(self._h))
        else:  # one handle per GPU, distributed sparse LU (include/pgx_sg.h); every call below is collective
            self._comm = comm
            rc = lib.pgx_sg_create_dist(C.byref(pm), C.byref(pp), comm._c, int(device), C.byref(self._h))
        if rc:
            msg = lib.pgx_sg_last_error(None)
            raise _lib.PgxError(f"pgx_sg_create failed (code {rc}): {msg.decode() if msg else ''}")
        nt, npsi = C.c_int64(0), C.c_int64(0)
        lib.pgx_sg_num_dofs(self._h, C.byref(nt), C.byref(npsi))
        self.ndofs, self.npsi, self.nv = nt.value, npsi.value, nv
        self.contact_vertices = np.zeros(self.npsi, dtype=np.int32)
        lib.pgx_sg_contact_vertices(self._h, _lib.iptr(self.contact_vertices))
        self._opts = _lib.pgx_snes_opts()
        lib.pgx_default_opts(C.byref(self._opts))  # PETSc defaults: stol 1e-8, max_it 50 (the script sets atol/rtol only)
        self._opts.ksp_max_it = 6
        self._flags = {"snes_error_if_not_converged": True}  # :278
        self.solver = _SNES(self._opts)

    def _check(self, rc, what):
        if rc:
            msg = self._lib.pgx_sg_last_error(self._h)
            raise _lib.PgxError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")

    def get_state(self):
        x = np.empty(self.ndofs)
        self._check(self._lib.pgx_sg_get_state(self._h, _lib.dptr(x)), "pgx_sg_get_state")
        return x

    def set_state(self, x):
        self._check(self._lib.pgx_sg_set_state(self._h, _lib.dptr(np.ascontiguousarray(x, dtype=np.float64))), "set_state")

    def set_prev(self, x):
        self._check(self._lib.pgx_sg_set_prev(self._h, _lib.dptr(np.ascontiguousarray(x, dtype=np.float64))), "set_prev")

    def advance_prev(self):
        """u_prev.x.array[:] = u.x.array; psi_k.x.array[:] = psi.x.array (:342-343), on the device"""
        self._check(self._lib.pgx_sg_advance_prev(self._h), "pgx_sg_advance_prev")

    def set_alpha(self, a):
        self._check(self._lib.pgx_sg_set_alpha(self._h, float(a)), "pgx_sg_set_alpha")

    def solve(self):
        reason, its, lin = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self._lib.pgx_sg_newton_solve(self._h, C.byref(self._opts), C.byref(reason), C.byref(its),
                                                  C.byref(lin)), "pgx_sg_newton_solve")
        s = self.solver
        s._reason, s._its = reason.value, its.value
        s.ksp._its, s.ksp._reason = lin.value, (-3 if reason.value == -3 else 4)
        if reason.value <= 0 and self._flags["snes_error_if_not_converged"]:
            raise ConvergenceError(f"SNES did not converge: reason {reason.value} after {its.value} iterations")
        return reason.value, its.value

    def u_increment(self):
        out = C.c_double(0)
        self._check(self._lib.pgx_sg_u_increment(self._h, C.byref(out)), "pgx_sg_u_increment")
        return out.value

    def residual(self, x=None):
        out = np.empty(self.ndofs)
        nrm = C.c_double(0)
        xx = None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        self._check(self._lib.pgx_sg_residual(self._h, _lib.dptr(xx), _lib.dptr(out), C.byref(nrm)), "pgx_sg_residual")
        return out, nrm.value

    def jacobian(self, x=None):
        import scipy.sparse as sp

        xx = None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        self._check(self._lib.pgx_sg_jacobian_fill(self._h, _lib.dptr(xx)), "pgx_sg_jacobian_fill")
        nr, nnz = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.pgx_sg_csr_export(self._h, C.byref(nr), C.byref(nnz), None, None, None), "csr_export")
        rp, col, val = np.empty(nr.value + 1, np.int32), np.empty(nnz.value, np.int32), np.empty(nnz.value)
        self._check(self._lib.pgx_sg_csr_export(self._h, None, None, _lib.iptr(rp), _lib.iptr(col), _lib.dptr(val)),
                    "csr_export")
        return sp.csr_matrix((val, col, rp), shape=(nr.value, nr.value))

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        self._check(self._lib.pgx_sg_spmv(self._h, _lib.dptr(x), _lib.dptr(y)), "pgx_sg_spmv")
        return y

    def partition_info(self):
        """(cells whose element matrices this rank assembled, cells of the mesh): equal on a single handle; a distributed handle
        assembles its slab only (include/pgx_sg.h: pgx_sg_create_dist)."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.pgx_sg_partition_info(self._h, C.byref(a), C.byref(b)), "pgx_sg_partition_info")
        return a.value, b.value

    def lu_stats(self) -> dict:
        st = _lib.pgx_nd_stats()
        self._check(self._lib.pgx_sg_lu_stats(self._h, C.byref(st)), "pgx_sg_lu_stats")
        out = {k: getattr(st, k) for k, _ in st._fields_}
        out["symmetric"] = bool(self._lib.pgx_sg_lu_is_symmetric(self._h))  # L D L^T in LU clothing: about half of `flops` executed
        return out

    def profile(self, enable=True):
        ms = (C.c_double * 6)()
        self._check(self._lib.pgx_sg_profile(self._h, int(enable), ms), "pgx_sg_profile")
        return dict(zip(("residual", "jacobian", "lu_factor", "lu_solve", "spmv", "newton_total"), ms))

    def close(self):
        if self._h:
            self._lib.pgx_sg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_contact_problem(mesh: TetMesh, facet_tag: MeshTags, boundary_conditions: dict, degree: int = 1, E: float = 2.0e4,
                          nu: float = 0.3, gap: float = 0.0, disp: float = -0.25, newton_max_its: int = 250,
                          newton_tol: float = 1e-6, max_iterations: int = 25, alpha_scheme: AlphaScheme = "doubling",
                          alpha_0: float = 1.0, alpha_c: float = 1.0, tol: float = 1e-6, output: Path | None = None,
                          quadrature_degree: int = 4, verbose: bool = True, return_solution: bool = False, device: int = 0,
                          comm=None):
    """signorini_dolfinx.solve_contact_problem (:156-360): returns (it, iterations) [, final state, problem data]."""
    if degree not in (1, 2):
        raise NotImplementedError("HIP backend: degrees 1 (BASELINE.json config 5) and 2 (the reference's default)")
    contact = np.concatenate([facet_tag.find(t) for t in boundary_conditions["contact"]])  # :186-189
    bc_facets = np.concatenate([facet_tag.find(t) for t in boundary_conditions["displacement"]])  # :265-266
    bc_vertices = np.unique(bc_facets.ravel())
    problem = SignoriniProblem(mesh, contact, bc_vertices if degree == 1 and not isinstance(mesh, HexMesh) else None, E, nu, gap, disp,
                               quadrature_degree, device=device, comm=comm, degree=degree, bc_facets=bc_facets)
    iterations = []
    normed_diff = -1.0
    it = 0
    for it in range(1, max_iterations + 1):
        if verbose:
            print(f"{it=}/{max_iterations} {normed_diff:.2e}")
        alpha = alpha_0
        if alpha_scheme == "linear":
            alpha = alpha_0 + alpha_c * it
        elif alpha_scheme == "doubling":
            alpha = alpha_0 * 2**it
        problem.set_alpha(alpha)
        solver_tol = 10 * newton_tol if it < 2 else newton_tol  # :330
        problem.solver.setTolerances(atol=solver_tol, rtol=solver_tol)  # :331-332
        problem.solve()  # :333
        num_its = problem.solver.getIterationNumber()
        converged = problem.solver.getConvergedReason() > 0
        iterations.append(num_its)
        normed_diff = problem.u_increment()  # :336-339
        if normed_diff <= tol:
            if verbose:
                print(f"Converged at {it=} with increment norm {normed_diff:.2e}<{tol:.2e}")
            break
        problem.advance_prev()  # :342-343
        if not converged:
            break
    if output is not None:
        output = Path(output)
        output.mkdir(parents=True, exist_ok=True)
        np.savetxt(output / "lvpp_history.csv", np.asarray(iterations, dtype=np.int64), header="newton")
        from .io import write_vtu  # displacement for ParaView - the reference writes uh.bp with VTXWriter (:293-294)

        xs = problem.get_state()
        nv = problem.nv
        if isinstance(mesh, HexMesh):  # all Q_d nodes as a point cloud of first-order cells on the refined lattice
            co, ce = mesh.lattice(degree)
            sub = HexMesh(degree * mesh.nx, degree * mesh.ny, degree * mesh.nz)
            write_vtu(output / "uh.vtu", co, sub.cells[:, [0, 1, 3, 2, 4, 5, 7, 6]],
                      {"displacement": np.stack([xs[:nv], xs[nv:2 * nv], xs[2 * nv:3 * nv]], axis=1)}, cell_type="hexahedron")
        else:
            nvert = mesh.geometry.shape[0]  # degree 2: the vertex values (the edge nodes follow them in every component)
            write_vtu(output / "uh.vtu", mesh.geometry, mesh.cells,
                      {"displacement": np.stack([xs[:nvert], xs[nv:nv + nvert], xs[2 * nv:2 * nv + nvert]], axis=1)})
    if verbose:
        print(f"num_dofs_u={3 * problem.nv}, num_cells={mesh.cells.shape[0]}")
    if return_solution:
        x = problem.get_state()
        cv = problem.contact_vertices.copy()
        problem.close()
        return it, iterations, x, cv
    problem.close()
    return it, iterations


class NonlinearProblem:
    """dolfinx.fem.petsc.NonlinearProblem(F, [u, psi], bcs=bcs, petsc_options=..., entity_maps=entity_maps, kind="mpi") as
    signorini_dolfinx.py:283-291 builds it, for the blocked residual FORM of :244-252 stated in proximalgalerkin_amd.ufl (tensor
    algebra sym / tr / Identity, the `ds` measure over the contact tags, MixedFunctionSpace(V, W) with W on the contact
    sub-mesh).  The front end recognises the family, reads mu and lambda off the coefficients, the gap off g, the contact facets
    off the measure; `.solve()` / `.solver` behave like the reference's.  u, psi, psi_k are host Functions."""

    def __init__(self, F, u, bcs=None, petsc_options=None, petsc_options_prefix="", entity_maps=None, kind="mpi", device=0):
        from . import ufl

        spec = ufl.compile_signorini(F, u)
        V = spec.u.function_space
        mesh = V.mesh
        W = spec.psi.function_space
        if V.degree not in (1, 2) or V.dim != 3 or W.degree != V.degree:
            raise NotImplementedError("HIP backend: equal degrees 1 or 2 for displacement and latent variable, in 3-D (signorini_dolfinx.py:221-222)")
        if not bcs or len(bcs) != 1:
            raise NotImplementedError("one Dirichlet condition on the displacement surface (signorini_dolfinx.py:255-269)")
        bc = bcs[0]
        vals = np.asarray(bc.values)
        if vals.shape[0] != 3 or np.any(vals[:2] != 0.0) or np.any(vals[2] != vals[2, 0]):
            raise NotImplementedError("Dirichlet data (0, 0, disp) (signorini_dolfinx.py:257-262)")
        mu, lam = spec.mu, spec.lmbda
        E, nu = mu * (3.0 * lam + 2.0 * mu) / (lam + mu), lam / (2.0 * (lam + mu))
        self.spec = spec
        self._p = SignoriniProblem(mesh, spec.contact_facets, bc.dofs, E, nu, spec.gap, float(vals[2, 0]), spec.quadrature_degree,
                                   device=device, degree=V.degree)
        assert np.array_equal(self._p.contact_vertices, W.nodes())
        self.solver = self._p.solver
        self._nu3 = V.num_dofs

    def solve(self):
        p, sp = self._p, self.spec
        p.set_alpha(sp.alpha.value)
        p.set_state(np.concatenate([sp.u.x.array, sp.psi.x.array]))
        p.set_prev(np.concatenate([np.zeros(self._nu3), sp.psi_k.x.array]))  # only psi_k enters the residual (:246)
        p.solve()
        if p.solver.getConvergedReason() > 0:
            x = p.get_state()
            sp.u.x.array[:] = x[: self._nu3]
            sp.psi.x.array[:] = x[self._nu3:]

    def close(self):
        self._p.close()


def solve_contact_problem_forms(mesh: TetMesh, facet_tag: MeshTags, boundary_conditions: dict, E: float = 2.0e4, nu: float = 0.3,
                                gap: float = 0.0, disp: float = -0.25, newton_tol: float = 1e-6, max_iterations: int = 25,
                                alpha_0: float = 1.0, tol: float = 1e-6, quadrature_degree: int = 4, device: int = 0, degree: int = 1):
    """signorini_dolfinx.solve_contact_problem (:156-360) with the problem stated as the reference states it: spaces, sub-mesh,
    measures, the residual form, NonlinearProblem - through the UFL-subset front end.  alpha doubling.  Returns
    (it, iterations, u Function)."""
    from . import ufl

    def epsilon(w):  # :146-147
        return ufl.sym(ufl.grad(w))

    def sigma(w, mu, lmbda):  # :150-153
        return 2.0 * mu * epsilon(w) + lmbda * ufl.tr(ufl.grad(w)) * ufl.Identity(gdim)

    contact_facets = np.concatenate([facet_tag.find(m) for m in boundary_conditions["contact"]])  # :199-202
    gdim, fdim = 3, 2
    submesh, submesh_to_mesh = fem.create_submesh(mesh, fdim, contact_facets)  # :207
    ds = ufl.Measure("ds", domain=mesh, subdomain_data=facet_tag, subdomain_id=boundary_conditions["contact"],
                     metadata={"quadrature_degree": quadrature_degree})  # :211-218
    V = fem.functionspace(mesh, ("Lagrange", degree, (gdim,)))  # :221
    W = fem.functionspace(submesh, ("Lagrange", degree))  # :222
    Q = ufl.MixedFunctionSpace(V, W)  # :225
    v, w = ufl.TestFunctions(Q)
    u, psi, psi_k = fem.Function(V, name="displacement"), fem.Function(W), fem.Function(W)
    mu = E / (2.0 * (1.0 + nu))
    lmbda = E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu))
    n_g = fem.Constant(mesh, np.zeros(gdim))
    n_g.value[-1] = -1
    alpha = fem.Constant(mesh, alpha_0)
    f = fem.Constant(mesh, np.zeros(gdim))
    x = ufl.SpatialCoordinate(mesh)
    g = x[gdim - 1] + fem.Constant(mesh, -gap)
    residual = alpha * ufl.inner(sigma(u, mu, lmbda), epsilon(v)) * ufl.dx(domain=mesh) - alpha * ufl.inner(f, v) * ufl.dx(domain=mesh)
    residual += -ufl.inner(psi - psi_k, ufl.dot(v, n_g)) * ds
    residual += ufl.inner(ufl.dot(u, n_g), w) * ds
    residual += ufl.inner(ufl.exp(psi), w) * ds - ufl.inner(g, w) * ds
    F = ufl.extract_blocks(residual)  # :252
    u_bc = fem.Function(V)

    def disp_func(xx):  # :257-260
        values = np.zeros((gdim, xx.shape[1]))
        values[gdim - 1, :] = disp
        return values

    u_bc.interpolate(disp_func)
    bc_facets = np.concatenate([facet_tag.find(d) for d in boundary_conditions["displacement"]])  # :265-266
    bc = fem.dirichletbc(u_bc, fem.locate_dofs_topological(V, fdim, bc_facets))  # :267
    solver = NonlinearProblem(F, [u, psi], bcs=[bc], petsc_options={"snes_type": "newtonls", "snes_linesearch_type": "none"},
                              petsc_options_prefix="signorini_", entity_maps=[submesh_to_mesh], kind="mpi", device=device)
    u_prev = np.zeros_like(u.x.array)
    iterations = []
    it = 0
    for it in range(1, max_iterations + 1):  # :317-358
        alpha.value = alpha_0 * 2**it
        solver_tol = 10 * newton_tol if it < 2 else newton_tol
        solver.solver.setTolerances(atol=solver_tol, rtol=solver_tol)
        solver.solve()
        iterations.append(solver.solver.getIterationNumber())
        normed_diff = float(np.linalg.norm(u.x.array - u_prev))  # :336-339
        if normed_diff <= tol:
            break
        u_prev[:] = u.x.array
        psi_k.x.array[:] = psi.x.array
        if solver.solver.getConvergedReason() <= 0:
            break
    solver.close()
    return it, iterations, u
