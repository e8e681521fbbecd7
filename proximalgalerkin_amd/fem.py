"""Host-side problem description: the small part of the DOLFINx surface that
/root/reference/examples/01_obstacle_problem/obstacle_pg.py uses, re-stated
declaratively because UFL/DOLFINx cannot ship.  Pure numpy, setup-time only; nothing here is on the
hot path.

Names follow the reference call sites:
    mesh.create_rectangle / create_unit_square     gradient_constraint_dolfinx.py:36
    functionspace(mesh, ("Lagrange", k), mixed)     obstacle_pg.py:68-70
    Function(V).x.array                             obstacle_pg.py:157-158,226
    Constant(mesh, value).value                     obstacle_pg.py:73-74,175-186
    exterior_boundary_dofs / dirichletbc            obstacle_pg.py:76-83
    QuadratureFunction.interpolate(phi_set)         obstacle_pg.py:106-111
"""
from __future__ import annotations

import json
import pathlib
from dataclasses import dataclass

import numpy as np

_TABLES = pathlib.Path(__file__).resolve().parent / "tables" / "quadrature.json"


def quadrature_rule(cell: str, degree: int, scheme: str | None = None):
    """(points (nq,2), weights (nq,)) on the reference triangle. Tables live in ONE file shared with
    the oracle (tools/make_quadrature_tables.py).  `scheme` names a table of that file explicitly (basix.ufl.quadrature_element's
    `scheme` argument): "tri_deg6_12_b" is the second fully symmetric 12-point degree-6 rule (tools/quadrature_uniqueness.py)."""
    if cell == "tetrahedron":  # collapsed-coordinate Gauss-Jacobi rule (basix's "GJ" scheme), exact to `degree`; weights sum to 1/6
        if scheme not in (None, "default", "GJ"):
            raise NotImplementedError(f"no quadrature table {scheme!r} for tetrahedron degree {degree}: Gauss-Jacobi is the only scheme")
        from scipy.special import roots_jacobi

        n = degree // 2 + 1
        (x0, w0), (x1, w1), (x2, w2) = roots_jacobi(n, 0.0, 0.0), roots_jacobi(n, 1.0, 0.0), roots_jacobi(n, 2.0, 0.0)
        a, b, c = 0.5 * (x2 + 1.0), 0.5 * (x1 + 1.0), 0.5 * (x0 + 1.0)
        A, B, Cc = np.meshgrid(a, b, c, indexing="ij")
        W = (w2 / 8.0)[:, None, None] * (w1 / 4.0)[None, :, None] * (w0 / 2.0)[None, None, :]
        pts = np.stack([(Cc * (1 - B) * (1 - A)).ravel(), (B * (1 - A)).ravel(), A.ravel()], axis=1)
        return np.ascontiguousarray(pts), np.ascontiguousarray(W.ravel())
    if cell == "quadrilateral":  # tensor Gauss-Legendre rule on the unit square, exact to `degree` in each variable
        if scheme not in (None, "default"):  # basix would reject an unknown scheme name: so does this
            raise NotImplementedError(f"no quadrature table {scheme!r} for quadrilateral degree {degree}: the tensor Gauss rule is the only scheme")
        g, w = np.polynomial.legendre.leggauss(degree // 2 + 1)
        g, w = 0.5 * (g + 1.0), 0.5 * w
        n = len(g)
        return (np.ascontiguousarray(np.stack([np.tile(g, n), np.repeat(g, n)], axis=1)), np.ascontiguousarray(np.repeat(w, n) * np.tile(w, n)))
    tabs = json.loads(_TABLES.read_text())
    if scheme not in (None, "default"):
        t = tabs.get(scheme)
        if t is None or t["cell"] != cell or t["degree"] != degree:
            raise NotImplementedError(f"no quadrature table {scheme!r} for {cell} degree {degree}")
        return (np.ascontiguousarray(t["points"], dtype=np.float64), np.ascontiguousarray(t["weights"], dtype=np.float64))
    for t in tabs.values():
        if t["cell"] == cell and t["degree"] == degree and not t.get("scheme_only"):
            return (np.ascontiguousarray(t["points"], dtype=np.float64),
                    np.ascontiguousarray(t["weights"], dtype=np.float64))
    raise NotImplementedError(f"no quadrature table for {cell} degree {degree} (available: "
                              f"{[(t['cell'], t['degree']) for t in tabs.values()]})")


# ------------------------------------------------------------------------------------------------
class Mesh:
    """Simplicial mesh. `structured=(nx, ny)` marks the right-diagonal triangulation with vertex
    v=j*(nx+1)+i (enables the geometric-multigrid preconditioner)."""

    def __init__(self, coords, cells, structured=None, partition=None, midside=None):
        self.geometry = np.ascontiguousarray(coords, dtype=np.float64)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        if self.geometry.ndim != 2 or self.geometry.shape[1] != 2 or self.cells.ndim != 2 or self.cells.shape[1] != 3:
            raise ValueError("expected coords (nv,2) and triangle cells (nc,3)")
        self.structured = tuple(int(s) for s in structured) if structured else None
        self.partition = partition  # StripPartition: this Mesh is one rank's strip (owned + ghost vertex rows)
        # ORDER-2 GEOMETRY (round 5): `midside[e]` = the geometry's mid-side node on edge e (numbering of edges()), as a gmsh mesh of
        # element order 2 carries it (the reference's own meshes: generate_mesh_gmsh.py:30-33, lvpp/mesh_generation.py:88,158).  The
        # cell map is then x(xi) = sum_a X_a N2_a(xi) over the 3 vertices + 3 mid-side nodes (local edge i opposite local vertex i).
        # Where every mid-side node IS its edge's midpoint the map is affine: the attribute is dropped and the affine kernels run.
        self.midside = None
        if midside is not None:
            midside = np.ascontiguousarray(midside, dtype=np.float64)
            e = self.edges()[0]
            if midside.shape != (len(e), 2):
                raise ValueError(f"midside must be (n_edges, 2) = {(len(e), 2)}")
            straight = 0.5 * (self.geometry[e[:, 0]] + self.geometry[e[:, 1]])
            length = np.linalg.norm(self.geometry[e[:, 0]] - self.geometry[e[:, 1]], axis=1)
            if np.any(np.linalg.norm(midside - straight, axis=1) > 1e-13 * length):
                self.midside = midside

    @property
    def curved(self):
        return self.midside is not None

    def flattened(self):
        """The same mesh with affine cells (mid-side nodes dropped): what a degree-1 run uses (include/pgx.h: pgx_create_curved)."""
        return Mesh(self.geometry, self.cells, structured=self.structured, partition=self.partition) if self.curved else self

    def geometry_at(self, points):
        """Cell map at reference points (nq, 2): physical points xq (nc, nq, 2) and the per-point geometry the curved kernels read,
        geo (nc, nq, 5) = |det J|, then J^-1 row-major (iJ[k][d] = d xi_k / d x_d: physical gradient G_a[d] = sum_k dN_a[k] iJ[k][d]).
        Affine meshes: the same quantities, constant over a cell."""
        X, Y = np.asarray(points)[:, 0], np.asarray(points)[:, 1]
        x = self.geometry[self.cells]
        if not self.curved:
            N = np.stack([1.0 - X - Y, X, Y], axis=1)
            xq = np.einsum("qa,cad->cqd", N, x)
            J = np.stack([x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]], axis=2)[:, None].repeat(len(X), axis=1)
        else:
            L0, L1, L2 = 1.0 - X - Y, X, Y
            N = np.stack([L0 * (2 * L0 - 1), L1 * (2 * L1 - 1), L2 * (2 * L2 - 1), 4 * L1 * L2, 4 * L0 * L2, 4 * L0 * L1], axis=1)
            z = np.zeros_like(X)
            dN = np.stack([np.stack([-(4 * L0 - 1), -(4 * L0 - 1)], axis=1), np.stack([4 * L1 - 1, z], axis=1),
                           np.stack([z, 4 * L2 - 1], axis=1), np.stack([4 * L2, 4 * L1], axis=1),
                           np.stack([-4 * L2, 4 * (L0 - L2)], axis=1), np.stack([4 * (L0 - L1), -4 * L1], axis=1)], axis=1)  # (nq,6,2)
            X6 = np.concatenate([x, self.midside[self.edges()[1]]], axis=1)
            xq = np.einsum("qa,cad->cqd", N, X6)
            J = np.einsum("cad,qak->cqdk", X6, dN)
        det = J[..., 0, 0] * J[..., 1, 1] - J[..., 0, 1] * J[..., 1, 0]
        if np.any(det.min(axis=1) * det.max(axis=1) <= 0):  # (a cell may be negatively oriented as a whole: |det J| is what enters)
            raise ValueError("the cell map is not orientation preserving at every quadrature point (tangled order-2 geometry?)")
        geo = np.empty(det.shape + (5,))
        geo[..., 0] = np.abs(det)
        geo[..., 1], geo[..., 2] = J[..., 1, 1] / det, -J[..., 0, 1] / det
        geo[..., 3], geo[..., 4] = -J[..., 1, 0] / det, J[..., 0, 0] / det
        return xq, np.ascontiguousarray(geo)

    @property
    def num_vertices(self):
        return self.geometry.shape[0]

    @property
    def num_cells(self):
        return self.cells.shape[0]

    def cell_name(self):
        return "triangle"

    def edges(self):
        """(edges (ne,2) sorted vertex pairs, cell_edges (nc,3)); local edge i is OPPOSITE local vertex i (the
        Basix reference-triangle convention). Edge numbering = lexicographic order of (min,max) vertex pairs."""
        if getattr(self, "_edges", None) is None:
            nv = self.num_vertices
            c = self.cells.astype(np.int64)
            pairs = np.stack([c[:, [1, 2]], c[:, [0, 2]], c[:, [0, 1]]], axis=1)
            pairs.sort(axis=2)
            key = pairs[:, :, 0] * nv + pairs[:, :, 1]
            uk, inv = np.unique(key.ravel(), return_inverse=True)
            self._edges = (np.stack([uk // nv, uk % nv], axis=1).astype(np.int32), inv.reshape(-1, 3).astype(np.int32))
        return self._edges

    def exterior_dofs(self, degree):
        """Dofs of a degree-k Lagrange space on exterior facets (locate_dofs_topological, obstacle_pg.py:76-79)."""
        bv = self.exterior_vertices()
        if degree == 1:
            return bv
        edges, cell_edges = self.edges()
        cnt = np.bincount(cell_edges.ravel(), minlength=len(edges))
        on_boundary = cnt == 1
        if self.partition is not None and self.structured:
            # a strip: the edges of the cut lines belong to one LOCAL cell only, but only the global boundary is exterior
            nx, _ = self.structured
            i, j = edges % (nx + 1), edges // (nx + 1) + self.partition.row0
            gy = self.partition.global_ny
            on_boundary &= ((i[:, 0] == i[:, 1]) & ((i[:, 0] == 0) | (i[:, 0] == nx))) | \
                           ((j[:, 0] == j[:, 1]) & ((j[:, 0] == 0) | (j[:, 0] == gy)))
        return np.concatenate([bv, self.num_vertices + np.flatnonzero(on_boundary)]).astype(np.int32)

    def exterior_vertices(self):
        """Vertices on exterior facets (edges that belong to exactly one cell):
        mesh.exterior_facet_indices + locate_dofs_topological of obstacle_pg.py:76-79 for P1."""
        if self.structured:
            nx, ny = self.structured
            i, j = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), indexing="xy")
            if self.partition is not None:  # a strip: only the GLOBAL boundary is exterior, not the cut lines
                j = j + self.partition.row0
                ny = self.partition.global_ny
            return np.flatnonzero(((i == 0) | (i == nx) | (j == 0) | (j == ny)).ravel()).astype(np.int32)
        c = self.cells.astype(np.int64)
        e = np.concatenate([c[:, [0, 1]], c[:, [1, 2]], c[:, [2, 0]]])
        e.sort(axis=1)
        key = e[:, 0] * self.num_vertices + e[:, 1]
        uk, cnt = np.unique(key, return_counts=True)
        b = uk[cnt == 1]
        return np.unique(np.concatenate([b // self.num_vertices, b % self.num_vertices])).astype(np.int32)


@dataclass(frozen=True)
class StripPartition:
    """Which vertex rows of the global nx x global_ny mesh this rank holds: rows [row0, row0+nrows), of which
    [own0, own0+nown) are owned and the rest are ghosts (pgx_partition_rows, include/pgx.h)."""
    comm: object
    global_ny: int
    dist_levels: int
    row0: int
    nrows: int
    own0: int
    nown: int

    @property
    def rank(self):
        return self.comm.rank

    @property
    def size(self):
        return self.comm.size


def strip_partition(comm, global_ny, dist_levels=0):
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    pt = _lib.pgx_partition(comm.rank, comm.size, int(global_ny), int(dist_levels))
    out = [C.c_int32(0) for _ in range(4)]
    rc = lib.pgx_partition_rows(C.byref(pt), *[C.byref(o) for o in out])
    _lib.check(lib, None, rc, "pgx_partition_rows")
    return StripPartition(comm, int(global_ny), int(pt.dist_levels), *[o.value for o in out])


def create_rectangle(points, n, diagonal="right", comm=None, dist_levels=0):
    """dolfinx.mesh.create_rectangle(comm, points, n) for triangles, default (right) diagonal.  With a
    `comm` (proximalgalerkin_amd.comm.Communicator) only this rank's strip of vertex rows is built - the analogue of
    DOLFINx distributing the mesh over MPI.COMM_WORLD (obstacle_pg.py:64)."""
    if diagonal != "right":
        raise NotImplementedError("only the default 'right' diagonal is implemented")
    (x0, y0), (x1, y1) = points
    nx, ny = int(n[0]), int(n[1])
    xs = np.linspace(x0, x1, nx + 1)
    ys = np.linspace(y0, y1, ny + 1)
    part = None
    if comm is not None:
        part = strip_partition(comm, ny, dist_levels)
        ys = ys[part.row0:part.row0 + part.nrows]  # bitwise the coordinates of the global mesh
        ny = part.nrows - 1
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    coords = np.stack([X.ravel(), Y.ravel()], axis=1)
    i, j = np.meshgrid(np.arange(nx, dtype=np.int64), np.arange(ny, dtype=np.int64), indexing="xy")
    v0 = (j * (nx + 1) + i).ravel()
    v1, v2 = v0 + 1, v0 + nx + 1
    v3 = v2 + 1
    cells = np.empty((2 * nx * ny, 3), dtype=np.int32)
    cells[0::2] = np.stack([v0, v1, v3], axis=1)
    cells[1::2] = np.stack([v0, v2, v3], axis=1)
    return Mesh(coords, cells, structured=(nx, ny), partition=part)


def create_disk(h: float, radius: float = 1.0):
    """Delaunay triangulation of a disk with mesh size ~h: concentric rings, staggered - the reference's example-01 domain is
    a gmsh disk of radius 1 (examples/01_obstacle_problem/generate_mesh_gmsh.py:23); gmsh itself is not available here.
    A general (unstructured) mesh: the solver preconditions with the sparse LU."""
    from scipy.spatial import Delaunay

    pts = [(0.0, 0.0)]
    nr = max(1, int(round(radius / h)))
    for k in range(1, nr + 1):
        r = radius * k / nr
        m = max(6, int(round(2 * np.pi * r / h)))
        th = 2 * np.pi * (np.arange(m) + 0.5 * (k % 2)) / m
        pts += list(zip(r * np.cos(th), r * np.sin(th)))
    pts = np.array(pts)
    tri = Delaunay(pts).simplices.astype(np.int32)
    # consistent (counter-clockwise) orientation
    a, b, c = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]]
    neg = ((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])) < 0
    tri[neg] = tri[neg][:, [0, 2, 1]]
    return Mesh(pts, tri)


class QuadMesh:
    """Structured grid of nx x ny rectangles over `box` (dolfinx.mesh.create_unit_square(..., cell_type=quadrilateral),
    gradient_constraint_dolfinx.py:34-36): vertex v = j (nx+1) + i, cell c = j nx + i with vertices in lattice order
    (lower-left, lower-right, upper-left, upper-right).  Cells are affine; `affine_corners` (origin, +x corner, +y corner) is what the
    element kernels take their geometry from."""

    def __init__(self, box, n):
        (x0, y0), (x1, y1) = box
        nx, ny = int(n[0]), int(n[1])
        if nx < 1 or ny < 1:
            raise ValueError("at least one cell per direction")
        self.box = ((float(x0), float(y0)), (float(x1), float(y1)))
        self.structured = (nx, ny)
        self.partition = None
        self.geometry = np.ascontiguousarray(np.stack([np.tile(np.linspace(x0, x1, nx + 1), ny + 1),
                                                       np.repeat(np.linspace(y0, y1, ny + 1), nx + 1)], axis=1))
        v0 = (np.repeat(np.arange(ny), nx) * (nx + 1) + np.tile(np.arange(nx), ny)).astype(np.int32)
        self.cells = np.ascontiguousarray(np.stack([v0, v0 + 1, v0 + nx + 1, v0 + nx + 2], axis=1), dtype=np.int32)

    @property
    def num_vertices(self):
        return self.geometry.shape[0]

    @property
    def num_cells(self):
        return self.cells.shape[0]

    def cell_name(self):
        return "quadrilateral"

    @property
    def affine_corners(self):
        return np.ascontiguousarray(self.cells[:, :3])

    def triangulated(self):
        """(cells (2 nc, 3)) two triangles per rectangle - for file output only"""
        c = self.cells
        return np.ascontiguousarray(np.concatenate([c[:, [0, 1, 3]], c[:, [0, 3, 2]]]))


def create_unit_square(nx, ny, cell_type="triangle"):
    if cell_type == "quadrilateral":
        return QuadMesh(((0.0, 0.0), (1.0, 1.0)), (nx, ny))
    if cell_type != "triangle":
        raise ValueError(f"cell_type {cell_type}")
    return create_rectangle(((0.0, 0.0), (1.0, 1.0)), (nx, ny))


class IntervalMesh:
    """dolfinx.mesh.create_unit_interval / create_interval (intersecting_constraints_dolfinx.py:13): vertices in increasing order,
    cell e joins vertices e and e + 1 - the topology include/pgx_ic.h takes.  `geometry` is (nv, 1) like DOLFINx's for gdim 1."""

    def __init__(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        if len(x) < 2 or not np.all(np.diff(x) > 0):
            raise ValueError("an interval mesh needs at least two strictly increasing vertex coordinates")
        self.geometry = x.reshape(-1, 1)
        self.cells = np.stack([np.arange(len(x) - 1), np.arange(1, len(x))], axis=1).astype(np.int32)
        self.structured = None
        self.partition = None

    @property
    def num_vertices(self):
        return self.geometry.shape[0]

    @property
    def num_cells(self):
        return self.cells.shape[0]

    def cell_name(self):
        return "interval"

    def exterior_vertices(self):
        return np.array([0, self.num_vertices - 1], dtype=np.int32)

    def exterior_dofs(self, degree):
        if degree != 1:
            raise NotImplementedError("P1 on the interval")
        return self.exterior_vertices()


def create_interval(n, points=(0.0, 1.0)):
    return IntervalMesh(np.linspace(float(points[0]), float(points[1]), int(n) + 1))


def create_unit_interval(n):
    return create_interval(n, (0.0, 1.0))


def interval_quadrature(degree):
    """Basix's default rule for `degree` on the interval - Gauss-Jacobi with (degree + 2) // 2 points, i.e. Gauss-Legendre - on
    (0, 1) with weights summing to 1."""
    m = (int(degree) + 2) // 2
    t, w = np.polynomial.legendre.leggauss(m)
    return np.ascontiguousarray(0.5 * (t + 1.0)), np.ascontiguousarray(0.5 * w)


# ------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class FunctionSpace:
    """Equal-order mixed Lagrange space [P_k]^ncomp on `mesh` (basix mixed_element([P,P]),
    obstacle_pg.py:68-70).  Dofs are blocked: [comp0 | comp1 | ...], vertex-numbered within a block."""
    mesh: Mesh
    degree: int = 1
    ncomp: int = 2

    @property
    def block_size(self):
        if self.degree == 1:
            return self.mesh.num_vertices
        if self.degree == 2:  # dofs per field: vertices then one per edge
            return self.mesh.num_vertices + len(self.mesh.edges()[0])
        raise NotImplementedError("Lagrange degree 1 or 2 (obstacle_pg.py:288)")

    def cell_dofs(self):
        """[nc][3] (P1) or [nc][6] (P2: 3 vertex dofs then 3 edge dofs, edge i opposite vertex i)."""
        if self.degree == 1:
            return self.mesh.cells
        return np.ascontiguousarray(
            np.concatenate([self.mesh.cells, self.mesh.num_vertices + self.mesh.edges()[1]], axis=1), dtype=np.int32)

    def dof_coordinates(self):
        x = self.mesh.geometry
        if self.degree == 1:
            return x
        if getattr(self.mesh, "curved", False):  # isoparametric: an edge dof sits on the geometry's mid-side node
            return np.concatenate([x, self.mesh.midside])
        e = self.mesh.edges()[0]
        return np.concatenate([x, 0.5 * (x[e[:, 0]] + x[e[:, 1]])])

    @property
    def num_dofs(self):
        return self.ncomp * self.block_size

    def sub(self, i):
        return SubSpace(self, i)


@dataclass(frozen=True)
class SubSpace:
    parent: object  # FunctionSpace | MixedSpace
    index: int

    def collapse(self):
        """V.sub(i).collapse() -> (scalar space of the component, dofs of the component within the parent), as
        gradient_constraint_dolfinx.py:54 uses it."""
        P = self.parent
        if isinstance(P, MixedSpace):
            sizes = P.block_sizes()
            off = int(sum(sizes[: self.index]))
            return P.scalar_space(self.index), np.arange(off, off + sizes[self.index])
        n = P.block_size
        return FunctionSpace(P.mesh, P.degree, 1), np.arange(self.index * n, (self.index + 1) * n)


@dataclass(frozen=True)
class Element:
    """basix.ufl.element(family, cell, degree, shape=...) (gradient_constraint_dolfinx.py:38-42): Lagrange, scalar (shape ()) or
    vector-valued (shape (dim,))."""
    family: str
    degree: int
    shape: tuple = ()


def element(family, cell, degree, shape=()):
    if family not in ("Lagrange", "P", "CG"):
        raise NotImplementedError(family)
    return Element("Lagrange", int(degree), tuple(shape))


def mixed_element(elements):
    """basix.ufl.mixed_element([el_0, el_1]) (gradient_constraint_dolfinx.py:44)."""
    return tuple(elements)


@dataclass(frozen=True)
class MixedSpace:
    """Mixed Lagrange space with per-component degree and shape, e.g. [P2, (P1)^2] of example 06.  Dofs are blocked in component
    order, vector components by Cartesian direction: [u | psi_x | psi_y] - the layout of include/pgx_gc.h."""
    mesh: Mesh
    elements: tuple

    @property
    def ncomp(self):
        return len(self.elements)

    def component_rank(self, i):
        return len(self.elements[i].shape)

    def scalar_space(self, i):
        return FunctionSpace(self.mesh, self.elements[i].degree, 1)

    def block_sizes(self):
        return [self.scalar_space(i).block_size * (int(np.prod(e.shape)) if e.shape else 1) for i, e in enumerate(self.elements)]

    @property
    def num_dofs(self):
        return int(sum(self.block_sizes()))

    def sub(self, i):
        return SubSpace(self, i)


@dataclass(frozen=True)
class VectorSpace:
    """functionspace(mesh, ("Lagrange", degree, (gdim,))) (signorini_dolfinx.py:221): vector-valued Lagrange space on a simplicial
    mesh (`mesh.geometry` (nv, gdim)); dofs blocked by Cartesian component, [u_x | u_y | u_z] - the layout of include/pgx_sg.h."""
    mesh: object
    degree: int
    dim: int
    ncomp = 1

    def component_rank(self, i):
        return 1

    def node_coordinates(self):
        """(n_nodes, gdim): the mesh vertices, at degree 2 followed by the edge midpoints (signorini.p2_nodes, include/pgx_sg.h)."""
        if self.degree == 1:
            return self.mesh.geometry
        if self.degree == 2 and self.dim == 3:
            from .signorini import p2_nodes

            return p2_nodes(self.mesh)[0]
        raise NotImplementedError("vector Lagrange spaces of degree 1, and degree 2 on tetrahedra")

    @property
    def num_dofs(self):
        return self.dim * self.node_coordinates().shape[0]


@dataclass(frozen=True)
class FacetSubMesh:
    """dolfinx.mesh.create_submesh(mesh, fdim, facets)[0] (signorini_dolfinx.py:207): the contact surface as a mesh of its own;
    its vertices are the parent's vertices on those facets, ordered by parent vertex id (the psi ordering of include/pgx_sg.h)."""
    parent: object
    facets: np.ndarray  # (nf, 3) parent vertex ids

    @property
    def vertices(self):
        return np.unique(self.facets)


def create_submesh(mesh, dim, facets):
    """-> (submesh, submesh_to_mesh entity map) like dolfinx.mesh.create_submesh(...)[0:2]; `facets` are vertex triples."""
    f = np.ascontiguousarray(facets, dtype=np.int32).reshape(-1, 3)
    return FacetSubMesh(mesh, f), f


@dataclass(frozen=True)
class FacetSpace:
    """functionspace(submesh, ("Lagrange", degree)) (signorini_dolfinx.py:222): scalar P1 / P2 on the contact surface; dofs ordered by
    the parent's node id (vertices, then edge nodes)."""
    mesh: FacetSubMesh
    degree: int = 1
    ncomp = 1

    def component_rank(self, i):
        return 0

    def nodes(self):
        """parent node id of every dof"""
        if self.degree == 1:
            return self.mesh.vertices
        from .signorini import p2_nodes

        return np.unique(p2_nodes(self.mesh.parent, self.mesh.facets)[2][0])

    @property
    def num_dofs(self):
        return int(len(self.nodes()))


def functionspace(mesh, element=("Lagrange", 1), ncomp=2):
    if isinstance(mesh, FacetSubMesh):
        if tuple(element)[0] != "Lagrange" or int(tuple(element)[1]) not in (1, 2):
            raise NotImplementedError("P1 or P2 on the contact surface")
        return FacetSpace(mesh, int(tuple(element)[1]))
    if isinstance(element, tuple) and len(element) == 3 and isinstance(element[2], tuple):  # ("Lagrange", k, (gdim,))
        return VectorSpace(mesh, int(element[1]), int(element[2][0]))
    if isinstance(element, tuple) and element and isinstance(element[0], Element):  # a mixed_element([...])
        return MixedSpace(mesh, tuple(element))
    if isinstance(element, Element):
        if element.shape:
            raise NotImplementedError("a vector-valued space on its own: put it in a mixed_element")
        return FunctionSpace(mesh, element.degree, 1)
    family, degree = element
    if family != "Lagrange":
        raise NotImplementedError(family)
    return FunctionSpace(mesh, int(degree), ncomp)


class _FormOperand:
    """Arithmetic on Constants / quadrature-space coefficients builds form expressions (proximalgalerkin_amd/ufl.py)."""

    def _e(self):
        from . import ufl

        return ufl.as_expr(self)

    def __mul__(self, o):
        return self._e() * o

    def __rmul__(self, o):
        return o * self._e()

    def __add__(self, o):
        return self._e() + o

    def __radd__(self, o):
        return o + self._e()

    def __sub__(self, o):
        return self._e() - o

    def __rsub__(self, o):
        return o - self._e()

    def __neg__(self):
        return -self._e()

    def __truediv__(self, o):
        return self._e() / o

    def __rtruediv__(self, o):
        return o / self._e()


class _Vector:
    """`.array` is the host numpy view. When bound to a device slot of a solver handle the two copies
    are synchronised lazily: reading `.array` pulls if the device copy is newer and (because the
    caller may write through the view) marks the device copy stale."""

    def __init__(self, n):
        self._a = np.zeros(n, dtype=np.float64)
        self._binding = None  # (NonlinearProblem, "state"|"prev")
        self._host_valid = True
        self._dev_valid = False

    @property
    def array(self):
        if not self._host_valid:
            self._binding[0]._pull(self._binding[1], self._a)
            self._host_valid = True
        self._dev_valid = False
        return self._a

    @property
    def array_r(self):
        """Read-only host view: pulls if the device copy is newer but, unlike `.array`, leaves the device copy valid
        (nothing can be written through it), so a diagnostic read between solves costs no re-upload."""
        if not self._host_valid:
            self._binding[0]._pull(self._binding[1], self._a)
            self._host_valid = True
        v = self._a.view()
        v.flags.writeable = False
        return v

    def assign_from(self, other: "_Vector"):
        """self <- other without a host round trip when both live on the same handle
        (the device-resident form of `sol_k.x.array[:] = sol.x.array[:]`, obstacle_pg.py:226)."""
        b, ob = self._binding, other._binding
        if (b and ob and b[0] is ob[0] and b[1] == "prev" and ob[1] == "state" and other._dev_valid):
            b[0]._advance_prev()
            self._dev_valid, self._host_valid = True, False
        else:
            self.array[:] = other.array


class Function(_FormOperand):
    def __init__(self, V, name="f"):
        self.function_space = V
        self.name = name
        self.x = _Vector(V.num_dofs)

    def sub_array(self, i):
        n = self.function_space.block_size
        return self.x.array[i * n:(i + 1) * n]

    def interpolate(self, fn):
        """Function.interpolate(callable) for a scalar Lagrange space: nodal values at the dof coordinates
        (gradient_constraint_dolfinx.py:55-61); fn takes x of shape (2, npts).  Vector P1 spaces (signorini_dolfinx.py:262):
        fn returns (gdim, npts)."""
        V = self.function_space
        if isinstance(V, VectorSpace):
            vals = np.asarray(fn(np.ascontiguousarray(V.node_coordinates().T)), dtype=np.float64)
            self.x.array[:] = vals.reshape(V.dim, -1).ravel()
            return
        if not isinstance(V, FunctionSpace) or V.ncomp != 1:
            raise NotImplementedError("interpolate: scalar Lagrange spaces")
        self.x.array[:] = np.asarray(fn(np.ascontiguousarray(V.dof_coordinates().T)), dtype=np.float64)


class Constant(_FormOperand):
    """fem.Constant(mesh, value): a scalar, or a vector (signorini_dolfinx.py:236-240: n_g, f) whose `.value` array may be edited."""

    def __init__(self, mesh, value):
        v = np.asarray(value, dtype=np.float64)
        self.value = float(v) if v.ndim == 0 else v.copy()

    @property
    def rank(self):
        return 0 if isinstance(self.value, float) else 1

    def __float__(self):
        return float(self.value)


class QuadratureFunction(_FormOperand):
    """Function in a quadrature space of the given degree: one value per (cell, point), dof =
    cell*nq + q (basix.ufl.quadrature_element + fem.functionspace, obstacle_pg.py:107-110)."""

    def __init__(self, mesh: Mesh, degree: int, name="phi", scheme: str | None = None):
        self.mesh, self.degree, self.name = mesh, degree, name
        self.points, self.weights = quadrature_rule(mesh.cell_name(), degree, scheme)
        self.values = np.zeros((mesh.num_cells, len(self.weights)))

    def physical_points(self):
        if getattr(self.mesh, "curved", False):
            return self.mesh.geometry_at(self.points)[0]
        X, Y = self.points[:, 0], self.points[:, 1]
        N = np.stack([1.0 - X - Y, X, Y], axis=1)
        x = self.mesh.geometry[self.mesh.cells]  # (nc,3,2)
        return np.einsum("qa,cad->cqd", N, x)

    def interpolate(self, fn, chunk=1 << 20):
        """fn takes x of shape (2, npts) like a dolfinx interpolation callable."""
        nc, nq = self.values.shape
        if getattr(self.mesh, "curved", False):  # order-2 geometry: the quadrature points sit on the curved cells
            xq = self.physical_points().reshape(-1, 2).T
            self.values[:] = np.asarray(fn(np.ascontiguousarray(xq))).reshape(-1, nq)
            return
        X, Y = self.points[:, 0], self.points[:, 1]
        N = np.stack([1.0 - X - Y, X, Y], axis=1)
        for s in range(0, nc, chunk):
            x = self.mesh.geometry[self.mesh.cells[s:s + chunk]]
            xq = np.einsum("qa,cad->cqd", N, x).reshape(-1, 2).T
            self.values[s:s + chunk] = np.asarray(fn(np.ascontiguousarray(xq))).reshape(-1, nq)


@dataclass
class DirichletBC:
    dofs: np.ndarray     # dofs within the sub-space block
    values: np.ndarray
    sub: int             # which component of the mixed space


def locate_dofs_topological(V, dim, facets):
    """dofs of a vector Lagrange space on the given facets (vertex triples): one block index per node - each carries V.dim
    components (signorini_dolfinx.py:267); at degree 2 the facets' edge nodes are included."""
    if isinstance(V, VectorSpace) and V.degree == 2:
        from .signorini import p2_nodes

        return np.unique(p2_nodes(V.mesh, facets)[2][0]).astype(np.int32)
    return np.unique(np.asarray(facets).ravel()).astype(np.int32)


def dirichletbc(value, dofs, V=None):
    """fem.dirichletbc(value=u_bc, dofs=dofs, V=V.sub(0)) (obstacle_pg.py:83); fem.dirichletbc(u_bc, dofs) with u_bc a Function
    of a vector space (signorini_dolfinx.py:267): `values` is then (gdim, len(dofs))."""
    sub = V.index if isinstance(V, SubSpace) else 0
    dofs = np.ascontiguousarray(dofs, dtype=np.int32)
    if isinstance(value, Function) and isinstance(value.function_space, VectorSpace):
        Vv = value.function_space
        return DirichletBC(dofs, np.ascontiguousarray(value.x.array.reshape(Vv.dim, -1)[:, dofs]), 0)
    if isinstance(value, Function):
        vals = np.ascontiguousarray(value.x.array[:value.function_space.block_size][dofs])
    else:
        vals = np.full(len(dofs), float(value))
    return DirichletBC(dofs, vals, sub)
