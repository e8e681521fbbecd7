"""Lagrange elements of arbitrary degree on triangles: local node lattice, basis tabulation, global dof numbering.

What DOLFINx / Basix provide the reference with for `basix.ufl.element("Lagrange", "triangle", k)`
(examples/06_gradient_constraints/gradient_constraint_dolfinx.py:38-46, primal degree k in 2..8, latent degree k - 1).  Conventions
(this package's own; include/pgx_gc.h "general degree" and oracle/gc_oracle.py follow them):

* local nodes of degree k: the 3 vertices; then k - 1 interior nodes per edge, edge i OPPOSITE vertex i (as for P2 in pgx_mesh.cell_dofs),
  running from the edge's lower local vertex to its higher one; then the (k-1)(k-2)/2 interior nodes, by rows of increasing second
  barycentric coordinate.  Points are the equispaced lattice (i, j, l) / k.  DEVIATION from the reference for k >= 3: Basix's
  default Lagrange variant places the nodes at GLL-warped positions.  The spanned space P_k is the same, so the discrete SOLUTION
  of a problem whose data is in the space is the same function; what differs is where `interpolate` samples phi and f (the
  interpolants differ at O(h^(k+1))) and the conditioning of the nodal basis at k = 7, 8.  The oracle uses the same lattice;
  degrees 1 and 2 (BASELINE configs) have one variant only.
* basis: the nodal (Lagrange) basis of P_k on that lattice in closed form (products of one-dimensional factors in the barycentric
  coordinates).
* global dofs: vertices (mesh numbering), then the edge-interior nodes edge by edge (edges in the order of `Mesh.edges()`), stored in
  the direction lower -> higher GLOBAL vertex id; then the cell-interior nodes cell by cell.
"""
from __future__ import annotations

import numpy as np

_EDGE_ENDS = ((1, 2), (0, 2), (0, 1))  # local edge i is opposite vertex i


def lattice(k: int) -> np.ndarray:
    """Reference coordinates (n, 2) of the local nodes of degree k >= 1 on the triangle (0,0), (1,0), (0,1)."""
    if k < 1:
        raise ValueError("degree >= 1")
    verts = np.array([(0.0, 0.0), (1.0, 0.0), (0.0, 1.0)])
    pts = [verts]
    for a, b in _EDGE_ENDS:
        t = np.arange(1, k)[:, None] / k
        pts.append(verts[a][None, :] * (1 - t) + verts[b][None, :] * t)
    inner = [(i / k, j / k) for j in range(1, k) for i in range(1, k - j)]
    if inner:
        pts.append(np.array(inner))
    return np.ascontiguousarray(np.concatenate(pts))


def num_nodes(k: int) -> int:
    return (k + 1) * (k + 2) // 2


def _factor(k, m, lam):
    """P_m(lam) = prod_{a<m} (k lam - a) / (m - a) and its derivative with respect to lam; lam (npts,)"""
    val = np.ones_like(lam)
    der = np.zeros_like(lam)
    for a in range(m):
        f = (k * lam - a) / (m - a)
        der = der * f + val * (k / (m - a))
        val = val * f
    return val, der


def tabulate(k: int, pts) -> tuple[np.ndarray, np.ndarray]:
    """Values (npts, n) and reference gradients (npts, n, 2) of the Lagrange basis of degree k at reference points `pts`: the closed
    form on the equispaced lattice, N_(i,j,l) = P_i(l0) P_j(l1) P_l(l2) with barycentric coordinates l and P_m as in `_factor`
    (nodal by construction, no Vandermonde inversion: accurate to rounding for every k)."""
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 2)
    if k == 0:
        return np.ones((len(pts), 1)), np.zeros((len(pts), 1, 2))
    lam = np.stack([1.0 - pts[:, 0] - pts[:, 1], pts[:, 0], pts[:, 1]], axis=1)
    dlam = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])
    nodes = lattice(k)
    idx = np.rint(np.stack([1.0 - nodes[:, 0] - nodes[:, 1], nodes[:, 0], nodes[:, 1]], axis=1) * k).astype(int)
    P = {(c, m): _factor(k, m, lam[:, c]) for c in range(3) for m in range(k + 1)}
    V = np.empty((len(pts), len(nodes)))
    G = np.empty((len(pts), len(nodes), 2))
    for n, (i, j, l) in enumerate(idx):
        (a, da), (b, db), (c, dc) = P[(0, i)], P[(1, j)], P[(2, l)]
        V[:, n] = a * b * c
        dl = np.stack([da * b * c, a * db * c, a * b * dc], axis=1)  # derivative with respect to each barycentric coordinate
        G[:, n, :] = dl @ dlam
    return np.ascontiguousarray(V), np.ascontiguousarray(G)


def numbering(mesh, k: int):
    """-> (n_dofs, cell_dofs (nc, n) int32, dof_coordinates (n_dofs, 2)) of the degree-k Lagrange space on a triangular `fem.Mesh`."""
    cells = mesh.cells.astype(np.int64)
    nv, nc = mesh.num_vertices, len(cells)
    cols = [cells]
    n = nv
    X = mesh.geometry
    coords = [X]
    if k >= 2:
        edges, cell_edges = mesh.edges()  # edges (ne,2) sorted pairs; cell_edges (nc,3), local edge i opposite vertex i
        ne = len(edges)
        m = k - 1
        t = np.arange(1, k)[None, :, None] / k
        coords.append((X[edges[:, 0]][:, None, :] * (1 - t) + X[edges[:, 1]][:, None, :] * t).reshape(-1, 2))
        for i, (a, b) in enumerate(_EDGE_ENDS):
            e = cell_edges[:, i].astype(np.int64)
            fwd = cells[:, a] < cells[:, b]  # local direction a -> b agrees with the stored direction low -> high
            idx = np.where(fwd[:, None], np.arange(m)[None, :], np.arange(m)[None, ::-1])
            cols.append(nv + e[:, None] * m + idx)
        n += ne * m
    ni = (k - 1) * (k - 2) // 2
    if ni:
        cols.append(n + np.arange(nc)[:, None] * ni + np.arange(ni)[None, :])
        ref = lattice(k)[3 + 3 * (k - 1):]
        L = np.stack([1 - ref[:, 0] - ref[:, 1], ref[:, 0], ref[:, 1]], axis=1)  # (ni, 3)
        coords.append(np.einsum("ia,cad->cid", L, X[cells]).reshape(-1, 2))
        n += nc * ni
    return int(n), np.ascontiguousarray(np.concatenate(cols, axis=1), dtype=np.int32), np.ascontiguousarray(np.concatenate(coords))


def exterior_dofs(mesh, k: int, cell_dofs) -> np.ndarray:
    """dofs on the boundary: vertices and edge-interior nodes of the exterior edges"""
    cells = mesh.cells
    e = np.concatenate([cells[:, list(p)] for p in _EDGE_ENDS])
    key = np.sort(e, axis=1)
    _, inv, cnt = np.unique(key, axis=0, return_inverse=True, return_counts=True)
    on = (cnt[inv] == 1).reshape(3, len(cells)).T  # (nc, 3): local edge i is exterior
    out = [np.unique(np.concatenate([cells[on[:, i]][:, list(_EDGE_ENDS[i])].ravel() for i in range(3)]))]
    m = k - 1
    for i in range(3):
        if m:
            out.append(cell_dofs[on[:, i]][:, 3 + i * m:3 + (i + 1) * m].ravel())
    return np.unique(np.concatenate(out)).astype(np.int32)


# ----------------------------------------------------------------------------------------------------------------------------------
# quadrilaterals: tensor-product Lagrange elements Q_k on the structured grid of rectangles that
# `create_unit_square(..., cell_type=quadrilateral)` is (gradient_constraint_dolfinx.py:34-36,229-236).  Conventions (this package's own;
# oracle/gc_oracle.py's quadrilateral section follows them): the Q_k dofs of an nx x ny grid are the points of the k-times refined
# vertex lattice, numbered row by row (x fastest); a cell's local nodes run over its (k+1) x (k+1) sub-lattice in the same order; the
# basis is the product of the one-dimensional Lagrange bases on the equispaced nodes i/k.
# ----------------------------------------------------------------------------------------------------------------------------------
def _lagrange_1d(k: int, t):
    """values (npts, k+1) and derivatives of the 1-D Lagrange basis on the nodes i/k, from the linear factors (k t - a) / (i - a)"""
    t = np.ascontiguousarray(t, dtype=np.float64)
    if k == 0:
        return np.ones((len(t), 1)), np.zeros((len(t), 1))
    V = np.empty((len(t), k + 1))
    D = np.empty((len(t), k + 1))
    for i in range(k + 1):
        val = np.ones_like(t)
        der = np.zeros_like(t)
        for a in range(k + 1):
            if a == i:
                continue
            fac = (k * t - a) / (i - a)
            der = der * fac + val * (k / (i - a))
            val = val * fac
        V[:, i], D[:, i] = val, der
    return V, D


def num_nodes_quad(k: int) -> int:
    return (k + 1) * (k + 1)


def tabulate_quad(k: int, pts) -> tuple[np.ndarray, np.ndarray]:
    """Values (npts, (k+1)^2) and reference gradients (npts, (k+1)^2, 2) of the Q_k basis on the unit square at `pts`."""
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 2)
    Vx, Dx = _lagrange_1d(k, pts[:, 0])
    Vy, Dy = _lagrange_1d(k, pts[:, 1])
    V = (Vy[:, :, None] * Vx[:, None, :]).reshape(len(pts), -1)  # node (iy, ix) -> iy (k+1) + ix
    G = np.stack([(Vy[:, :, None] * Dx[:, None, :]).reshape(len(pts), -1), (Dy[:, :, None] * Vx[:, None, :]).reshape(len(pts), -1)],
                 axis=2)
    return np.ascontiguousarray(V), np.ascontiguousarray(G)


def numbering_quad(mesh, k: int):
    """-> (n_dofs, cell_dofs (nc, (k+1)^2) int32, dof_coordinates (n_dofs, 2)) of Q_k on a structured quadrilateral `fem.QuadMesh`;
    k = 0 is the piecewise-constant space (one dof per cell, at its centre)."""
    nx, ny = mesh.structured
    (x0, y0), (x1, y1) = mesh.box
    cx = np.tile(np.arange(nx), ny)
    cy = np.repeat(np.arange(ny), nx)
    if k == 0:
        xc = np.stack([x0 + (cx + 0.5) * (x1 - x0) / nx, y0 + (cy + 0.5) * (y1 - y0) / ny], axis=1)
        return nx * ny, np.arange(nx * ny, dtype=np.int32)[:, None].copy(), np.ascontiguousarray(xc)
    Lx, Ly = k * nx + 1, k * ny + 1
    loc = (np.repeat(np.arange(k + 1), k + 1) * Lx + np.tile(np.arange(k + 1), k + 1))[None, :]
    cd = (k * cy * Lx + k * cx)[:, None] + loc
    X = np.stack([np.tile(np.linspace(x0, x1, Lx), Ly), np.repeat(np.linspace(y0, y1, Ly), Lx)], axis=1)
    return Lx * Ly, np.ascontiguousarray(cd, dtype=np.int32), np.ascontiguousarray(X)


def exterior_dofs_quad(mesh, k: int) -> np.ndarray:
    """lattice points on the boundary of the box (locate_dofs_topological on the exterior facets, :63-69)"""
    nx, ny = mesh.structured
    Lx, Ly = k * nx + 1, k * ny + 1
    ix = np.tile(np.arange(Lx), Ly)
    iy = np.repeat(np.arange(Ly), Lx)
    return np.flatnonzero((ix == 0) | (ix == Lx - 1) | (iy == 0) | (iy == Ly - 1)).astype(np.int32)
