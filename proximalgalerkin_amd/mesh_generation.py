"""Native stand-ins for the reference's gmsh-based mesh generators (`src/lvpp/mesh_generation.py`: `create_half_disk` :11-83,
`create_half_sphere` :86-168; driven by `examples/02_signorini/generate_mesh.py:12`).  gmsh is not available offline, so the same
GEOMETRIES - same centre, radius and surface markers, same return shape `(mesh, cell_tags, facet_tags)` - are meshed by mapping a
structured simplicial grid of the (half) cube onto the (half) ball: `p -> c + r p |p|_inf / |p|_2` sends the cube's faces to the
sphere and keeps the plane z = 0 (y = 0 in 2-D) flat.  The meshes are conforming, shape-regular away from the images of the cube's
edges, and linear (order 1): the reference's default order 2 is a curved-geometry refinement this package reduces to its vertices on
input anyway (`io.py`).  `res` is the edge length on the surface near the pole, as in the reference; the grading towards the pole
(gmsh Threshold field, lc from `res` to `2 res`) is not reproduced - the grid is quasi-uniform at `res`.
"""
from __future__ import annotations

import numpy as np

from . import fem
from .signorini import MeshTags, TetMesh

__all__ = ["create_half_disk", "create_half_sphere"]


def _ball_map(p):
    """cube [-1,1]^d -> unit ball, radial: p |p|_inf / |p|_2 (0 -> 0)"""
    n2 = np.linalg.norm(p, axis=1)
    ninf = np.abs(p).max(axis=1)
    s = np.divide(ninf, n2, out=np.zeros_like(n2), where=n2 > 0)
    return p * s[:, None]


def _untangle(coords, cells, edges, curved, ratio=0.2):
    """Mid-edge nodes of a valid order-2 tetrahedral mesh: `curved` where the quadratic cell maps allow it.  Where the cube's faces meet,
    the map to the ball opens a 90 degree dihedral angle to 180 degrees: a tetrahedron with two faces on the surface keeps a positive
    volume, but its QUADRATIC map can fold once the surface edges bulge outward (33 of 648 cells at res = 0.15).  gmsh untangles such
    cells by optimisation; here the offset of every edge of an offending cell is halved until det J, sampled at the vertices, the edge
    midpoints and the degree-5 quadrature points, stays within `ratio` of its maximum over the cell (the mesh stays conforming: an
    edge has ONE node)."""
    from . import fem
    from .signorini import _TET_EDGES

    nv = len(coords)
    key = edges[:, 0] * nv + edges[:, 1]
    c = cells.astype(np.int64)
    ce = np.stack([np.searchsorted(key, np.minimum(c[:, a], c[:, b]) * nv + np.maximum(c[:, a], c[:, b])) for a, b in _TET_EDGES], axis=1)
    pts = np.concatenate([fem.quadrature_rule("tetrahedron", 5)[0], np.eye(4)[:, 1:], 0.5 * (np.eye(4)[[a for a, _ in _TET_EDGES]] + np.eye(4)[[b for _, b in _TET_EDGES]])[:, 1:]])
    L = np.concatenate([1.0 - pts.sum(axis=1, keepdims=True), pts], axis=1)
    gref = np.array([[-1.0, -1.0, -1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
    dN = np.empty((len(L), 10, 3))
    for a in range(4):
        dN[:, a] = (4 * L[:, a] - 1)[:, None] * gref[a][None]
    for k, (a, b) in enumerate(_TET_EDGES):
        dN[:, 4 + k] = 4 * (L[:, a][:, None] * gref[b][None] + L[:, b][:, None] * gref[a][None])
    straight = 0.5 * (coords[edges[:, 0]] + coords[edges[:, 1]])
    x = coords[cells]
    sgn = np.sign(np.linalg.det(np.stack([x[:, 1] - x[:, 0], x[:, 2] - x[:, 0], x[:, 3] - x[:, 0]], axis=2)))
    theta = np.ones(len(edges))
    for _ in range(12):
        mid = straight + theta[:, None] * (curved - straight)
        X10 = np.concatenate([x, mid[ce]], axis=1)
        det = np.linalg.det(np.einsum("cad,qak->cqdk", X10, dN)) * sgn[:, None]
        bad = det.min(axis=1) < ratio * det.max(axis=1)
        if not bad.any():
            break
        theta[np.unique(ce[bad])] *= 0.5
    else:
        theta[np.unique(ce[bad])] = 0.0
    return straight + theta[:, None] * (curved - straight)


def create_half_sphere(model_name=None, order: int = 2, center=(0.0, 0.0, 0.5), res: float = 0.02, r: float = 0.4, comm=None, rank: int = 0,
                       sphere_surface: int = 2, flat_surface: int = 1):
    """Half ball of radius `r` below the plane z = center[2] (the reference's geometry, `mesh_generation.py:86-168`): tetrahedra,
    curved surface tagged `sphere_surface` (the potential contact surface of example 02), flat top tagged `flat_surface` (where the
    displacement is prescribed).  Returns (TetMesh, None, MeshTags): the reference's (mesh, cell_tags, facet_tags); the cell tags
    (one physical volume) carry no information and are not modelled.  order = 2 (the reference's default, :88): the mid-edge nodes
    of the 10-node tetrahedra are the images of the grid's edge midpoints under the same map (`mesh.midside`; on the curved surface they
    lie ON the sphere) - isoparametric P2 in example 02 since round 5; order = 1: vertices only."""
    if order not in (1, 2):
        raise ValueError("order must be 1 or 2")
    n = max(2, 2 * int(np.ceil(np.pi * r / (4.0 * res))))  # a quarter circle of the surface spans n/2 cells of length ~res; even
    nz = n // 2
    xs = np.linspace(-1.0, 1.0, n + 1)
    zs = np.linspace(-1.0, 0.0, nz + 1)
    Z, Y, X = np.meshgrid(zs, xs, xs, indexing="ij")
    P = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    iz, iy, ix = np.meshgrid(np.arange(nz), np.arange(n), np.arange(n), indexing="ij")
    v0 = (iz * (n + 1) * (n + 1) + iy * (n + 1) + ix).ravel()
    v1, v2 = v0 + 1, v0 + (n + 1)
    v3 = v1 + (n + 1)
    off = (n + 1) * (n + 1)
    v4, v5, v6, v7 = v0 + off, v1 + off, v2 + off, v3 + off
    tets = [(v0, v1, v3, v7), (v0, v1, v7, v5), (v0, v5, v7, v4), (v0, v3, v2, v7), (v0, v6, v4, v7), (v0, v2, v6, v7)]
    cells = np.ascontiguousarray(np.stack([np.stack(t, axis=1) for t in tets], axis=1).reshape(-1, 4), dtype=np.int32)
    coords = np.ascontiguousarray(np.asarray(center, dtype=float)[None, :] + r * _ball_map(P))
    mesh = TetMesh(coords, cells)
    if order == 2:
        e = mesh.edges()
        curved = np.asarray(center, dtype=float)[None, :] + r * _ball_map(0.5 * (P[e[:, 0]] + P[e[:, 1]]))
        mesh = TetMesh(coords, cells, _untangle(coords, cells, e, curved))
    # exterior facets: flat <=> all three vertices come from the plane z = 0 of the cube (mapped to z = center[2] exactly)
    on_top = np.isclose(P[:, 2], 0.0)
    ext = mesh.facets_where(lambda x: np.ones(x.shape[1], dtype=bool))
    flat = on_top[ext].all(axis=1)
    return mesh, None, MeshTags({flat_surface: ext[flat], sphere_surface: ext[~flat]})


def create_half_disk(c_y: float, R: float, res: float, order: int = 1, refinement_level: int = 1, disk_marker: int = 2, top_marker: int = 1):
    """Half disk of radius `R` below the line y = c_y (`mesh_generation.py:11-83`): triangles; curved boundary tagged
    `disk_marker`, flat top `top_marker`.  Returns (fem.Mesh, None, {marker: boundary edges as vertex pairs})."""
    n = max(2, 2 * int(np.ceil(np.pi * R / (4.0 * res)))) * max(1, int(refinement_level))
    ny = n // 2
    xs = np.linspace(-1.0, 1.0, n + 1)
    ys = np.linspace(-1.0, 0.0, ny + 1)
    Y, X = np.meshgrid(ys, xs, indexing="ij")
    P = np.stack([X.ravel(), Y.ravel()], axis=1)
    jy, ix = np.meshgrid(np.arange(ny), np.arange(n), indexing="ij")
    a = (jy * (n + 1) + ix).ravel()
    b, c = a + 1, a + (n + 1)
    d = c + 1
    cells = np.ascontiguousarray(np.stack([np.stack([a, b, d], axis=1), np.stack([a, d, c], axis=1)], axis=1).reshape(-1, 3), dtype=np.int32)
    coords = np.ascontiguousarray(np.array([0.0, float(c_y)])[None, :] + R * _ball_map(P))
    mesh = fem.Mesh(coords, cells)
    e = np.concatenate([cells[:, [0, 1]], cells[:, [1, 2]], cells[:, [2, 0]]])
    key = np.sort(e, axis=1)
    _, idx, cnt = np.unique(key, axis=0, return_index=True, return_counts=True)
    ext = e[np.sort(idx[cnt == 1])]
    top = np.isclose(P[:, 1], 0.0)[ext].all(axis=1)
    return mesh, None, {int(top_marker): ext[top].astype(np.int32), int(disk_marker): ext[~top].astype(np.int32)}
