"""Mesh in / field out - host-side conveniences around the solvers, no GPU involved.

The reference reads its meshes from XDMF/HDF5 written by gmsh scripts (examples/01_obstacle_problem/generate_mesh_gmsh.py:12-43,
obstacle_pg.py:64-65) and writes results with VTXWriter / XDMFFile (obstacle_pg.py:239-243,
gradient_constraint_dolfinx.py:145-158, signorini_dolfinx.py:293-299).  HDF5 and ADIOS2 are not available offline; the
formats offered here are the ones every gmsh / ParaView installation handles directly:

    read_msh(path)                      gmsh MSH 2.2 / 4.1 ASCII  -> (points, cells_by_type, cell_tags_by_type)
    write_vtu(path, points, cells, ...) VTK unstructured grid (XML, ASCII) with point / cell data; linear and quadratic
                                        triangles, linear tetrahedra
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

# gmsh element type -> (name, nodes)
_GMSH = {1: ("line", 2), 2: ("triangle", 3), 4: ("tetra", 4), 8: ("line3", 3), 9: ("triangle6", 6), 15: ("vertex", 1)}
_VTK = {"triangle": 5, "triangle6": 22, "tetra": 10, "line": 3, "vertex": 1}


def read_msh(path):
    """Parse a gmsh ASCII mesh (format 2.2 or 4.1).  Returns (points (n,3), {type: (m,k) int32 connectivity, 0-based},
    {type: (m,) int32 physical tags})."""
    tok = Path(path).read_text().split("\n")
    i = 0
    sections = {}
    while i < len(tok):
        line = tok[i].strip()
        if line.startswith("$") and not line.startswith("$End"):
            name = line[1:]
            j = i + 1
            while tok[j].strip() != "$End" + name:
                j += 1
            sections[name] = tok[i + 1: j]
            i = j
        i += 1
    version = float(sections["MeshFormat"][0].split()[0])
    if int(sections["MeshFormat"][0].split()[1]) != 0:
        raise NotImplementedError("binary MSH files are not supported: export ASCII (gmsh -format msh41 / msh22)")
    cells, tags = {}, {}

    def add(etype, conn, tag):
        name, k = _GMSH[etype]
        cells.setdefault(name, []).append(conn[:k])
        tags.setdefault(name, []).append(tag)

    if version < 3.0:
        nl = sections["Nodes"]
        n = int(nl[0])
        ids = np.empty(n, dtype=np.int64)
        pts = np.empty((n, 3))
        for r in range(n):
            f = nl[1 + r].split()
            ids[r] = int(f[0])
            pts[r] = [float(v) for v in f[1:4]]
        el = sections["Elements"]
        for r in range(int(el[0])):
            f = [int(v) for v in el[1 + r].split()]
            etype, ntags = f[1], f[2]
            if etype in _GMSH:
                add(etype, f[3 + ntags:], f[3] if ntags else 0)
    else:
        nl = sections["Nodes"]
        nblocks, n = int(nl[0].split()[0]), int(nl[0].split()[1])
        ids = np.empty(n, dtype=np.int64)
        pts = np.empty((n, 3))
        r, q = 1, 0
        for _ in range(nblocks):
            nb = int(nl[r].split()[3])
            for k in range(nb):
                ids[q + k] = int(nl[r + 1 + k])
                pts[q + k] = [float(v) for v in nl[r + 1 + nb + k].split()[:3]]
            r += 1 + 2 * nb
            q += nb
        # entity -> physical tag
        phys = {}
        if "Entities" in sections:
            en = sections["Entities"]
            counts = [int(v) for v in en[0].split()]
            r = 1
            for dim in range(4):
                for _ in range(counts[dim]):
                    f = en[r].split()
                    etag = int(f[0])
                    off = 4 if dim == 0 else 7
                    nphys = int(f[off])
                    phys[(dim, etag)] = int(f[off + 1]) if nphys else 0
                    r += 1
        el = sections["Elements"]
        nblocks = int(el[0].split()[0])
        r = 1
        for _ in range(nblocks):
            dim, etag, etype, nb = (int(v) for v in el[r].split())
            for k in range(nb):
                f = [int(v) for v in el[r + 1 + k].split()]
                if etype in _GMSH:
                    add(etype, f[1:], phys.get((dim, etag), 0))
            r += 1 + nb
    lookup = np.full(int(ids.max()) + 1, -1, dtype=np.int64)
    lookup[ids] = np.arange(len(ids))
    out_c = {k: np.ascontiguousarray(lookup[np.asarray(v, dtype=np.int64)], dtype=np.int32) for k, v in cells.items()}
    out_t = {k: np.asarray(v, dtype=np.int32) for k, v in tags.items()}
    return pts, out_c, out_t


def mesh_from_msh(path):
    """A 2-D `fem.Mesh` from the triangles of a gmsh file (z dropped, counter-clockwise orientation enforced): what
    obstacle_pg.py:64-65 obtains from `xdmf.read_mesh`."""
    from . import fem

    pts, cells, _ = read_msh(path)
    tri = cells["triangle"].copy()
    p = pts[:, :2]
    a, b, c = p[tri[:, 0]], p[tri[:, 1]], p[tri[:, 2]]
    neg = ((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])) < 0
    tri[neg] = tri[neg][:, [0, 2, 1]]
    used = np.unique(tri)
    remap = np.full(len(p), -1, dtype=np.int64)
    remap[used] = np.arange(len(used))
    return fem.Mesh(np.ascontiguousarray(p[used]), np.ascontiguousarray(remap[tri], dtype=np.int32))


def write_vtu(path, points, cells, point_data: dict | None = None, cell_data: dict | None = None, cell_type: str | None = None):
    """VTK XML unstructured grid (ASCII).  cells: (m,3) triangles, (m,6) quadratic triangles in the dof order used here
    (3 vertices, then the edge midpoint OPPOSITE each vertex - reordered to VTK's edge order) or (m,4) tetrahedra."""
    points = np.asarray(points, dtype=np.float64)
    if points.shape[1] == 2:
        points = np.concatenate([points, np.zeros((len(points), 1))], axis=1)
    cells = np.asarray(cells)
    k = cells.shape[1]
    cell_type = cell_type or {3: "triangle", 6: "triangle6", 4: "tetra"}[k]
    if cell_type == "triangle6":  # VTK: mid(0,1), mid(1,2), mid(2,0); here: opposite 0 = (1,2), opposite 1 = (0,2), opposite 2 = (0,1)
        cells = cells[:, [0, 1, 2, 5, 3, 4]]

    def arr(a, fmt):
        return "\n".join(" ".join(fmt % v for v in row) for row in np.atleast_2d(a))

    def data_block(d, n):
        out = []
        for name, v in (d or {}).items():
            v = np.asarray(v, dtype=np.float64)
            assert v.shape[0] == n, (name, v.shape, n)
            comps = 1 if v.ndim == 1 else v.shape[1]
            if comps == 2:  # ParaView wants 3-vectors
                v = np.concatenate([v, np.zeros((n, 1))], axis=1)
                comps = 3
            out.append(f'<DataArray type="Float64" Name="{name}" NumberOfComponents="{comps}" format="ascii">\n'
                       f'{arr(v.reshape(n, -1), "%.17g")}\n</DataArray>')
        return "\n".join(out)

    n, m = len(points), len(cells)
    xml = f'''<?xml version="1.0"?>
<VTKFile type="UnstructuredGrid" version="0.1" byte_order="LittleEndian">
<UnstructuredGrid><Piece NumberOfPoints="{n}" NumberOfCells="{m}">
<Points><DataArray type="Float64" NumberOfComponents="3" format="ascii">
{arr(points, "%.17g")}
</DataArray></Points>
<Cells>
<DataArray type="Int32" Name="connectivity" format="ascii">
{arr(cells, "%d")}
</DataArray>
<DataArray type="Int32" Name="offsets" format="ascii">
{arr((np.arange(1, m + 1) * cells.shape[1])[None, :], "%d")}
</DataArray>
<DataArray type="UInt8" Name="types" format="ascii">
{arr(np.full((1, m), _VTK[cell_type]), "%d")}
</DataArray>
</Cells>
<PointData>
{data_block(point_data, n)}
</PointData>
<CellData>
{data_block(cell_data, m)}
</CellData>
</Piece></UnstructuredGrid>
</VTKFile>
'''
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    Path(path).write_text(xml)
    return Path(path)
