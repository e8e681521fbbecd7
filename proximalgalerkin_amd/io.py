"""Mesh in / field out - host-side conveniences around the solvers, no GPU involved.

The reference reads its meshes from XDMF/HDF5 written by gmsh scripts (examples/01_obstacle_problem/generate_mesh_gmsh.py:12-43,
obstacle_pg.py:64-65) and writes results with VTXWriter / XDMFFile (obstacle_pg.py:239-243,
gradient_constraint_dolfinx.py:145-158, signorini_dolfinx.py:293-299).  No HDF5 binding and no ADIOS2 exist offline: HDF5 files of
the kind DOLFINx writes are read and written by proximalgalerkin_amd/h5.py, results go to VTU:

    read_msh(path)                      gmsh MSH 2.2 / 4.1 ASCII  -> (points, cells_by_type, cell_tags_by_type)
    read_xdmf(path, name)               XDMF with the heavy data in HDF5 (DOLFINx's default: read through the pure-Python reader
                                        proximalgalerkin_amd/h5.py - contiguous, unfiltered datasets, default format bounds) or
                                        INLINE (dolfinx.io.XDMFFile(..., encoding=XDMFFile.Encoding.ASCII))
    read_mesh(path) / read_tet_mesh     -> fem.Mesh / (TetMesh, MeshTags) from either format; ORDER-2 geometry (generate_mesh_gmsh.py:31,
                                        mesh_generation.py:88,158): 6-node triangles keep their mid-side nodes (`mesh.midside`:
                                        curved cells in example 01), 10-node tetrahedra their mid-edge nodes
                                        (`TetMesh.midside`: isoparametric P2 in example 02)
    write_vtu(path, points, cells, ...) VTK unstructured grid (XML, ASCII) with point / cell data; linear and quadratic
                                        triangles, linear tetrahedra
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

# gmsh element type -> (name, nodes)
_GMSH = {1: ("line", 2), 2: ("triangle", 3), 4: ("tetra", 4), 8: ("line3", 3), 9: ("triangle6", 6), 11: ("tetra10", 10),
         15: ("vertex", 1)}
_VTK = {"triangle": 5, "triangle6": 22, "tetra": 10, "line": 3, "vertex": 1, "hexahedron": 12}


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


# XDMF TopologyType (case-insensitive; DOLFINx writes "Triangle", "Triangle_6", "Tetrahedron", "Tetrahedron_10") -> (name, nodes)
_XDMF = {"triangle": ("triangle", 3), "triangle_6": ("triangle6", 6), "tri_6": ("triangle6", 6), "tetrahedron": ("tetra", 4),
         "tetrahedron_10": ("tetra10", 10), "tet_10": ("tetra10", 10), "polyline": ("line", 2), "edge_3": ("line3", 3)}


def read_xdmf(path, name: str = "mesh"):
    """Grid `name` of an XDMF file whose DataItems carry their numbers inline (Format="XML").  Returns (points (n, 2|3),
    {cell type: (m, k) int32 connectivity}, {grid name: (cell type, connectivity, values)} for every further grid with an
    Attribute - DOLFINx's meshtags grids, e.g. "facet_tags" of examples/02_signorini/signorini_dolfinx.py:407)."""
    import xml.etree.ElementTree as ET

    root = ET.parse(str(path)).getroot()

    h5files = {}

    def numbers(item, dtype):
        fmt = (item.get("Format") or "XML").upper()
        dims = [int(v) for v in item.get("Dimensions").split()]
        if fmt in ("HDF", "HDF5"):  # "file.h5:/Mesh/mesh/geometry" - what XDMFFile writes by default (obstacle_pg.py:64-65)
            from . import h5

            fname, _, dset = (item.text or "").strip().partition(":")
            hp = Path(path).parent / fname
            if hp not in h5files:
                h5files[hp] = h5.H5File(hp)
            return np.asarray(h5files[hp][dset], dtype=dtype).reshape(dims)
        if fmt != "XML":
            raise NotImplementedError(f"{path}: DataItem Format=\"{item.get('Format')}\"")
        return np.array((item.text or "").split(), dtype=dtype).reshape(dims)

    grids = {g.get("Name"): g for g in root.iter("Grid") if g.find("Topology") is not None}
    if name not in grids:
        raise KeyError(f"{path}: no grid named {name!r} (found {sorted(grids)})")

    def topo(g):
        t = g.find("Topology")
        key = (t.get("TopologyType") or t.get("Type") or "").lower()
        if key not in _XDMF:
            raise NotImplementedError(f"{path}: TopologyType {key!r}")
        cname, k = _XDMF[key]
        return cname, np.ascontiguousarray(numbers(t.find("DataItem"), np.int64).reshape(-1, k), dtype=np.int32)

    g = grids[name]
    pts = numbers(g.find("Geometry").find("DataItem"), np.float64)
    cname, conn = topo(g)
    tags = {}
    for gname, gg in grids.items():
        att = gg.find("Attribute")
        if gname == name or att is None:
            continue
        tname, tconn = topo(gg)
        tags[gname] = (tname, tconn, np.asarray(numbers(att.find("DataItem"), np.float64)).ravel().astype(np.int32))
    return pts, {cname: conn}, tags


def _triangles(pts, cells):
    """vertex triangles of a 2-D mesh given as linear or quadratic (order-2 geometry) triangles, counter-clockwise, compact"""
    from . import fem

    six = None if "triangle" in cells else np.asarray(cells["triangle6"]).copy()
    tri = (cells["triangle"] if six is None else six[:, :3]).copy()
    p = np.asarray(pts)[:, :2]
    a, b, c = p[tri[:, 0]], p[tri[:, 1]], p[tri[:, 2]]
    neg = ((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])) < 0
    tri[neg] = tri[neg][:, [0, 2, 1]]
    used = np.unique(tri)  # the vertices; the mid-side nodes of an order-2 geometry become the mesh's `midside` attribute
    remap = np.full(len(p), -1, dtype=np.int64)
    remap[used] = np.arange(len(used))
    vt = np.ascontiguousarray(remap[tri], dtype=np.int32)
    if six is None:
        return fem.Mesh(np.ascontiguousarray(p[used]), vt)
    # order-2 geometry (round 5): gmsh / VTK / XDMF `triangle6` = v0 v1 v2 m01 m12 m20.  After the orientation swap v1 <-> v2 the
    # mid-side nodes of the new edges (0,1), (1,2), (2,0) are the old m20, m12, m01.  One node per edge -> the mesh's edge numbering.
    mids = six[:, 3:6].copy()
    mids[neg] = mids[neg][:, [2, 1, 0]]
    mesh = fem.Mesh(np.ascontiguousarray(p[used]), vt)
    edges, cell_edges = mesh.edges()  # local edge i is OPPOSITE local vertex i: (1,2), (0,2), (0,1)  <-  m12, m20, m01
    midside = np.empty((len(edges), 2))
    midside[cell_edges[:, 0]] = p[mids[:, 1]]
    midside[cell_edges[:, 1]] = p[mids[:, 2]]
    midside[cell_edges[:, 2]] = p[mids[:, 0]]
    return fem.Mesh(mesh.geometry, mesh.cells, midside=midside)


def mesh_from_msh(path):
    """A 2-D `fem.Mesh` from the triangles of a gmsh file (z dropped, counter-clockwise orientation enforced; the mid-side nodes of
    an order-2 geometry are KEPT as `mesh.midside` since round 5 - isoparametric P2 cells for `-p 2`): what
    obstacle_pg.py:64-65 obtains from `xdmf.read_mesh`."""
    pts, cells, _ = read_msh(path)
    return _triangles(pts, cells)


def read_mesh(path, name: str = "mesh"):
    """`xdmf.read_mesh(name="mesh")` of obstacle_pg.py:64-65 for a 2-D triangle mesh: .xdmf (inline data) or gmsh .msh."""
    path = Path(path)
    if path.suffix.lower() == ".xdmf":
        pts, cells, _ = read_xdmf(path, name)
        return _triangles(pts, cells)
    return mesh_from_msh(path)


def read_tet_mesh(path, name: str = "mesh", tags_name: str = "facet_tags"):
    """(TetMesh, MeshTags) of example 02's `file` branch (signorini_dolfinx.py:406-409: read_mesh + read_meshtags "facet_tags";
    the half-sphere of lvpp/mesh_generation.py:86-168 has order-2 geometry): tetrahedra and tagged boundary triangles from a
    gmsh .msh file (physical groups) or an inline-data XDMF file.  10-node tetrahedra keep their mid-edge nodes (`mesh.midside`, round
    5: isoparametric P2 in example 02; a degree-1 run uses the vertices)."""
    from .signorini import MeshTags, TetMesh

    path = Path(path)
    if path.suffix.lower() == ".xdmf":
        pts, cells, tagged = read_xdmf(path, name)
        if tags_name not in tagged:
            raise KeyError(f"{path}: no meshtags grid {tags_name!r} (found {sorted(tagged)})")
        tname, fconn, fval = tagged[tags_name]
    else:
        pts, cells, tags = read_msh(path)
        tname = "triangle" if "triangle" in cells else "triangle6"
        fconn, fval = cells[tname], tags[tname]
    ten = None if "tetra" in cells else np.asarray(cells["tetra10"]).copy()
    tet = (cells["tetra"] if ten is None else ten[:, :4]).copy()
    fconn = fconn[:, :3]
    p = np.asarray(pts, dtype=np.float64)[:, :3]
    # positive orientation
    a, b, c, d = (p[tet[:, k]] for k in range(4))
    vol = np.einsum("ij,ij->i", np.cross(b - a, c - a), d - a)
    tet[vol < 0] = tet[vol < 0][:, [0, 2, 1, 3]]
    used = np.unique(tet)
    remap = np.full(len(p), -1, dtype=np.int64)
    remap[used] = np.arange(len(used))
    midside = None
    if ten is not None:
        # order-2 geometry (round 5): the six mid-edge nodes of a 10-node tetrahedron follow the vertices in the FILE's edge order -
        # gmsh: (0,1) (1,2) (0,2) (0,3) (2,3) (1,3); XDMF / VTK: (0,1) (1,2) (0,2) (0,3) (1,3) (2,3).  An edge is its vertex pair,
        # so the node of every edge goes to its place in TetMesh.edges()' numbering whatever the cell's orientation.
        order = ((0, 1), (1, 2), (0, 2), (0, 3), (1, 3), (2, 3)) if path.suffix.lower() == ".xdmf" else ((0, 1), (1, 2), (0, 2), (0, 3), (2, 3), (1, 3))
        nvu = len(used)
        pair = np.sort(np.concatenate([remap[ten[:, list(e)]] for e in order]), axis=1)
        node = np.concatenate([ten[:, 4 + k] for k in range(6)])
        key = pair[:, 0] * nvu + pair[:, 1]
        ukey, first = np.unique(key, return_index=True)
        midside = np.ascontiguousarray(p[node[first]])  # (edges in sorted (min, max) order = TetMesh.edges())
    mesh = TetMesh(np.ascontiguousarray(p[used]), np.ascontiguousarray(remap[tet], dtype=np.int32), midside)
    tagged_facets = {int(t): np.ascontiguousarray(remap[fconn[fval == t]], dtype=np.int32) for t in np.unique(fval) if t != 0}
    return mesh, MeshTags(tagged_facets)


def write_xdmf_mesh(path, mesh, name: str = "mesh", encoding: str = "ASCII"):
    """XDMFFile.write_mesh (generate_mesh_gmsh.py:41-43) for a triangular fem.Mesh.  encoding "ASCII": inline (Format="XML") data
    items; "HDF5" (DOLFINx's default): the heavy data in `<stem>.h5` as /Mesh/<name>/geometry (float64) and /Mesh/<name>/topology
    (int64), written by proximalgalerkin_amd/h5.py.  `read_xdmf` / `read_mesh` accept both."""
    path = Path(path)
    p, t = np.asarray(mesh.geometry), np.asarray(mesh.cells)
    path.parent.mkdir(parents=True, exist_ok=True)
    if encoding.upper() in ("HDF5", "HDF"):
        from . import h5

        h5name = path.with_suffix(".h5")
        h5.write(h5name, {f"/Mesh/{name}/geometry": np.asarray(p, dtype=np.float64), f"/Mesh/{name}/topology": np.asarray(t, dtype=np.int64)})
        with open(path, "w") as f:
            f.write('<?xml version="1.0"?>\n<!DOCTYPE Xdmf SYSTEM "Xdmf.dtd" []>\n<Xdmf Version="3.0" xmlns:xi="http://www.w3.org/2001/XInclude">'
                    '<Domain>\n<Grid Name="%s" GridType="Uniform">\n' % name)
            f.write('<Topology TopologyType="Triangle" NumberOfElements="%d" NodesPerElement="3">\n<DataItem Dimensions="%d 3" '
                    'NumberType="Int" Format="HDF">%s:/Mesh/%s/topology</DataItem></Topology>\n' % (len(t), len(t), h5name.name, name))
            f.write('<Geometry GeometryType="XY"><DataItem Dimensions="%d 2" Format="HDF">%s:/Mesh/%s/geometry</DataItem></Geometry>\n'
                    '</Grid>\n</Domain></Xdmf>\n' % (len(p), h5name.name, name))
        return
    with open(path, "w") as f:
        f.write('<?xml version="1.0"?>\n<Xdmf Version="3.0"><Domain>\n<Grid Name="%s" GridType="Uniform">\n' % name)
        f.write('<Topology TopologyType="Triangle" NumberOfElements="%d" NodesPerElement="3">\n<DataItem Dimensions="%d 3" NumberType="Int" '
                'Format="XML">\n%s\n</DataItem></Topology>\n' % (len(t), len(t), "\n".join(" ".join(map(str, c)) for c in t)))
        f.write('<Geometry GeometryType="XY"><DataItem Dimensions="%d 2" Format="XML">\n%s\n</DataItem></Geometry>\n</Grid>\n</Domain></Xdmf>\n'
                % (len(p), "\n".join("%.17g %.17g" % tuple(x) for x in p)))


def write_xdmf_tet(path, mesh, facet_tags, tags=None, name: str = "mesh", tags_name: str = "facet_tags"):
    """XDMFFile.write_mesh + write_meshtags of `examples/02_signorini/generate_mesh.py:13-17` for a TetMesh and its MeshTags, with
    inline (Format="XML") data items - the encoding `read_xdmf` accepts (dolfinx: `XDMFFile(..., encoding=XDMFFile.Encoding.ASCII)`)."""
    path = Path(path)
    p, t = np.asarray(mesh.geometry), np.asarray(mesh.cells)
    ttype, npe = "Tetrahedron", 4
    if getattr(mesh, "midside", None) is not None:  # order-2 geometry: Tetrahedron_10 in XDMF / VTK node order, mid-edge nodes appended
        e = mesh.edges()
        nv = len(p)
        key = e[:, 0] * nv + e[:, 1]
        mids = []
        for a, b in ((0, 1), (1, 2), (0, 2), (0, 3), (1, 3), (2, 3)):
            pr = np.sort(t[:, [a, b]].astype(np.int64), axis=1)
            mids.append(nv + np.searchsorted(key, pr[:, 0] * nv + pr[:, 1]))
        t = np.concatenate([t, np.stack(mids, axis=1)], axis=1)
        p = np.concatenate([p, mesh.midside])
        ttype, npe = "Tetrahedron_10", 10
    tags = sorted(facet_tags._t) if tags is None else list(tags)
    fc = [np.asarray(facet_tags.find(k)) for k in tags]
    allf = np.concatenate(fc) if fc else np.zeros((0, 3), dtype=np.int32)
    vals = np.concatenate([np.full(len(f), k) for f, k in zip(fc, tags)]) if fc else np.zeros(0, dtype=np.int64)
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, "w") as f:
        f.write('<?xml version="1.0"?>\n<Xdmf Version="3.0"><Domain>\n<Grid Name="%s" GridType="Uniform">\n' % name)
        f.write('<Topology TopologyType="%s" NumberOfElements="%d" NodesPerElement="%d">\n<DataItem Dimensions="%d %d" NumberType="Int" '
                'Format="XML">\n%s\n</DataItem></Topology>\n' % (ttype, len(t), npe, len(t), npe, "\n".join(" ".join(map(str, c)) for c in t)))
        f.write('<Geometry GeometryType="XYZ"><DataItem Dimensions="%d 3" Format="XML">\n%s\n</DataItem></Geometry>\n</Grid>\n'
                % (len(p), "\n".join("%.17g %.17g %.17g" % tuple(x) for x in p)))
        f.write('<Grid Name="%s" GridType="Uniform">\n<Topology TopologyType="Triangle" NumberOfElements="%d" NodesPerElement="3">\n'
                '<DataItem Dimensions="%d 3" NumberType="Int" Format="XML">\n%s\n</DataItem></Topology>\n'
                % (tags_name, len(allf), len(allf), "\n".join(" ".join(map(str, c)) for c in allf)))
        f.write('<Attribute Name="%s" AttributeType="Scalar" Center="Cell"><DataItem Dimensions="%d 1" Format="XML">\n%s\n'
                '</DataItem></Attribute>\n</Grid>\n</Domain></Xdmf>\n' % (tags_name, len(vals), "\n".join(map(str, vals))))


def write_vtu(path, points, cells, point_data: dict | None = None, cell_data: dict | None = None, cell_type: str | None = None):
    """VTK XML unstructured grid (ASCII).  cells: (m,3) triangles, (m,6) quadratic triangles in the dof order used here
    (3 vertices, then the edge midpoint OPPOSITE each vertex - reordered to VTK's edge order) or (m,4) tetrahedra."""
    points = np.asarray(points, dtype=np.float64)
    if points.shape[1] == 2:
        points = np.concatenate([points, np.zeros((len(points), 1))], axis=1)
    cells = np.asarray(cells)
    k = cells.shape[1]
    cell_type = cell_type or {3: "triangle", 6: "triangle6", 4: "tetra", 8: "hexahedron"}[k]  # hexahedron: VTK corner order
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
