#!/usr/bin/env python3
"""Writes the small mesh FILES the I/O tests read (tests/golden/): data files in the formats the reference's mesh scripts produce.
  disk_h0.2_order2.msh    gmsh MSH 2.2, unit disk, 6-node triangles (order-2 geometry: generate_mesh_gmsh.py:31 `setOrder(2)`), with
                          the boundary mid-side nodes on the circle
  disk_h0.2.xdmf          the same vertices as linear triangles, XDMF with inline (ASCII-encoded) data, grid "mesh"
  cube_3x2x2_order2.msh   MSH 4.1, unit cube of 10-node tetrahedra + 6-node boundary triangles in physical groups 1 (top, z = 1)
                          and 2 (bottom, z = 0) - the layout of lvpp.mesh_generation.create_half_sphere's output (facet tags)
  cube_3x2x2.xdmf         linear tetrahedra + a "facet_tags" grid, inline data (what signorini_dolfinx.py:406-409 reads)
"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from proximalgalerkin_amd import fem  # noqa: E402
from proximalgalerkin_amd import signorini as sg  # noqa: E402

OUT = ROOT / "tests" / "golden"


def midpoints(pts, pairs):
    """unique mid-side nodes for the vertex pairs (m, 2) -> (new points, index per pair)"""
    key = np.sort(pairs, axis=1)
    uk, inv = np.unique(key, axis=0, return_inverse=True)
    return 0.5 * (pts[uk[:, 0]] + pts[uk[:, 1]]), inv.ravel(), uk


def disk():
    m = fem.create_disk(0.2)
    p, t = m.geometry, m.cells
    mids, inv, uk = midpoints(p, np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]))
    # boundary mid-side nodes go onto the circle (curved geometry)
    on = np.isclose(np.hypot(*p[uk[:, 0]].T), 1.0) & np.isclose(np.hypot(*p[uk[:, 1]].T), 1.0)
    mids[on] /= np.hypot(*mids[on].T)[:, None]
    nv, nc = len(p), len(t)
    allp = np.concatenate([p, mids])
    t6 = np.concatenate([t, nv + inv.reshape(3, nc).T], axis=1)  # gmsh order: v0 v1 v2 e01 e12 e20
    with open(OUT / "disk_h0.2_order2.msh", "w") as f:
        f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % len(allp))
        for i, (x, y) in enumerate(allp):
            f.write("%d %.17g %.17g 0\n" % (i + 1, x, y))
        f.write("$EndNodes\n$Elements\n%d\n" % nc)
        for i, c in enumerate(t6):
            f.write("%d 9 2 1 1 %s\n" % (i + 1, " ".join(str(v + 1) for v in c)))
        f.write("$EndElements\n")
    with open(OUT / "disk_h0.2.xdmf", "w") as f:
        f.write('<?xml version="1.0"?>\n<Xdmf Version="3.0"><Domain>\n<Grid Name="mesh" GridType="Uniform">\n')
        f.write('<Topology TopologyType="Triangle" NumberOfElements="%d" NodesPerElement="3">\n<DataItem Dimensions="%d 3" NumberType="Int" '
                'Format="XML">\n%s\n</DataItem></Topology>\n' % (nc, nc, "\n".join(" ".join(map(str, c)) for c in t)))
        f.write('<Geometry GeometryType="XY"><DataItem Dimensions="%d 2" Format="XML">\n%s\n</DataItem></Geometry>\n</Grid>\n</Domain></Xdmf>\n'
                % (nv, "\n".join("%.17g %.17g" % (x, y) for x, y in p)))


def cube():
    m = sg.create_unit_cube(3, 2, 2)
    p, t = m.geometry, m.cells
    mt, _ = sg.native_tags(m)
    faces = {1: mt.find(1), 2: mt.find(2)}
    edges = [(0, 1), (1, 2), (0, 2), (0, 3), (2, 3), (1, 3)]  # gmsh 10-node tetrahedron: e01 e12 e02 e03 e23 e13
    tet_pairs = np.concatenate([t[:, list(e)] for e in edges])
    tri_pairs = np.concatenate([np.concatenate([fc[:, [0, 1]], fc[:, [1, 2]], fc[:, [2, 0]]]) for fc in faces.values()])
    mids, inv, _ = midpoints(p, np.concatenate([tet_pairs, tri_pairs]))
    nv, nc = len(p), len(t)
    allp = np.concatenate([p, mids])
    t10 = np.concatenate([t, nv + inv[: 6 * nc].reshape(6, nc).T], axis=1)
    off = 6 * nc
    f6 = {}
    for tag, fc in faces.items():
        k = len(fc)
        f6[tag] = np.concatenate([fc, nv + inv[off: off + 3 * k].reshape(3, k).T], axis=1)
        off += 3 * k
    with open(OUT / "cube_3x2x2_order2.msh", "w") as f:
        f.write("$MeshFormat\n4.1 0 8\n$EndMeshFormat\n")
        # entities: surfaces 1, 2 with physical tags 1, 2; volume 1 with physical tag 7
        f.write("$Entities\n0 0 2 1\n1 0 0 1 1 1 1 1 1 0\n2 0 0 0 1 1 0 1 2 0\n1 0 0 0 1 1 1 1 7 0\n$EndEntities\n")
        f.write("$Nodes\n1 %d 1 %d\n3 1 0 %d\n" % (len(allp), len(allp), len(allp)))
        f.write("\n".join(str(i + 1) for i in range(len(allp))) + "\n")
        f.write("\n".join("%.17g %.17g %.17g" % tuple(x) for x in allp) + "\n$EndNodes\n")
        nel = nc + sum(len(v) for v in f6.values())
        f.write("$Elements\n3 %d 1 %d\n" % (nel, nel))
        e = 1
        for tag in (1, 2):
            f.write("2 %d 9 %d\n" % (tag, len(f6[tag])))
            for c in f6[tag]:
                f.write("%d %s\n" % (e, " ".join(str(v + 1) for v in c)))
                e += 1
        f.write("3 1 11 %d\n" % nc)
        for c in t10:
            f.write("%d %s\n" % (e, " ".join(str(v + 1) for v in c)))
            e += 1
        f.write("$EndElements\n")
    allf = np.concatenate([faces[1], faces[2]])
    vals = np.concatenate([np.full(len(faces[1]), 1), np.full(len(faces[2]), 2)])
    with open(OUT / "cube_3x2x2.xdmf", "w") as f:
        f.write('<?xml version="1.0"?>\n<Xdmf Version="3.0"><Domain>\n<Grid Name="mesh" GridType="Uniform">\n')
        f.write('<Topology TopologyType="Tetrahedron" NumberOfElements="%d" NodesPerElement="4">\n<DataItem Dimensions="%d 4" NumberType="Int" '
                'Format="XML">\n%s\n</DataItem></Topology>\n' % (nc, nc, "\n".join(" ".join(map(str, c)) for c in t)))
        f.write('<Geometry GeometryType="XYZ"><DataItem Dimensions="%d 3" Format="XML">\n%s\n</DataItem></Geometry>\n</Grid>\n'
                % (nv, "\n".join("%.17g %.17g %.17g" % tuple(x) for x in p)))
        f.write('<Grid Name="facet_tags" GridType="Uniform">\n<Topology TopologyType="Triangle" NumberOfElements="%d" NodesPerElement="3">\n'
                '<DataItem Dimensions="%d 3" NumberType="Int" Format="XML">\n%s\n</DataItem></Topology>\n'
                % (len(allf), len(allf), "\n".join(" ".join(map(str, c)) for c in allf)))
        f.write('<Attribute Name="facet_tags" AttributeType="Scalar" Center="Cell"><DataItem Dimensions="%d 1" Format="XML">\n%s\n'
                '</DataItem></Attribute>\n</Grid>\n</Domain></Xdmf>\n' % (len(vals), "\n".join(map(str, vals))))


if __name__ == "__main__":
    disk()
    cube()
    print("wrote", sorted(q.name for q in OUT.glob("*.msh")) + sorted(q.name for q in OUT.glob("*.xdmf")))
