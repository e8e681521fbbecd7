"""Mesh in / field out helpers (proximalgalerkin_amd/io.py): gmsh MSH 2.2 and 4.1 ASCII parsing, VTU writing."""
import xml.etree.ElementTree as ET

import numpy as np

from proximalgalerkin_amd import io

MSH22 = """$MeshFormat
2.2 0 8
$EndMeshFormat
$Nodes
4
1 0 0 0
2 1 0 0
3 1 1 0
4 0 1 0
$EndNodes
$Elements
4
1 1 2 7 1 1 2
2 1 2 7 1 2 3
3 2 2 9 1 1 2 3
4 2 2 9 1 1 4 3
$EndElements
"""

MSH41 = """$MeshFormat
4.1 0 8
$EndMeshFormat
$Entities
0 0 1 0
1 0 0 0 1 1 0 1 5 0
$EndEntities
$Nodes
1 4 1 4
2 1 0 4
1
2
3
4
0 0 0
1 0 0
1 1 0
0 1 0
$EndNodes
$Elements
1 2 1 2
2 1 2 2
1 1 2 3
2 1 3 4
$EndElements
"""


def test_read_msh_both_formats(tmp_path):
    for name, text, tag in (("a.msh", MSH22, 9), ("b.msh", MSH41, 5)):
        f = tmp_path / name
        f.write_text(text)
        pts, cells, tags = io.read_msh(f)
        assert pts.shape == (4, 3) and cells["triangle"].shape == (2, 3)
        assert np.array_equal(cells["triangle"][0], [0, 1, 2]) and np.all(tags["triangle"] == tag)
        msh = io.mesh_from_msh(f)
        assert msh.num_vertices == 4 and msh.num_cells == 2
        x = msh.geometry[msh.cells]
        det = (x[:, 1, 0] - x[:, 0, 0]) * (x[:, 2, 1] - x[:, 0, 1]) - (x[:, 1, 1] - x[:, 0, 1]) * (x[:, 2, 0] - x[:, 0, 0])
        assert np.all(det > 0) and abs(det.sum() / 2 - 1.0) < 1e-14
        assert len(msh.exterior_vertices()) == 4


def test_write_vtu_is_wellformed(tmp_path):
    from proximalgalerkin_amd import fem

    msh = fem.create_unit_square(3, 2)
    u = msh.geometry[:, 0] ** 2
    g = np.stack([2 * msh.geometry[:, 0], np.zeros(msh.num_vertices)], axis=1)
    f = io.write_vtu(tmp_path / "out" / "u.vtu", msh.geometry, msh.cells, {"u": u, "grad": g}, {"cell_id": np.arange(msh.num_cells)})
    root = ET.parse(f).getroot()
    piece = root.find("UnstructuredGrid/Piece")
    assert int(piece.get("NumberOfPoints")) == msh.num_vertices and int(piece.get("NumberOfCells")) == msh.num_cells
    names = [d.get("Name") for d in piece.find("PointData")]
    assert names == ["u", "grad"]
    vals = np.array(piece.find("PointData")[0].text.split(), dtype=float)
    assert np.array_equal(vals, u)
    types = np.array(piece.find("Cells")[2].text.split(), dtype=int)
    assert np.all(types == 5)
    # quadratic triangles: midpoints reordered to VTK's edge order
    V = fem.FunctionSpace(msh, 2, 1)
    f2 = io.write_vtu(tmp_path / "p2.vtu", V.dof_coordinates(), V.cell_dofs(), {"x": V.dof_coordinates()[:, 0]})
    piece = ET.parse(f2).getroot().find("UnstructuredGrid/Piece")
    conn = np.array(piece.find("Cells")[0].text.split(), dtype=int).reshape(-1, 6)
    X = V.dof_coordinates()
    assert np.allclose(X[conn[:, 3]], 0.5 * (X[conn[:, 0]] + X[conn[:, 1]]))
    assert np.allclose(X[conn[:, 4]], 0.5 * (X[conn[:, 1]] + X[conn[:, 2]]))
