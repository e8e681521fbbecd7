"""Mesh in / field out helpers (proximalgalerkin_amd/io.py): gmsh MSH 2.2 and 4.1 ASCII parsing, VTU writing."""
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from proximalgalerkin_amd import fem, io

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


GOLD = __import__("pathlib").Path(__file__).resolve().parent / "golden"


def test_order2_msh_and_inline_xdmf_give_the_same_vertex_mesh():
    """generate_mesh_gmsh.py:31 writes order-2 geometry; obstacle_pg.py:64-65 reads XDMF: both reduce to the same affine
    triangles here (mid-side nodes dropped, vertices renumbered compactly, counter-clockwise)."""
    from proximalgalerkin_amd import fem

    a = io.read_mesh(GOLD / "disk_h0.2_order2.msh")
    b = io.read_mesh(GOLD / "disk_h0.2.xdmf")
    ref = fem.create_disk(0.2)
    for m in (a, b):
        assert m.geometry.shape == ref.geometry.shape and m.cells.shape == ref.cells.shape
        assert np.allclose(m.geometry, ref.geometry, atol=1e-15) and np.array_equal(m.cells, ref.cells)
        x = m.geometry[m.cells]
        det = (x[:, 1, 0] - x[:, 0, 0]) * (x[:, 2, 1] - x[:, 0, 1]) - (x[:, 1, 1] - x[:, 0, 1]) * (x[:, 2, 0] - x[:, 0, 0])
        assert (det > 0).all()
    pts, cells, _ = io.read_msh(GOLD / "disk_h0.2_order2.msh")
    assert cells["triangle6"].shape[1] == 6 and len(pts) > len(a.geometry)  # the file really carries the mid-side nodes


def test_tet_mesh_with_facet_tags_from_order2_msh_and_xdmf():
    """signorini_dolfinx.py:406-409 (read_mesh + read_meshtags "facet_tags") on the layout mesh_generation.create_half_sphere
    writes: order-2 tetrahedra, tagged boundary triangles."""
    from proximalgalerkin_amd import signorini as sg

    ref = sg.create_unit_cube(3, 2, 2)
    rt, _ = sg.native_tags(ref)
    for f in ("cube_3x2x2_order2.msh", "cube_3x2x2.xdmf"):
        mesh, mt = io.read_tet_mesh(GOLD / f)
        assert np.allclose(mesh.geometry, ref.geometry, atol=1e-15)
        key = lambda c: np.unique(np.sort(c, axis=1), axis=0)  # noqa: E731
        assert np.array_equal(key(mesh.cells), key(ref.cells))
        for tag in (1, 2):
            assert np.array_equal(key(mt.find(tag)), key(rt.find(tag)))
        x = mesh.geometry[mesh.cells]
        vol = np.einsum("ij,ij->i", np.cross(x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]), x[:, 3] - x[:, 0])
        assert (vol > 0).all() and np.isclose(vol.sum() / 6.0, 1.0)


def test_hdf5_backed_xdmf_written_by_the_real_libhdf5_is_read():
    """tests/golden/dolfinx_like_square.{xdmf,h5}: the .h5 was written by libhdf5 itself (tools/make_h5_fixtures.c: DOLFINx's dataset
    names, contiguous, default format bounds), the .xdmf is what XDMFFile.write_mesh + write_meshtags put next to it.  The
    pure-Python reader (proximalgalerkin_amd/h5.py) must return exactly the arrays the C program wrote - `obstacle_pg.py -f
    mesh.xdmf` (obstacle_pg.py:64-65) on a file of the reference's own pipeline."""
    from proximalgalerkin_amd import h5

    f = h5.H5File(GOLD / "dolfinx_like_square.h5")
    assert f.keys("/") == ["Mesh", "MeshTags"] and f.keys("/Mesh/mesh") == ["geometry", "topology"]
    g, t = f["/Mesh/mesh/geometry"], f["/Mesh/mesh/topology"]
    nx, ny = 5, 4
    i, j = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), indexing="xy")
    assert g.dtype == np.float64 and np.array_equal(g, np.stack([i.ravel() / nx, j.ravel() / ny], axis=1))
    assert t.dtype == np.int64 and t.shape == (40, 3) and t[3].tolist() == [1, 7, 8] and t.max() == 29
    assert f["/MeshTags/facet_tags/Values"].tolist() == [7] * 4 + [9] * 5
    pts, cells, tags = io.read_xdmf(GOLD / "dolfinx_like_square.xdmf")
    assert np.array_equal(pts, g) and np.array_equal(cells["triangle"], t)
    assert tags["facet_tags"][0] == "line" and tags["facet_tags"][2].tolist() == [7] * 4 + [9] * 5
    mesh = io.read_mesh(GOLD / "dolfinx_like_square.xdmf")
    x = mesh.geometry[mesh.cells]
    area = 0.5 * ((x[:, 1, 0] - x[:, 0, 0]) * (x[:, 2, 1] - x[:, 0, 1]) - (x[:, 1, 1] - x[:, 0, 1]) * (x[:, 2, 0] - x[:, 0, 0]))
    assert (area > 0).all() and np.isclose(area.sum(), 1.0)


def test_hdf5_xdmf_round_trip_and_unsupported_features(tmp_path):
    from proximalgalerkin_amd import fem, h5

    msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (7, 5))
    io.write_xdmf_mesh(tmp_path / "m.xdmf", msh, encoding="HDF5")
    assert (tmp_path / "m.h5").exists() and 'Format="HDF">m.h5:/Mesh/mesh/geometry' in (tmp_path / "m.xdmf").read_text()
    back = io.read_mesh(tmp_path / "m.xdmf")
    # read_mesh orients every triangle counter-clockwise: compare the vertex sets of the cells
    assert np.array_equal(back.geometry, msh.geometry) and np.array_equal(np.sort(back.cells, axis=1), np.sort(msh.cells, axis=1))
    # dtypes and shapes the writer covers, an empty dataset, nested groups
    data = {"/a/b/c": np.arange(12, dtype=np.int32).reshape(3, 4), "/a/x": np.linspace(0, 1, 5, dtype=np.float32), "/y": np.zeros((0, 3)),
            "/a/b/d": np.array([2**40, -3], dtype=np.int64)}
    h5.write(tmp_path / "t.h5", data)
    f = h5.H5File(tmp_path / "t.h5")
    assert f.keys("/a") == ["b", "x"]
    for k, v in data.items():
        assert f[k].dtype == v.dtype and np.array_equal(f[k], v)
    with pytest.raises(KeyError):
        f["/a/nothing"]
    bad = bytearray((tmp_path / "t.h5").read_bytes())
    bad[8] = 2  # superblock version 2 (libver='latest'): refused by name, never misread
    (tmp_path / "v2.h5").write_bytes(bytes(bad))
    with pytest.raises(NotImplementedError, match="superblock version 2"):
        h5.H5File(tmp_path / "v2.h5")


def test_generate_disk_writes_the_reference_named_xdmf_files(tmp_path):
    """examples/01_obstacle_problem/generate_mesh_gmsh.py: `generate_disk(filename, res, order, refinement_level)` writes
    `<stem>_<level>.xdmf` (the reference's naming, generate_mesh_gmsh.py:40), every refinement level halves the mesh size, and the
    file reads back as the mesh `fem.create_disk` makes (HDF5-backed XDMF through io.write_xdmf_mesh / io.read_mesh)."""
    import importlib.util
    import pathlib

    from proximalgalerkin_amd import fem, io

    src = pathlib.Path(__file__).resolve().parents[1] / "examples" / "01_obstacle_problem" / "generate_mesh_gmsh.py"
    spec = importlib.util.spec_from_file_location("generate_mesh_gmsh", src)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sizes = []
    for level in (0, 1):
        out = mod.generate_disk(tmp_path / "meshes" / "disk.xdmf", res=0.4, order=2, refinement_level=level)
        assert out.name == f"disk_{level}.xdmf" and out.exists() and out.with_suffix(".h5").exists()  # HDF5-backed, as upstream
        m = io.read_mesh(out)
        ref = fem.create_disk(0.4 / 2**level)
        assert np.array_equal(m.cells, ref.cells) and np.abs(m.geometry - ref.geometry).max() == 0.0
        sizes.append(m.num_cells)
    assert 3.0 < sizes[1] / sizes[0] < 5.0  # h halves: about four times the cells


def test_order2_triangles_keep_their_mid_side_nodes():
    """io.read_mesh on a gmsh mesh of element order 2 (round 5): the vertices form the fem.Mesh as before, the mid-side nodes are kept
    per edge (`mesh.midside`, edge numbering of mesh.edges()) - on the unit circle for the boundary edges of the committed disk, at
    the edge midpoints inside - and `mesh.flattened()` is the affine mesh a degree-1 run uses."""
    m = io.read_mesh(GOLD / "disk_h0.2_order2.msh")
    assert m.curved and m.midside.shape == (len(m.edges()[0]), 2)
    e, ce = m.edges()
    straight = 0.5 * (m.geometry[e[:, 0]] + m.geometry[e[:, 1]])
    boundary = np.bincount(ce.ravel(), minlength=len(e)) == 1
    assert np.allclose(np.linalg.norm(m.midside[boundary], axis=1), 1.0, atol=1e-12)
    assert np.abs(m.midside[~boundary] - straight[~boundary]).max() < 1e-14
    assert np.linalg.norm(m.midside[boundary] - straight[boundary], axis=1).min() > 1e-3
    xq, geo = m.geometry_at(fem.quadrature_rule("triangle", 6)[0])
    w = fem.quadrature_rule("triangle", 6)[1]
    assert abs((geo[..., 0] * w).sum() - np.pi) < 2e-5  # the curved cells tile the disk (the polygon misses 2e-2)
    flat = m.flattened()
    assert not flat.curved and np.array_equal(flat.cells, m.cells) and np.array_equal(flat.geometry, m.geometry)
    assert not io.read_mesh(GOLD / "disk_h0.2.xdmf").curved
