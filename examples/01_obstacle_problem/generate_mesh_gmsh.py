"""Unit-disk meshes of example 01 - counterpart of the reference's generate_mesh_gmsh.py (`generate_disk(filename, res, order,
refinement_level)`, :12-43; `__main__` writes meshes/disk_0 ... disk_3.xdmf with res = 0.1, the files its README, CI and
compare_all.py's default `-P ./meshes/disk_3.xdmf` use).  gmsh is not available offline: the disk comes from the package's own
Delaunay mesher (`fem.create_disk`), every gmsh refinement halving the mesh size; the file is XDMF with its heavy data in
`<name>.h5` (HDF5, DOLFINx's default encoding and dataset names, written by proximalgalerkin_amd/h5.py), which `obstacle_pg.py -f`,
`compare_all.py -P` and `obstacle_ipopt_galahad.py -P` read.  The reference's order-2 geometry is not
generated - its reader side here reduces order-2 meshes to their vertices anyway (io.read_mesh).

    python generate_mesh_gmsh.py        ->  meshes/disk_0.xdmf ... meshes/disk_3.xdmf
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from proximalgalerkin_amd import fem, io  # noqa: E402

__all__ = ["generate_disk"]


def generate_disk(filename: Path, res: float, order: int = 1, refinement_level: int = 1):
    """A disk around the origin with radius 1 and resolution `res`, refined `refinement_level` times; written to
    `<stem>_<refinement_level>.xdmf` next to `filename` (the reference's naming).  Returns the path."""
    if order not in (1, 2):
        raise ValueError("order 1 or 2")
    mesh = fem.create_disk(res / 2**refinement_level)
    filename = Path(filename)
    out_name = filename.with_name(f"{filename.stem}_{refinement_level}").with_suffix(".xdmf")
    io.write_xdmf_mesh(out_name, mesh, encoding="HDF5")  # XDMFFile's default encoding (generate_mesh_gmsh.py:41-43)
    return out_name


if __name__ == "__main__":
    for i in range(4):
        out = generate_disk(Path("meshes/disk.xdmf"), res=0.1, order=2, refinement_level=i)
        print(out)
