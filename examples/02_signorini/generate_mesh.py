"""Generate the half sphere for the contact problem - counterpart of the reference's examples/02_signorini/generate_mesh.py
(`lvpp.mesh_generation.create_half_sphere(res=0.04)` + XDMFFile.write_mesh / write_meshtags).  Then:

    python examples/02_signorini/signorini.py --filename meshes/half_sphere.xdmf --contact-tag 2 --displacement-tag 1
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from proximalgalerkin_amd import io, mesh_generation  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--res", type=float, default=0.04, help="surface edge length (the reference's `res`)")
    ap.add_argument("--output", type=Path, default=Path("meshes/half_sphere.xdmf"))
    a = ap.parse_args()
    mesh, cell_marker, facet_marker = mesh_generation.create_half_sphere(res=a.res)
    io.write_xdmf_tet(a.output, mesh, facet_marker)
    print(f"{a.output}: {mesh.geometry.shape[0]} vertices, {mesh.cells.shape[0]} tetrahedra, "
          f"{len(facet_marker.find(2))} contact facets (tag 2), {len(facet_marker.find(1))} displacement facets (tag 1)")
