#!/usr/bin/env python3
"""Regenerate tests/golden/dolfinx_like_square.h5 with the real libhdf5 (compiles tools/make_h5_fixtures.c against /opt/conda's HDF5
1.10) and cross-check the pure-Python writer: a file written by proximalgalerkin_amd/h5.py must be readable by libhdf5's h5dump.
Build-container tool; the committed fixture is what the tests use (no HDF5 library is needed to run them)."""
import os
import pathlib
import subprocess
import sys
import tempfile

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

from proximalgalerkin_amd import h5  # noqa: E402

CONDA = pathlib.Path("/opt/conda")
with tempfile.TemporaryDirectory() as td:
    exe = pathlib.Path(td) / "mk"
    subprocess.run(["gcc", f"-I{CONDA / 'include'}", str(ROOT / "tools" / "make_h5_fixtures.c"), "-o", str(exe), f"-L{CONDA / 'lib'}", "-lhdf5",
                    f"-Wl,-rpath,{CONDA / 'lib'}"], check=True)
    subprocess.run([str(exe), str(ROOT / "tests" / "golden" / "dolfinx_like_square.h5")], check=True)
    f = h5.H5File(ROOT / "tests" / "golden" / "dolfinx_like_square.h5")
    mine = pathlib.Path(td) / "mine.h5"
    h5.write(mine, {k: f[k] for k in ("/Mesh/mesh/geometry", "/Mesh/mesh/topology", "/MeshTags/facet_tags/topology", "/MeshTags/facet_tags/Values")})
    out = subprocess.run([str(CONDA / "bin" / "h5dump"), "-d", "/MeshTags/facet_tags/Values", str(mine)], capture_output=True, text=True,
                         env=dict(os.environ, LD_LIBRARY_PATH=str(CONDA / "lib")))
    assert out.returncode == 0 and "7, 7, 7, 7, 9, 9, 9, 9, 9" in out.stdout, out.stdout + out.stderr
    print("fixture written; h5dump reads the pure-Python writer's file:", np.array_equal(h5.H5File(mine)["/Mesh/mesh/geometry"], f["/Mesh/mesh/geometry"]))
