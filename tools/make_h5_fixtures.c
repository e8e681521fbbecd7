/* Writes tests/golden/dolfinx_like_square.h5 with the REAL libhdf5 (h5cc tools/make_h5_fixtures.c -o /tmp/mk && /tmp/mk out.h5):
 * the datasets DOLFINx's XDMFFile.write_mesh / write_meshtags create - /Mesh/mesh/geometry (float64 n x 2), /Mesh/mesh/topology
 * (int64 nc x 3), /MeshTags/facet_tags/topology (int64 m x 2), /MeshTags/facet_tags/Values (int32 m) - contiguous, unfiltered,
 * default format bounds (superblock 0, symbol-table groups).  proximalgalerkin_amd/h5.py (pure Python) is tested against this
 * file, which it did not write.  Mesh: the right-diagonal triangulation of [0,1]^2 with 5 x 4 cells; facets of the side x = 0
 * tagged 7, of y = 1 tagged 9. */
#include <hdf5.h>
#include <stdint.h>
#include <stdlib.h>

int main(int argc, char** argv) {
  const int nx = 5, ny = 4, sx = nx + 1;
  const int n = sx * (ny + 1), nc = 2 * nx * ny;
  double* g = malloc(sizeof(double) * 2 * n);
  int64_t* t = malloc(sizeof(int64_t) * 3 * nc);
  for (int j = 0; j <= ny; ++j)
    for (int i = 0; i <= nx; ++i) {
      g[2 * (j * sx + i)] = (double)i / nx;
      g[2 * (j * sx + i) + 1] = (double)j / ny;
    }
  int c = 0;
  for (int j = 0; j < ny; ++j)
    for (int i = 0; i < nx; ++i) {
      const int64_t v0 = j * sx + i;
      t[3 * c] = v0; t[3 * c + 1] = v0 + 1; t[3 * c + 2] = v0 + sx + 1; ++c;
      t[3 * c] = v0; t[3 * c + 1] = v0 + sx; t[3 * c + 2] = v0 + sx + 1; ++c;
    }
  const int m = ny + nx;
  int64_t* ft = malloc(sizeof(int64_t) * 2 * m);
  int32_t* fv = malloc(sizeof(int32_t) * m);
  int k = 0;
  for (int j = 0; j < ny; ++j, ++k) { ft[2 * k] = j * sx; ft[2 * k + 1] = (j + 1) * sx; fv[k] = 7; }
  for (int i = 0; i < nx; ++i, ++k) { ft[2 * k] = ny * sx + i; ft[2 * k + 1] = ny * sx + i + 1; fv[k] = 9; }
  hid_t f = H5Fcreate(argc > 1 ? argv[1] : "dolfinx_like_square.h5", H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  hid_t gm = H5Gcreate2(f, "/Mesh", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  hid_t gmm = H5Gcreate2(f, "/Mesh/mesh", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  hid_t gt = H5Gcreate2(f, "/MeshTags", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  hid_t gtt = H5Gcreate2(f, "/MeshTags/facet_tags", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  hsize_t d2[2] = {(hsize_t)n, 2};
  hid_t s = H5Screate_simple(2, d2, NULL);
  hid_t d = H5Dcreate2(f, "/Mesh/mesh/geometry", H5T_IEEE_F64LE, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, g); H5Dclose(d); H5Sclose(s);
  hsize_t d3[2] = {(hsize_t)nc, 3};
  s = H5Screate_simple(2, d3, NULL);
  d = H5Dcreate2(f, "/Mesh/mesh/topology", H5T_STD_I64LE, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  H5Dwrite(d, H5T_NATIVE_INT64, H5S_ALL, H5S_ALL, H5P_DEFAULT, t); H5Dclose(d); H5Sclose(s);
  hsize_t d4[2] = {(hsize_t)m, 2};
  s = H5Screate_simple(2, d4, NULL);
  d = H5Dcreate2(f, "/MeshTags/facet_tags/topology", H5T_STD_I64LE, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  H5Dwrite(d, H5T_NATIVE_INT64, H5S_ALL, H5S_ALL, H5P_DEFAULT, ft); H5Dclose(d); H5Sclose(s);
  hsize_t d5[1] = {(hsize_t)m};
  s = H5Screate_simple(1, d5, NULL);
  d = H5Dcreate2(f, "/MeshTags/facet_tags/Values", H5T_STD_I32LE, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  H5Dwrite(d, H5T_NATIVE_INT32, H5S_ALL, H5S_ALL, H5P_DEFAULT, fv); H5Dclose(d); H5Sclose(s);
  H5Gclose(gtt); H5Gclose(gt); H5Gclose(gmm); H5Gclose(gm); H5Fclose(f);
  return 0;
}
