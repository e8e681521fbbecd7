// Vertex-star (patch) smoother for the P2 Newton matrices of example 01 - gfx950.
//
// Reference: the reference solves the P2 systems of `obstacle_pg.py -p 2` exactly (/root/reference/examples/01_obstacle_problem/
// obstacle_pg.py:68-70,129-131,288).  Here they are solved by FGMRES with a two-level cycle: a smoother on the P2 level + the P1
// hierarchy as coarse space (pgx_api.hip: pcycle_p2).  The collective point-Jacobi smoother of rounds 1-2 is not robust there: on
// the late proximal steps e^psi spans tens of orders of magnitude inside one element and the Krylov counts grow with N (30-90 at
// 64^2-128^2, hundreds at 256^2).  The remedy (prototype: oracle/p2_patch_proto.py, 8-18 iterations at 16^2 ... 256^2 on the same
// systems) is an ADDITIVE VERTEX-STAR SCHWARZ smoother: for every mesh vertex the dofs (u, psi) on the vertex and on the edges that
// meet in it - 2 (1 + deg) <= 16 unknowns - are solved for EXACTLY, the corrections of overlapping patches are averaged
// (an edge dof belongs to the patches of its two end vertices).
//
// Data (per handle, HBM): pdof [np][NN] patch dofs (-1 = unused slot), ppos [np][NN][NN] positions of the patch's scalar-block
// entries in the P2 block-CSR (-1 = structurally zero), pinv (P = 2 NN) the inverses of the patch matrices in 4-vectors - symmetric packing
// [np][chunk q][rows 4q..P-1][4] by default, the full [np][P/4][P][4] behind PGX_P2_PATCH_SYM=0 and for the double form -,
// rebuilt once per Newton step (only D(psi) and alpha change); computed in double, STORED in float by default (the sweep is a
// smoother inside FGMRES: identical Krylov counts, half the stream).  At 2048^2 P2: 4.2 M patches, 3.8 GB of inverses.
//
// Mapping: a group of 16 lanes owns one patch, lane l its row l (four patches per 64-wide wavefront); rows meet through
// width-16 shuffles.  The kernels are streaming kernels over pinv (HBM-bound): 512 B per patch and sweep (packed float; 896 B full
// float, 1792 B in double).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pgx_internal.h"

namespace {

constexpr int GRP = 16;  // lanes per patch

// ppos[p][i][j] = position of (row dof_i, column dof_j) in the scalar CSR pattern, or -1
__global__ void __launch_bounds__(256) k_patch_positions(int np, int NN, const int32_t* __restrict__ pdof,
                                                         const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colm,
                                                         int32_t* __restrict__ ppos) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)np * NN) return;
  const int p = (int)(t / NN), i = (int)(t % NN);
  const int32_t* d = pdof + (size_t)p * NN;
  int32_t* out = ppos + ((size_t)p * NN + i) * NN;
  for (int j = 0; j < NN; ++j) out[j] = -1;
  const int row = d[i];
  if (row < 0) return;
  for (int k = rowptr[row]; k < rowptr[row + 1]; ++k) {
    const int c = colm[k] & 0x7fffffff;
    for (int j = 0; j < NN; ++j)
      if (d[j] == c) out[j] = k;
  }
}

// Patch matrix [[aK, M], [M, -D]] restricted to the patch dofs with the Dirichlet rows / columns of u replaced by identity
// (the contract of src/lvpp/problem.py:69-77) and unused slots as identity, inverted in place by Gauss-Jordan WITHOUT pivoting:
// u rows first (aK_pp is SPD), then the psi rows whose Schur complement -D_pp - M_pp (aK_pp)^-1 M_pp is negative definite - the
// quasi-definite ordering that pgx_nd relies on as well.
// Symmetric packing (round 4): the patch matrix and hence its inverse are symmetric, so only the 4-column chunks at or left of a
// row's diagonal block are kept - chunk q holds rows 4q .. P-1: [patch][q][row - 4q][4], 32 instead of 56 4-vectors at P = 14
// (512 instead of 896 B per patch and sweep in float), 40 instead of 64 at P = 16.
__host__ __device__ constexpr int patch_sym_off(int P, int q) { return q * P - 2 * q * (q - 1); }  // in 4-vectors
__host__ __device__ constexpr int patch_sym_vecs(int P) { return patch_sym_off(P, (P + 3) / 4); }

// storage conversions of an inverse entry: double, float, or bfloat16 kept as its 16 bits (PT = unsigned short; round 4: the
// sweep is a smoother inside FGMRES - entries of 8 significant bits give the same Krylov counts as float ones and halve the stream
// that bounds k_patch_apply once more; the symmetric packing only)
typedef unsigned short pgx_bf16;
template <typename PT>
__device__ __forceinline__ PT patch_to_store(double a) {
  return (PT)a;
}
template <>
__device__ __forceinline__ pgx_bf16 patch_to_store<pgx_bf16>(double a) {
  const unsigned u = __float_as_uint((float)a);
  return (pgx_bf16)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);  // round to nearest even
}
template <typename PT>
__device__ __forceinline__ float patch_from_store(PT v) {
  return (float)v;
}
template <>
__device__ __forceinline__ float patch_from_store<pgx_bf16>(pgx_bf16 v) {
  return __uint_as_float((unsigned)v << 16);
}

template <int NN, typename PT, bool SYM>
__global__ void __launch_bounds__(256) k_patch_invert(int np, const int32_t* __restrict__ pdof, const int32_t* __restrict__ ppos,
                                                      const double* __restrict__ K, const double* __restrict__ M,
                                                      const double* __restrict__ D, const uint8_t* __restrict__ mask, double alpha,
                                                      PT* __restrict__ pinv) {
  constexpr int P = 2 * NN, PQ = (P + 3) / 4;
  const int p = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / GRP);
  const int l = threadIdx.x & (GRP - 1);
  const bool live = p < np;
  const int pp = live ? p : np - 1;  // idle groups repeat the last patch (their shuffles must still run) and do not store
  const int32_t* d = pdof + (size_t)pp * NN;
  int dof[NN];
  bool bc[NN];
#pragma unroll
  for (int j = 0; j < NN; ++j) {
    dof[j] = d[j];
    bc[j] = dof[j] >= 0 && mask[dof[j]];
  }
  double a[P];
#pragma unroll
  for (int j = 0; j < P; ++j) a[j] = 0.0;
  if (l < P) {
    const int i = l < NN ? l : l - NN;
    const bool urow = l < NN;
    if (dof[i] < 0 || (urow && bc[i])) {
      a[l] = 1.0;  // identity row (unused slot / Dirichlet dof of u)
    } else {
      const int32_t* pos = ppos + ((size_t)pp * NN + i) * NN;
#pragma unroll
      for (int j = 0; j < NN; ++j) {
        const int k = pos[j];
        if (k < 0 || dof[j] < 0) continue;
        if (urow) {
          a[j] = bc[j] ? 0.0 : alpha * K[k];
          a[NN + j] = M[k];
        } else {
          a[j] = bc[j] ? 0.0 : M[k];
          a[NN + j] = -D[k];
        }
      }
    }
  } else {
    a[0] = 0.0;  // lanes P..15 hold no row
  }
  // in-place Gauss-Jordan: after step k column k holds the k-th column of the inverse of the leading block
#pragma unroll
  for (int k = 0; k < P; ++k) {
    double pr[P];
#pragma unroll
    for (int j = 0; j < P; ++j) pr[j] = __shfl(a[j], k, GRP);  // pivot row to every lane of the group
    // reciprocal of the pivot: v_rcp_f64 + two Newton steps (<= 1 ulp from the IEEE quotient; the result is stored as float) instead of an
    // fp64 division - ~200 cycles per pivot and lane, 14 of them per patch (round 5)
    double piv = __builtin_amdgcn_rcp(pr[k]);
    piv = fma(fma(-pr[k], piv, 1.0), piv, piv);
    piv = fma(fma(-pr[k], piv, 1.0), piv, piv);
    if (l == k) {
#pragma unroll
      for (int j = 0; j < P; ++j) a[j] = (j == k) ? piv : pr[j] * piv;
    } else if (l < P) {
      const double f = a[k];
#pragma unroll
      for (int j = 0; j < P; ++j) a[j] = (j == k) ? -f * piv : a[j] - f * (pr[j] * piv);
    }
  }
  if (live && l < P) {
    // layout [patch][chunk of 4 columns][row l][4]: a lane stores (and k_patch_apply loads) its row as PQ 4-vectors, and the lanes of
    // a group touch 4 P consecutive values per instruction
    typedef PT v4 __attribute__((ext_vector_type(4)));
    PT* out = pinv + (SYM ? (size_t)p * patch_sym_vecs(P) * 4 : (size_t)p * PQ * P * 4 + (size_t)l * 4);
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
      PT v[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = patch_to_store<PT>((4 * q + t < P) ? a[4 * q + t] : 0.0);
      if (!SYM)
        *(v4*)(out + (size_t)q * P * 4) = (v4){v[0], v[1], v[2], v[3]};
      else if (4 * q <= l)
        *(v4*)(out + (size_t)(patch_sym_off(P, q) + l - 4 * q) * 4) = (v4){v[0], v[1], v[2], v[3]};
    }
  }
}

// One additive sweep: y_p = A_p^-1 r_p for every patch; the vertex dof of a patch belongs to it alone (x += omega y), an edge dof
// to the patches of its two end vertices: their contributions are parked in stash[2 e + side] and averaged by k_patch_edges
// (no atomics: bitwise reproducible).
template <int NN, typename PT, bool SYM>
__global__ void __launch_bounds__(256) k_patch_apply(int np, int nv, int nd, const int32_t* __restrict__ pdof,
                                                     const int32_t* __restrict__ edge_ends, const PT* __restrict__ pinv,
                                                     const double* __restrict__ ru, const double* __restrict__ rp, double omega,
                                                     double* __restrict__ xu, double* __restrict__ xp, float* __restrict__ su,
                                                     float* __restrict__ sp) {
  constexpr int P = 2 * NN, PQ = (P + 3) / 4;
  const int p = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / GRP);
  const int l = threadIdx.x & (GRP - 1);
  const bool live = p < np;
  const int pp = live ? p : np - 1;
  const int i = l < NN ? l : l - NN;
  int dof = -1;
  double r = 0.0;
  double row[P];
  if (SYM) {
    // packed inverse: a lane loads the chunks at or left of its diagonal block (1 .. 4 16-byte loads), the group rebuilds the full
    // matrix in LDS - every loaded entry lands at (l, j), those left of the diagonal block at (j, l) as well - and each lane reads
    // its row back as four 16-byte LDS loads.  (The float form only: the LDS image is float.)
    constexpr int LD = 4 * PQ + 4;  // 80-byte rows: 16-byte aligned, the 16 rows of a patch on distinct bank groups
    __shared__ float sA[256 / GRP][P][LD];
    const int g = threadIdx.x / GRP;
    typedef PT v4 __attribute__((ext_vector_type(4)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (l < P) {
      dof = pdof[(size_t)pp * NN + i];
      if (dof >= 0) r = (l < NN) ? ru[dof] : rp[dof];
      const PT* in = pinv + (size_t)pp * patch_sym_vecs(P) * 4;
      v4 v[PQ];
#pragma unroll
      for (int q = 0; q < PQ; ++q)
        if (4 * q <= l) v[q] = __builtin_nontemporal_load((const v4*)(in + (size_t)(patch_sym_off(P, q) + l - 4 * q) * 4));
#pragma unroll
      for (int q = 0; q < PQ; ++q)
        if (4 * q <= l) {
          *(f4*)&sA[g][l][4 * q] = (f4){patch_from_store<PT>(v[q][0]), patch_from_store<PT>(v[q][1]), patch_from_store<PT>(v[q][2]),
                                        patch_from_store<PT>(v[q][3])};
          if (4 * q + 4 <= (l & ~3)) {  // strictly left of the diagonal block: the mirrored entries
#pragma unroll
            for (int t = 0; t < 4; ++t) sA[g][4 * q + t][l] = patch_from_store<PT>(v[q][t]);
          }
        }
    }
    __syncthreads();
    if (l < P) {
#pragma unroll
      for (int q = 0; q < PQ; ++q) {
        const f4 w = *(const f4*)&sA[g][l][4 * q];
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (4 * q + t < P) row[4 * q + t] = (double)w[t];
      }
    } else {
#pragma unroll
      for (int j = 0; j < P; ++j) row[j] = 0.0;
    }
  } else if (l < P) {
    dof = pdof[(size_t)pp * NN + i];
    if (dof >= 0) r = (l < NN) ? ru[dof] : rp[dof];
    // Row l of the inverse as PQ 4-vectors (layout of k_patch_invert): four 16-byte loads per lane, the lanes of a group side by side.
    // (History: row-major rows read row-wise 2.9 ms per sweep at 2048^2 P2 in double; read column-wise through the symmetry of the
    // patch matrix 1.6 ms; stored in float 1.07 ms; this layout 0.93 ms.)
    typedef PT v4 __attribute__((ext_vector_type(4)));
    const PT* in = pinv + (size_t)pp * PQ * P * 4 + (size_t)l * 4;
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
      const v4 v = __builtin_nontemporal_load((const v4*)(in + (size_t)q * P * 4));  // streamed once per sweep
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (4 * q + t < P) row[4 * q + t] = sizeof(PT) == 2 ? (double)patch_from_store<PT>(v[t]) : (double)v[t];
    }
  } else {
#pragma unroll
    for (int j = 0; j < P; ++j) row[j] = 0.0;
  }
  double y0 = 0.0, y1 = 0.0;
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const double rj = __shfl(r, j, GRP);
    if (j & 1)
      y1 = fma(row[j], rj, y1);
    else
      y0 = fma(row[j], rj, y0);
  }
  const double y = y0 + y1;
  if (!live || l >= P || dof < 0) return;
  const int v = pp;  // patch p = vertex p
  if (i == 0) {
    if (l < NN)
      xu[v] += omega * y;
    else
      xp[v] += omega * y;
  } else {
    const int e = dof - nv;
    const int side = (edge_ends[2 * e] == v) ? 0 : 1;
    (l < NN ? su : sp)[2 * (size_t)e + side] = (float)y;  // the stash is float (round 4): corrections of a float-accurate smoother
  }
}

__global__ void __launch_bounds__(256) k_patch_edges(int ne, int nv, double omega, const float* __restrict__ su,
                                                     const float* __restrict__ sp, double* __restrict__ xu, double* __restrict__ xp) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  const float2 a = ((const float2*)su)[e], b = ((const float2*)sp)[e];
  xu[nv + e] += omega * 0.5 * ((double)a.x + (double)a.y);
  xp[nv + e] += omega * 0.5 * ((double)b.x + (double)b.y);
}

}  // namespace

void pgxk_patch_positions(hipStream_t st, int np, int NN, const int32_t* pdof, const int32_t* rowptr, const int32_t* colm,
                          int32_t* ppos) {
  const int64_t t = (int64_t)np * NN;
  hipLaunchKernelGGL(k_patch_positions, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, st, np, NN, pdof, rowptr, colm, ppos);
}

size_t pgxk_patch_inverse_bytes(int np, int NN, int f32, int sym) {
  const int P = 2 * NN <= 14 ? 14 : 16;
  const size_t vecs = sym ? (size_t)patch_sym_vecs(P) : (size_t)P * ((P + 3) / 4);
  return (size_t)np * vecs * 4 * (f32 == 2 ? sizeof(pgx_bf16) : f32 ? sizeof(float) : sizeof(double));
}

void pgxk_patch_invert(hipStream_t st, int np, int NN, const int32_t* pdof, const int32_t* ppos, const double* K, const double* M,
                       const double* D, const uint8_t* mask, double alpha, void* pinv, int f32, int sym) {
  const unsigned blocks = (unsigned)(((int64_t)np * GRP + 255) / 256);
#define PGX_INV(N, T, S) \
  hipLaunchKernelGGL((k_patch_invert<N, T, S>), dim3(blocks), dim3(256), 0, st, np, pdof, ppos, K, M, D, mask, alpha, (T*)pinv)
  if (NN <= 7) {
    if (f32 == 2 && sym) PGX_INV(7, pgx_bf16, true);
    else if (f32 && sym) PGX_INV(7, float, true); else if (f32) PGX_INV(7, float, false); else PGX_INV(7, double, false);
  } else {
    if (f32 == 2 && sym) PGX_INV(8, pgx_bf16, true);
    else if (f32 && sym) PGX_INV(8, float, true); else if (f32) PGX_INV(8, float, false); else PGX_INV(8, double, false);
  }
#undef PGX_INV
}

void pgxk_patch_sweep(hipStream_t st, int np, int NN, int nv, int nd, const int32_t* pdof, const int32_t* edge_ends,
                      const void* pinv, int f32, int sym, const double* ru, const double* rp, double omega, double* xu, double* xp,
                      double* su, double* sp) {
  const unsigned blocks = (unsigned)(((int64_t)np * GRP + 255) / 256);
#define PGX_APP(N, T, S)                                                                                                          \
  hipLaunchKernelGGL((k_patch_apply<N, T, S>), dim3(blocks), dim3(256), 0, st, np, nv, nd, pdof, edge_ends, (const T*)pinv, ru, rp, \
                     omega, xu, xp, (float*)su, (float*)sp)
  if (NN <= 7) {
    if (f32 == 2 && sym) PGX_APP(7, pgx_bf16, true);
    else if (f32 && sym) PGX_APP(7, float, true); else if (f32) PGX_APP(7, float, false); else PGX_APP(7, double, false);
  } else {
    if (f32 == 2 && sym) PGX_APP(8, pgx_bf16, true);
    else if (f32 && sym) PGX_APP(8, float, true); else if (f32) PGX_APP(8, float, false); else PGX_APP(8, double, false);
  }
#undef PGX_APP
  const int ne = nd - nv;
  hipLaunchKernelGGL(k_patch_edges, dim3((ne + 255) / 256), dim3(256), 0, st, ne, nv, omega, (const float*)su, (const float*)sp, xu, xp);
}
