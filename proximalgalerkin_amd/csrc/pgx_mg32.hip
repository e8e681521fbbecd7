// Single-precision V-cycle legs for gfx950 (round 4).  Same algebra and the same row mapping as the fp64 kernels k_st_smoothR /
// k_st_resid_restrict_r of pgx_kernels.hip - a workgroup of 8 waves owns a 64-wide image, a wave owns image rows, interior tiles
// run without tests on scalar K / M stencils, boundary tiles (scheduled first) through the general per-point code - but on HALF
// the bytes: what the level-0 launches of the fp64 cycle wait for is the unique bytes of their coefficient and vector streams
// (cycle stamps + timing experiments), not instructions, LDS or occupancy.  Here
//   * D(psi) is ONE float4 per vertex (centre + the three forward links): one 16-byte load instead of four 8-byte loads from four
//     arrays; the three mirrored links still come from the neighbours (DPP lane shift, LDS row hand-over);
//   * the cycle's vectors are interleaved (u, psi) float2: one 8-byte load / store per vertex and vector;
//   * arithmetic in float: 20 registers of coefficients per image row instead of 36, LDS images of 8 B per vertex.
// 40 B per vertex and smoother launch instead of 84.  The V-cycle is a preconditioner inside FGMRES, which is flexible; the operator
// apply, the true residual that decides convergence and the Krylov space stay fp64, so the accuracy of the Newton steps is
// untouched (tools/mg32_study.py: identical Krylov counts with a float32 cycle on the numpy twin).
// The fp64 ends of the cycle: the first launch on the finest level reads the Krylov vector (fp64) and leaves its float2 copy for
// the launches that follow; the last launch writes the preconditioned vector as fp64; a level below the last single-precision
// one receives its right-hand side, and returns its correction, in fp64.
#include <algorithm>

#include "pgx_internal.h"
#include "pgx_stencil.h"

#define F32_BLOCK 512
// A/B switches of the interior tiles, both measured slower at 2049^2 and off: F32_WAVES_EU = 8 (a fourth workgroup per CU by a 64-VGPR
// cap: 12-28 B of scratch per lane, level-0 launches +20 %) and F32_LEAN (the 2x2 block inverse recomputed per sweep: +1 %).
// 32-row tiles (halo 1.31 instead of 1.52, but 94 VGPRs = two workgroups per CU) were +15 % and are not instantiated.
#ifndef F32_LEAN
#define F32_LEAN 0
#endif
#ifndef F32_WAVES_EU
#define F32_WAVES_EU 2
#endif
#define F32_RR_CYB 2  // coarse rows of a boundary sub-tile of the residual + restriction
#define F32_TB 2  // rows of a boundary sub-tile of the smoother: image of 2 + 2K <= 8 rows, one per wave

__device__ __forceinline__ float lane_shr1f(float x) {  // the value of lane - 1 (lane 0 keeps its own)
  int v = __float_as_int(x);
  v = __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);  // wave_shr:1
  return __int_as_float(v);
}

__global__ void __launch_bounds__(256) k_f_pack_d(int n, const dsten_t* __restrict__ Dh, float4* __restrict__ Dq) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const float big = 1e30f;
  Dq[v] = make_float4(fminf((float)Dh[v], big), fminf((float)Dh[(size_t)n + v], big), fminf((float)Dh[2 * (size_t)n + v], big),
                      fminf((float)Dh[3 * (size_t)n + v], big));
}
void pgxk_f_pack_d(hipStream_t st, const GridLevel& L) {
  hipLaunchKernelGGL(k_f_pack_d, dim3((L.n + 255) / 256), dim3(256), 0, st, L.n, L.Dh, L.Dq);
}

// ------------------------------------------------------------------------------------------------
// general per-point pieces (boundary tiles)
// ------------------------------------------------------------------------------------------------
struct FCoef {
  float kv[7], mv[7], dv[7];  // kv already holds alpha K
  int rowbc;
};
// uniform interior stencils as the kernels use them, converted once on the host (scalar registers: 22 instead of the 30 + conversions
// of the fp64 StConst): alpha K and M per slot (boundary tiles) and with the symmetric link pairs averaged (interior tiles)
struct FConst {
  float kc[7], mc[7];
  float k0, k1, k3, k5, m0, m1, m3, m5;
  int uniform;
};
static inline FConst make_fconst(const GridLevel& L, double alpha) {
  FConst c;
  for (int s = 0; s < 7; ++s) {
    c.kc[s] = (float)alpha * (float)L.Kc[s];
    c.mc[s] = (float)L.Mc[s];
  }
  const float a = (float)alpha;
  c.k0 = a * (float)L.Kc[0];
  c.k1 = a * (float)(0.5 * (L.Kc[1] + L.Kc[2]));
  c.k3 = a * (float)(0.5 * (L.Kc[3] + L.Kc[4]));
  c.k5 = a * (float)(0.5 * (L.Kc[5] + L.Kc[6]));
  c.m0 = (float)L.Mc[0];
  c.m1 = (float)(0.5 * (L.Mc[1] + L.Mc[2]));
  c.m3 = (float)(0.5 * (L.Mc[3] + L.Mc[4]));
  c.m5 = (float)(0.5 * (L.Mc[5] + L.Mc[6]));
  c.uniform = L.uniform;
  return c;
}
// Stencil slots: 0:(0,0) 1:(+1,0) 2:(-1,0) 3:(0,+1) 4:(0,-1) 5:(+1,+1) 6:(-1,-1).  Links that leave the grid hold 0 in K, M and Dq.
__device__ __forceinline__ void f_load_coef(int v, int i, int j, int nx, int ny, int n, const double* __restrict__ K,
                                            const double* __restrict__ M, const float4* __restrict__ Dq, const FConst& sc,
                                            const uint8_t* __restrict__ mask, float alpha, FCoef& c) {
  const int sx = nx + 1;
  if (sc.uniform && i > 0 && i < nx && j > 0 && j < ny) {
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      c.kv[s] = sc.kc[s];
      c.mv[s] = sc.mc[s];
    }
  } else {
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      c.kv[s] = alpha * (float)K[(size_t)s * n + v];
      c.mv[s] = (float)M[(size_t)s * n + v];
    }
  }
  const float4 q = Dq[v];
  c.dv[0] = q.x;
  c.dv[1] = q.y;
  c.dv[3] = q.z;
  c.dv[5] = q.w;
  c.dv[2] = (i > 0) ? Dq[v - 1].y : 0.f;
  c.dv[4] = (j > 0) ? Dq[v - sx].z : 0.f;
  c.dv[6] = (i > 0 && j > 0) ? Dq[v - sx - 1].w : 0.f;
  c.rowbc = mask[v];
}

__device__ __forceinline__ void f_jacobi(const FCoef& c, float omega, float au, float ap, float xur, float xpr, float buv, float bpv,
                                         float& yu, float& yp) {
  if (c.rowbc) au = xur;
  const float su = buv - au, sp = bpv - ap;
  float a = c.kv[0], b = c.mv[0];
  const float dd = c.dv[0];
  float om_u = omega;
  if (c.rowbc) {  // Dirichlet row of u: solved exactly
    a = 1.f;
    b = 0.f;
    om_u = 1.f;
  }
  const float det = -a * dd - b * b;
  float du = 0.f, dpsi = 0.f;
  if (det != 0.f) {
    const float r = __builtin_amdgcn_rcpf(det);
    du = (-dd * su - b * sp) * r;
    dpsi = (-b * su + a * sp) * r;
  } else if (c.rowbc) {
    du = su;
  }
  yu = xur + om_u * du;
  yp = xpr + omega * dpsi;
}

// x + P x_c at fine vertex (gi, gj).  CADD: 0 = no correction, 1 = the coarse correction is float2 (cf), 2 = a pair of fp64 arrays
template <int CADD>
__device__ __forceinline__ float2 f_add_coarse(float2 x, int gi, int gj, int nxc, const float2* __restrict__ cf,
                                               const double* __restrict__ cdu, const double* __restrict__ cdp) {
  if (CADD == 0) return x;
  const int sxc = nxc + 1;
  const int jc = gj >> 1, ic = gi >> 1;
  const unsigned c0 = (unsigned)(jc * sxc + ic), c1 = (unsigned)((jc + (gj & 1)) * sxc + ic + (gi & 1));
  if (CADD == 1) {
    const float2 a = cf[c0], b = cf[c1];
    x.x += 0.5f * (a.x + b.x);
    x.y += 0.5f * (a.y + b.y);
  } else {
    x.x += 0.5f * (float)(cdu[c0] + cdu[c1]);
    x.y += 0.5f * (float)(cdp[c0] + cdp[c1]);
  }
  return x;
}

// ------------------------------------------------------------------------------------------------
// K sweeps per launch.  IO: 0 = float2 in and out; 1 = right-hand side from fp64 arrays, copied to bfo (FIRST launches only);
// 2 = result to fp64 arrays (launches with an iterate only).  CADD: the coarse correction added to the iterate (f_add_coarse).
// RR != 0: the launch ALSO restricts the residual of its result, b_c = P^T (b - J S^K(x)), to the coarse level (1: float2 cbf,
// 2: fp64 arrays) - the last pre-smoothing launch and the residual + restriction launch of a level in one: the image carries
// K + 2 halo rows / columns instead of K (K sweeps, one residual, one restriction stencil), the result is kept in LDS, the rows'
// coefficients are still in registers.  Saves a launch per level and cycle (what the small levels are made of) and one pass
// over D, b and x; costs (64 - 2K)(TY) / ((60 - 2K)(TY)) ... more halo work, so the finest level keeps the separate launches.
// ------------------------------------------------------------------------------------------------
struct FSmoothArgs {
  int nx, ny, n, nbnd, nxc, nyc, remap;
  RowmapGrid g;
  const double *K, *M;  // boundary rows only
  const float4* Dq;
  const uint8_t *mask, *mask_c;
  const float2 *xf, *cf, *bf;
  const double *cdu, *cdp, *b64u, *b64p;
  float2 *bfo, *yf, *cbf;
  double *y64u, *y64p, *cb64u, *cb64p;
  float alpha, omega;
  double bscale;  // IO == 1: the fp64 right-hand side is multiplied by this as it is read (lazily normalised Krylov vectors)
  FConst sc;
};

template <int TY, int K, bool FIRST, int IO, int CADD, int RR>
__device__ __forceinline__ void f_smooth_fast(int b, const FSmoothArgs& A, float2* img0, float2* img1, float2* exch) {
  constexpr int W = 64, HALO = K + (RR ? 2 : 0), TX = W - 2 * HALO, H0 = TY + 2 * HALO, NW = F32_BLOCK / 64, R = (H0 + NW - 1) / NW;
  const int sx = A.nx + 1;
  const int tx = 1 + b % A.g.nfx, ty = 1 + b / A.g.nfx;
  const int i0 = tx * TX - HALO, j0 = ty * TY - HALO;  // origin of the image
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gi = i0 + lane;
  const bool own_lane = lane >= HALO && lane < W - HALO;
  const float k0 = A.sc.k0, k1 = A.sc.k1, k3 = A.sc.k3, k5 = A.sc.k5, m0 = A.sc.m0, m1 = A.sc.m1, m3 = A.sc.m3, m5 = A.sc.m5;
  // A wave owns the SAME image rows in every sweep (lj = wave + NW k): their D links and right-hand side are loaded once, all
  // loads in flight together, and stay in registers for the K sweeps.
  float4 dq[R];
  float2 rb[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int lj = wave + NW * k;
    if (lj < H0 - 1) {  // rows 1 .. H0-2 are updated; row 0 only hands its upward links to row 1
      const unsigned v = (unsigned)((j0 + lj) * sx + gi);
      dq[k] = A.Dq[v];
      if (lj >= 1) rb[k] = (IO == 1) ? make_float2((float)(A.b64u[v] * A.bscale), (float)(A.b64p[v] * A.bscale)) : A.bf[v];
    }
  }
  if (!FIRST) {
    float2 xa[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int lj = wave + NW * k;
      if (lj < H0) {
        const int gj = j0 + lj;
        xa[k] = f_add_coarse<CADD>(A.xf[(unsigned)(gj * sx + gi)], gi, gj, A.nxc, A.cf, A.cdu, A.cdp);
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int lj = wave + NW * k;
      if (lj < H0) img0[lj * W + lane] = xa[k];
    }
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int lj = wave + NW * k;
    if (lj < H0 - 1) exch[lj * W + lane] = make_float2(dq[k].z, dq[k].w);
  }
  __syncthreads();
  // omega * inverse of the vertex block [[aK0, M0], [M0, -D0]] (det < 0: k0 > 0, d0 >= 0, m0 > 0): kept per row for the K sweeps,
  // or (LEAN: images of three rows per wave) recomputed in every sweep - nine registers less, which is what lets a fourth
  // workgroup live on the CU (64 VGPRs)
  constexpr bool LEAN = F32_LEAN && R >= 3 && !RR;
  const float nm2 = -m0 * m0, mo = -m0 * A.omega, ko = k0 * A.omega;
  float d2[R], d4[R], d6[R], g0[LEAN ? 1 : R], g1[LEAN ? 1 : R], g3[LEAN ? 1 : R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int lj = wave + NW * k;
    if (lj >= 1 && lj < H0 - 1) {
      d2[k] = lane_shr1f(dq[k].y);                   // D(+1,0) of the left neighbour
      d4[k] = exch[(lj - 1) * W + lane].x;           // D(0,+1) of the vertex below
      d6[k] = exch[(lj - 1) * W + lane - 1].y;       // D(+1,+1) of the vertex below left; lane 0 (halo, never updated) reads the guard band
      if (!LEAN) {
        const float rc = __builtin_amdgcn_rcpf(fmaf(-k0, dq[k].x, nm2));
        g0[k] = -dq[k].x * A.omega * rc;
        g1[k] = mo * rc;
        g3[k] = ko * rc;
      }
      if (IO == 1 && lj >= HALO && lj < H0 - HALO && own_lane) A.bfo[(unsigned)((j0 + lj) * sx + gi)] = rb[k];
    }
  }
  // (au, ap) = the two rows of J at image vertex q of row slot k
#define F_ROWS(src, q, k, au, ap, x0)                                                                                              \
  const float2 x0 = src[q], x1_ = src[q + 1], x2_ = src[q - 1], x3_ = src[q + W], x4_ = src[q - W], x5_ = src[q + W + 1],          \
               x6_ = src[q - W - 1];                                                                                               \
  const float u12_ = x1_.x + x2_.x, u34_ = x3_.x + x4_.x, u56_ = x5_.x + x6_.x;                                                    \
  const float p12_ = x1_.y + x2_.y, p34_ = x3_.y + x4_.y, p56_ = x5_.y + x6_.y;                                                    \
  au = (k0 * x0.x + k1 * u12_) + (k3 * u34_ + k5 * u56_) + ((m0 * x0.y + m1 * p12_) + (m3 * p34_ + m5 * p56_));                    \
  ap = ((m0 * x0.x + m1 * u12_) + (m3 * u34_ + m5 * u56_)) -                                                                       \
       (((dq[k].x * x0.y + dq[k].y * x1_.y) + (d2[k] * x2_.y + dq[k].z * x3_.y)) + ((d4[k] * x4_.y + dq[k].w * x5_.y) + d6[k] * x6_.y));
#pragma unroll
  for (int s = 1; s <= K; ++s) {
    const float2* const src = ((s - 1) & 1) ? img1 : img0;
    float2* const dst = (s & 1) ? img1 : img0;
    const bool act = lane >= s && lane < W - s;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int lj = wave + NW * k;
      if (lj < s || lj >= H0 - s) continue;  // wave-uniform
      float au = 0.f, ap = 0.f, xur = 0.f, xpr = 0.f;
      if (!FIRST || s > 1) {
        F_ROWS(src, lj * W + lane, k, au, ap, x0)
        xur = x0.x;
        xpr = x0.y;
      }
      const float su = rb[k].x - au, sp = rb[k].y - ap;
      float ou, op;
      if (LEAN) {
        const float rc = __builtin_amdgcn_rcpf(fmaf(-k0, dq[k].x, nm2));
        ou = xur + fmaf(-dq[k].x * A.omega, su, mo * sp) * rc;
        op = xpr + fmaf(mo, su, ko * sp) * rc;
      } else {
        ou = xur + fmaf(g0[k], su, g1[k] * sp);
        op = xpr + fmaf(g1[k], su, g3[k] * sp);
      }
      if (act) {
        if (s == K && lj >= HALO && lj < H0 - HALO && own_lane) {
          const unsigned v = (unsigned)((j0 + lj) * sx + gi);
          if (IO == 2) {
            A.y64u[v] = (double)ou;
            A.y64p[v] = (double)op;
          } else {
            A.yf[v] = make_float2(ou, op);
          }
        }
        if (s < K || RR) dst[lj * W + lane] = make_float2(ou, op);
      }
    }
    if (s < K || RR) __syncthreads();
  }
  if (RR) {
    const float2* const xk = (K & 1) ? img1 : img0;  // S^K(x) on rows / lanes [K, . - K)
    float2* const rim = (K & 1) ? img0 : img1;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int lj = wave + NW * k;
      if (lj < K + 1 || lj >= H0 - K - 1) continue;  // wave-uniform
      float au, ap;
      F_ROWS(xk, lj * W + lane, k, au, ap, x0)
      if (lane >= K + 1 && lane < W - K - 1) rim[lj * W + lane] = make_float2(rb[k].x - au, rb[k].y - ap);
    }
    __syncthreads();
    // coarse vertex (I0 + ci, J0 + cj) = fine vertex at image row HALO + 2 cj, lane HALO + 2 ci
    const int sxc = A.nxc + 1, I0 = (tx * TX) >> 1, J0 = (ty * TY) >> 1;
    for (int cj = wave; cj < TY / 2; cj += NW)
      if (lane < TX / 2) {
        const int q = (HALO + 2 * cj) * W + HALO + 2 * lane;
        const float2 r0 = rim[q], r1 = rim[q + 1], r2 = rim[q - 1], r3 = rim[q + W], r4 = rim[q - W], r5 = rim[q + W + 1],
                     r6 = rim[q - W - 1];
        const float su = r0.x + 0.5f * (((r1.x + r2.x) + (r3.x + r4.x)) + (r5.x + r6.x));
        const float sp = r0.y + 0.5f * (((r1.y + r2.y) + (r3.y + r4.y)) + (r5.y + r6.y));
        const int C = (J0 + cj) * sxc + I0 + lane;
        if (RR == 2) {
          A.cb64u[C] = (double)su;
          A.cb64p[C] = (double)sp;
        } else {
          A.cbf[C] = make_float2(su, sp);
        }
      }
  }
#undef F_ROWS
}

// Boundary tiles (tile index as in k_st_smoothR: row ty = 0 | rows 1..nfy: columns 0 and nfx+1.. | rows nfy+1..), cut into
// sub-tiles of TB rows.  Round 3's boundary path evaluated every vertex from scratch in every sweep - coefficient loads behind
// per-lane tests, a chain of dependent memory round trips of ~1 us per row iteration, and on the small levels that chain IS the
// launch time.  Here the boundary tiles are built like the interior ones: a wave owns image rows, EVERYTHING a row needs - its D
// links, right-hand side, iterate, Dirichlet flag and, for the vertices of the grid's frame only, the K / M rows from the arrays
// (every other vertex takes the uniform constants) - is loaded once, all loads in flight together, and stays in registers for the
// K sweeps; the mirrored D links come from the neighbours (out-of-grid entries are loaded as 0, and a link that leaves the grid is
// stored as 0, so no per-link test is left); the 2x2 vertex block is inverted once per launch.  With TB = 2 an image has
// 2 + 2K <= 8 rows: one row per wave.
template <int TY, int TB, int K, bool FIRST, int IO, int CADD, int RR>
__device__ __forceinline__ void f_smooth_bnd(int b, const FSmoothArgs& A, float2* img0, float2* img1, float2* exch) {
  constexpr int W = 64, HALO = K + (RR ? 2 : 0), TX = W - 2 * HALO, H0 = TB + 2 * HALO, NSUB = TY / TB, NW = F32_BLOCK / 64,
                R = (H0 + NW - 1) / NW;
  static_assert(TY % TB == 0 && TB % 2 == 0, "sub-tiles must cover a tile and start on even rows");
  const int nx = A.nx, ny = A.ny, sx = nx + 1;
  const RowmapGrid& g = A.g;
  int tx, ty;
  const int sub = b % NSUB;
  b /= NSUB;
  const int side = g.ntx - g.nfx;  // boundary tiles in a row that also holds fast tiles
  if (b < g.ntx) {
    tx = b;
    ty = 0;
  } else if ((b -= g.ntx) < g.nfy * side) {
    ty = 1 + b / side;
    const int r = b % side;
    tx = r == 0 ? 0 : g.nfx + r;
  } else {
    b -= g.nfy * side;
    ty = g.nfy + 1 + b / g.ntx;
    tx = b % g.ntx;
  }
  const int i0 = tx * TX - HALO, j0 = ty * TY + sub * TB - HALO;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gi = i0 + lane;
  const bool own_lane = lane >= HALO && lane < W - HALO;
  float4 dq[R];
  float2 rb[R], xa[R];
  float kv[R][7], mv[R][7];
  bool in[R], bc[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int lj = wave + NW * k, gj = j0 + lj;
    in[k] = lj < H0 && gi >= 0 && gi <= nx && gj >= 0 && gj <= ny;
    bc[k] = false;
    dq[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    rb[k] = xa[k] = make_float2(0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      kv[k][t] = A.sc.kc[t];
      mv[k][t] = A.sc.mc[t];
    }
    if (in[k]) {
      const int v = gj * sx + gi;
      dq[k] = A.Dq[v];
      bc[k] = A.mask[v] != 0;
      if (lj >= 1 && lj < H0 - 1) rb[k] = (IO == 1) ? make_float2((float)(A.b64u[v] * A.bscale), (float)(A.b64p[v] * A.bscale)) : A.bf[v];
      if (!FIRST) xa[k] = f_add_coarse<CADD>(A.xf[v], gi, gj, A.nxc, A.cf, A.cdu, A.cdp);
      if (!(A.sc.uniform && gi > 0 && gi < nx && gj > 0 && gj < ny)) {  // the grid's frame (or a level without uniform stencils)
#pragma unroll
        for (int t = 0; t < 7; ++t) {
          kv[k][t] = A.alpha * (float)A.K[(size_t)t * A.n + v];
          mv[k][t] = (float)A.M[(size_t)t * A.n + v];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int lj = wave + NW * k;
    if (lj < H0) {
      if (!FIRST) img0[lj * W + lane] = make_float2(bc[k] ? 0.f : xa[k].x, xa[k].y);  // pre-masked: Dirichlet u reads as 0
      exch[lj * W + lane] = make_float2(dq[k].z, dq[k].w);
    }
  }
  __syncthreads();
  float d2[R], d4[R], d6[R], g0[R], g1[R], g2[R], g3[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int lj = wave + NW * k;
    d2[k] = d4[k] = d6[k] = g0[k] = g1[k] = g2[k] = g3[k] = 0.f;
    if (lj >= 1 && lj < H0 - 1) {
      d2[k] = lane_shr1f(dq[k].y);
      d4[k] = exch[(lj - 1) * W + lane].x;
      d6[k] = exch[(lj - 1) * W + lane - 1].y;
      // damped inverse of the vertex block, once per launch (f_jacobi: a Dirichlet row of u is solved exactly)
      float a = kv[k][0], bm = mv[k][0], om_u = A.omega;
      if (bc[k]) {
        a = 1.f;
        bm = 0.f;
        om_u = 1.f;
      }
      const float det = -a * dq[k].x - bm * bm;
      if (det != 0.f) {
        const float r = __builtin_amdgcn_rcpf(det);
        g0[k] = om_u * (-dq[k].x * r);
        g1[k] = om_u * (-bm * r);
        g2[k] = A.omega * (-bm * r);
        g3[k] = A.omega * (a * r);
      } else if (bc[k]) {
        g0[k] = 1.f;
      }
    }
  }
  // (au, ap) = the two rows of J at image vertex q of row slot k, per-lane K / M coefficients
#define F_ROWS_G(src, q, k, au, ap, x0)                                                                                            \
  const float2 x0 = src[q], x1_ = src[q + 1], x2_ = src[q - 1], x3_ = src[q + W], x4_ = src[q - W], x5_ = src[q + W + 1],          \
               x6_ = src[q - W - 1];                                                                                               \
  au = ((kv[k][0] * x0.x + kv[k][1] * x1_.x) + (kv[k][2] * x2_.x + kv[k][3] * x3_.x)) +                                            \
       ((kv[k][4] * x4_.x + kv[k][5] * x5_.x) + kv[k][6] * x6_.x) +                                                                \
       (((mv[k][0] * x0.y + mv[k][1] * x1_.y) + (mv[k][2] * x2_.y + mv[k][3] * x3_.y)) +                                           \
        ((mv[k][4] * x4_.y + mv[k][5] * x5_.y) + mv[k][6] * x6_.y));                                                               \
  ap = (((mv[k][0] * x0.x + mv[k][1] * x1_.x) + (mv[k][2] * x2_.x + mv[k][3] * x3_.x)) +                                           \
        ((mv[k][4] * x4_.x + mv[k][5] * x5_.x) + mv[k][6] * x6_.x)) -                                                              \
       (((dq[k].x * x0.y + dq[k].y * x1_.y) + (d2[k] * x2_.y + dq[k].z * x3_.y)) + ((d4[k] * x4_.y + dq[k].w * x5_.y) + d6[k] * x6_.y));
#pragma unroll
  for (int s = 1; s <= K; ++s) {
    const float2* const src = ((s - 1) & 1) ? img1 : img0;
    float2* const dst = (s & 1) ? img1 : img0;
    const bool act = lane >= s && lane < W - s;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int lj = wave + NW * k;
      if (lj < s || lj >= H0 - s) continue;  // wave-uniform
      float au = 0.f, ap = 0.f, xur = 0.f, xpr = 0.f;
      if (!FIRST || s > 1) {
        F_ROWS_G(src, lj * W + lane, k, au, ap, x0)
        xur = x0.x;
        xpr = x0.y;
      }
      if (bc[k]) au = xur;  // (the image holds 0 there: the row reads u = b_u)
      const float su = rb[k].x - au, sp = rb[k].y - ap;
      const float ou = xur + fmaf(g0[k], su, g1[k] * sp);
      const float op = xpr + fmaf(g2[k], su, g3[k] * sp);
      if (act) {
        if (s == K && in[k] && lj >= HALO && lj < H0 - HALO && own_lane) {
          const int v = (j0 + lj) * sx + gi;
          if (IO == 1) A.bfo[v] = rb[k];
          if (IO == 2) {
            A.y64u[v] = (double)ou;
            A.y64p[v] = (double)op;
          } else {
            A.yf[v] = make_float2(ou, op);
          }
        }
        if (s < K || RR) dst[lj * W + lane] = in[k] ? make_float2(bc[k] ? 0.f : ou, op) : make_float2(0.f, 0.f);
      }
    }
    if (s < K || RR) __syncthreads();
  }
  if (RR) {
    const float2* const xk = (K & 1) ? img1 : img0;
    float2* const rim = (K & 1) ? img0 : img1;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int lj = wave + NW * k;
      if (lj < K + 1 || lj >= H0 - K - 1) continue;  // wave-uniform
      float au, ap;
      F_ROWS_G(xk, lj * W + lane, k, au, ap, x0)
      // a Dirichlet row of u holds u = b_u after any sweep: its residual is 0 (the image keeps 0 there for the neighbours)
      if (lane >= K + 1 && lane < W - K - 1)
        rim[lj * W + lane] = in[k] ? make_float2(bc[k] ? 0.f : rb[k].x - au, rb[k].y - ap) : make_float2(0.f, 0.f);
    }
    __syncthreads();
    const int sxc = A.nxc + 1, I0 = (tx * TX) >> 1, J0 = (ty * TY + sub * TB) >> 1;
    for (int cj = wave; cj < TB / 2; cj += NW)
      if (lane < TX / 2 && I0 + lane <= A.nxc && J0 + cj <= A.nyc) {
        const int q = (HALO + 2 * cj) * W + HALO + 2 * lane;
        const float2 r0 = rim[q], r1 = rim[q + 1], r2 = rim[q - 1], r3 = rim[q + W], r4 = rim[q - W], r5 = rim[q + W + 1],
                     r6 = rim[q - W - 1];
        float su = r0.x + 0.5f * (((r1.x + r2.x) + (r3.x + r4.x)) + (r5.x + r6.x));
        const float sp = r0.y + 0.5f * (((r1.y + r2.y) + (r3.y + r4.y)) + (r5.y + r6.y));
        const int C = (J0 + cj) * sxc + I0 + lane;
        if (A.mask_c[C]) su = 0.f;
        if (RR == 2) {
          A.cb64u[C] = (double)su;
          A.cb64p[C] = (double)sp;
        } else {
          A.cbf[C] = make_float2(su, sp);
        }
      }
  }
#undef F_ROWS_G
}

// ONE launch per smoother call: blocks [0, nbnd) are the boundary sub-tiles - they start first, so their long dependent-load
// chains overlap with the interior tiles that follow - blocks [nbnd, nbnd + nfast) the interior tiles.
template <int TY, int K, bool FIRST, int IO, int CADD, int RR>
__global__ void __launch_bounds__(F32_BLOCK, RR ? 2 : F32_WAVES_EU) k_f_smooth(const FSmoothArgs A) {
  constexpr int W = 64, H0 = TY + 2 * (K + (RR ? 2 : 0)), PAD = W + 1;
  __shared__ float2 img_[3][H0 * W + 2 * PAD];  // guard bands: inactive edge lanes read (and discard) one entry outside a row;
                                                 // [2]: the (D(0,+1), D(+1,+1)) links every image row hands to the row above it
  const int blk = blockIdx.x;
  if (blk < A.nbnd)
    f_smooth_bnd<TY, F32_TB, K, FIRST, IO, CADD, RR>(blk, A, img_[0] + PAD, img_[1] + PAD, img_[2] + PAD);
  else
    f_smooth_fast<TY, K, FIRST, IO, CADD, RR>(xcd_block(blk - A.nbnd, gridDim.x - A.nbnd, A.remap), A, img_[0] + PAD, img_[1] + PAD, img_[2] + PAD);
}

// last pre-smoothing launch + residual + restriction (no coarse correction, float2 result); TY >= 8: the first interior tile's
// image (K + 2 halo rows) must start inside the grid
template <int TY, int K>
static void launch_f_smooth_rr(hipStream_t st, int first, FSmoothArgs& A, int fast_ok) {
  const int rr = A.cbf ? 1 : 2;
  A.g = rowmap_grid<TY, K + 2>(A.nx, A.ny, fast_ok);
  const int nfast = A.g.nfx * A.g.nfy;
  A.nbnd = (A.g.ntx * A.g.nty - nfast) * (TY / F32_TB);
  const dim3 grid(A.nbnd + nfast), block(F32_BLOCK);
  if (first) {
    if (A.b64u) {
      if (rr == 1)
        hipLaunchKernelGGL((k_f_smooth<TY, K, true, 1, 0, 1>), grid, block, 0, st, A);
      else
        hipLaunchKernelGGL((k_f_smooth<TY, K, true, 1, 0, 2>), grid, block, 0, st, A);
    } else {
      if (rr == 1)
        hipLaunchKernelGGL((k_f_smooth<TY, K, true, 0, 0, 1>), grid, block, 0, st, A);
      else
        hipLaunchKernelGGL((k_f_smooth<TY, K, true, 0, 0, 2>), grid, block, 0, st, A);
    }
  } else {
    if (rr == 1)
      hipLaunchKernelGGL((k_f_smooth<TY, K, false, 0, 0, 1>), grid, block, 0, st, A);
    else
      hipLaunchKernelGGL((k_f_smooth<TY, K, false, 0, 0, 2>), grid, block, 0, st, A);
  }
}

template <int TY, int K>
static void launch_f_smooth(hipStream_t st, int first, FSmoothArgs& A, int fast_ok) {
  A.g = rowmap_grid<TY, K>(A.nx, A.ny, fast_ok);
  const int nfast = A.g.nfx * A.g.nfy;
  A.nbnd = (A.g.ntx * A.g.nty - nfast) * (TY / F32_TB);  // boundary tiles: sub-tiles of F32_TB rows
  const dim3 grid(A.nbnd + nfast), block(F32_BLOCK);
  if (first) {
    if (A.b64u)
      hipLaunchKernelGGL((k_f_smooth<TY, K, true, 1, 0, 0>), grid, block, 0, st, A);
    else
      hipLaunchKernelGGL((k_f_smooth<TY, K, true, 0, 0, 0>), grid, block, 0, st, A);
    return;
  }
  const int cadd = A.cf ? 1 : (A.cdu ? 2 : 0);
  if (A.y64u) {
    if (cadd == 0)
      hipLaunchKernelGGL((k_f_smooth<TY, K, false, 2, 0, 0>), grid, block, 0, st, A);
    else if (cadd == 1)
      hipLaunchKernelGGL((k_f_smooth<TY, K, false, 2, 1, 0>), grid, block, 0, st, A);
    else
      hipLaunchKernelGGL((k_f_smooth<TY, K, false, 2, 2, 0>), grid, block, 0, st, A);
  } else {
    if (cadd == 0)
      hipLaunchKernelGGL((k_f_smooth<TY, K, false, 0, 0, 0>), grid, block, 0, st, A);
    else if (cadd == 1)
      hipLaunchKernelGGL((k_f_smooth<TY, K, false, 0, 1, 0>), grid, block, 0, st, A);
    else
      hipLaunchKernelGGL((k_f_smooth<TY, K, false, 0, 2, 0>), grid, block, 0, st, A);
  }
}

static int f32_tile_rows(const GridLevel& L) {
  static PgxTuneInt t_ty("PGX_F32_TY", 0);
  const int ty = t_ty.get();
  if (ty == 4 || ty == 8 || ty == 16) return ty;
  return L.n >= 2000000 ? 16 : L.n >= 500000 ? 8 : 4;  // as k_st_smoothR (measured there per level size)
}

void pgxk_f_smooth(hipStream_t st, int K, int first, const GridLevel& L, double alpha, const float2* xf, const double* b64u,
                   const double* b64p, const GridLevel* C, const float2* cf, const double* cdu, const double* cdp, double omega,
                   int remap, float2* yf, double* y64u, double* y64p, float2* cbf, double* cb64u, double* cb64p, double bscale) {
  FSmoothArgs A;
  A.bscale = bscale;
  A.nyc = C ? C->ny : 0;
  A.mask_c = C ? C->mask : nullptr;
  A.cbf = cbf;
  A.cb64u = cb64u;
  A.cb64p = cb64p;
  A.nx = L.nx;
  A.ny = L.ny;
  A.n = L.n;
  A.nxc = C ? C->nx : 0;
  A.remap = remap;
  A.K = L.K;
  A.M = L.M;
  A.Dq = L.Dq;
  A.mask = L.mask;
  A.xf = first ? nullptr : xf;
  A.cf = first ? nullptr : cf;
  A.cdu = first ? nullptr : cdu;
  A.cdp = first ? nullptr : cdp;
  A.b64u = first ? b64u : nullptr;
  A.b64p = first ? b64p : nullptr;
  A.bf = A.b64u ? nullptr : L.bf;
  A.bfo = A.b64u ? L.bf : nullptr;
  A.yf = yf;
  A.y64u = first ? nullptr : y64u;
  A.y64p = first ? nullptr : y64p;
  A.alpha = (float)alpha;
  A.omega = (float)omega;
  A.sc = make_fconst(L, alpha);
  const int ty = f32_tile_rows(L);
  if (K == 6) {  // small levels (launch-latency bound): a whole leg of six sweeps in ONE launch
    if (cbf || cb64u)
      launch_f_smooth_rr<16, 6>(st, first, A, L.interior_free && (!C || C->interior_free));
    else
      launch_f_smooth<8, 6>(st, first, A, L.interior_free);
    return;
  }
  if (cbf || cb64u) {
    if (K == 3) {
      if (ty <= 8)
        launch_f_smooth_rr<8, 3>(st, first, A, L.interior_free && (!C || C->interior_free));
      else
        launch_f_smooth_rr<16, 3>(st, first, A, L.interior_free && (!C || C->interior_free));
    } else {
      if (ty <= 8)
        launch_f_smooth_rr<8, 2>(st, first, A, L.interior_free && (!C || C->interior_free));
      else
        launch_f_smooth_rr<16, 2>(st, first, A, L.interior_free && (!C || C->interior_free));
    }
    return;
  }
  if (K == 3) {
    if (ty == 4)
      launch_f_smooth<4, 3>(st, first, A, L.interior_free);
    else if (ty == 8)
      launch_f_smooth<8, 3>(st, first, A, L.interior_free);
    else
      launch_f_smooth<16, 3>(st, first, A, L.interior_free);
  } else {
    if (ty == 4)
      launch_f_smooth<4, 2>(st, first, A, L.interior_free);
    else if (ty == 8)
      launch_f_smooth<8, 2>(st, first, A, L.interior_free);
    else
      launch_f_smooth<16, 2>(st, first, A, L.interior_free);
  }
}

// ------------------------------------------------------------------------------------------------
// b_c = P^T (b - J x): a workgroup of 8 waves owns CX x CY = 30 x 8 coarse vertices.  (1) the iterate on the 63 x 19 fine
// footprint goes into an LDS image, a wave per row, together with the (D(0,+1), D(+1,+1)) links every row hands to the row above;
// (2) a wave per fine row evaluates the residual on 61 x 17 vertices into a second image; (3) wave w restricts coarse row w.
// ------------------------------------------------------------------------------------------------
struct FRrArgs {
  int nx, ny, n, nbnd, nxc, nyc, remap;
  RrGrid g;
  const double *K, *M;
  const float4* Dq;
  const uint8_t *mask, *mask_c;
  const float2 *xf, *bf;
  float2* cbf;
  double *cb64u, *cb64p;
  float alpha;
  FConst sc;
};

template <bool CB64>
__device__ __forceinline__ void f_rr_fast(int tx, int ty, const FRrArgs& A, float2* ximg, float2* rimg, float2* exch) {
  constexpr int W = 64, CX = 30, CY = 8, HX = 2 * CY + 3, HR = 2 * CY + 1, NW = F32_BLOCK / 64, R = (HX + NW - 1) / NW;
  const int sx = A.nx + 1, sxc = A.nxc + 1;
  const int I0 = tx * CX, J0 = ty * CY;
  const int i0 = 2 * I0 - 2, j0 = 2 * J0 - 2;  // origin of the x image; the residual image starts one row / column further in
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gi = i0 + lane;
  const bool act = lane >= 1 && lane < W - 1;
  // every global load of the wave's (up to three) rows in flight before the first LDS store
  float2 xa[R], rb[R];
  float4 dq[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int lj = wave + NW * k;
    if (lj < HX) {
      const unsigned v = (unsigned)((j0 + lj) * sx + gi);
      xa[k] = A.xf[v];
      if (lj <= HR) dq[k] = A.Dq[v];
      if (lj >= 1 && lj <= HR) rb[k] = A.bf[v];
    }
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int lj = wave + NW * k;
    if (lj < HX) ximg[lj * W + lane] = xa[k];
    if (lj <= HR) exch[lj * W + lane] = make_float2(dq[k].z, dq[k].w);
  }
  __syncthreads();
  const float k0 = A.sc.k0, k1 = A.sc.k1, k3 = A.sc.k3, k5 = A.sc.k5, m0 = A.sc.m0, m1 = A.sc.m1, m3 = A.sc.m3, m5 = A.sc.m5;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int lj = wave + NW * k;
    if (lj < 1 || lj > HR) continue;  // wave-uniform
    const float d2 = lane_shr1f(dq[k].y), d4 = exch[(lj - 1) * W + lane].x, d6 = exch[(lj - 1) * W + lane - 1].y;
    const int q = lj * W + lane;
    const float2 x0 = ximg[q], x1 = ximg[q + 1], x2 = ximg[q - 1], x3 = ximg[q + W], x4 = ximg[q - W], x5 = ximg[q + W + 1],
                 x6 = ximg[q - W - 1];
    const float u12 = x1.x + x2.x, u34 = x3.x + x4.x, u56 = x5.x + x6.x;
    const float p12 = x1.y + x2.y, p34 = x3.y + x4.y, p56 = x5.y + x6.y;
    const float au = (k0 * x0.x + k1 * u12) + (k3 * u34 + k5 * u56) + ((m0 * x0.y + m1 * p12) + (m3 * p34 + m5 * p56));
    const float ap = ((m0 * x0.x + m1 * u12) + (m3 * u34 + m5 * u56)) -
                     (((dq[k].x * x0.y + dq[k].y * x1.y) + (d2 * x2.y + dq[k].z * x3.y)) + ((d4 * x4.y + dq[k].w * x5.y) + d6 * x6.y));
    if (act) rimg[(lj - 1) * W + lane] = make_float2(rb[k].x - au, rb[k].y - ap);
  }
  __syncthreads();
  // restriction: wave w -> coarse row J0 + w, lane -> coarse column I0 + lane.  Fine vertex (2I, 2J) sits at residual-image
  // row 2w + 1 (image rows start at fine row 2 J0 - 1) and column 2 lane + 2
  if (wave < CY && lane < CX) {
    const int q = (2 * wave + 1) * W + 2 * lane + 2;
    const float2 r0 = rimg[q], r1 = rimg[q + 1], r2 = rimg[q - 1], r3 = rimg[q + W], r4 = rimg[q - W], r5 = rimg[q + W + 1],
                 r6 = rimg[q - W - 1];
    const float su = r0.x + 0.5f * (((r1.x + r2.x) + (r3.x + r4.x)) + (r5.x + r6.x));
    const float sp = r0.y + 0.5f * (((r1.y + r2.y) + (r3.y + r4.y)) + (r5.y + r6.y));
    const int C = (J0 + wave) * sxc + I0 + lane;
    if (CB64) {
      A.cb64u[C] = (double)su;
      A.cb64p[C] = (double)sp;
    } else {
      A.cbf[C] = make_float2(su, sp);
    }
  }
}

// boundary tiles, cut into sub-tiles of CYB = 2 coarse rows (x image of 7 fine rows: one per wave), register-resident like the
// boundary tiles of the smoother (f_smooth_bnd): one batch of loads, no dependent chains
template <bool CB64>
__device__ __forceinline__ void f_rr_bnd(int tx, int ty, int sub, const FRrArgs& A, float2* ximg, float2* rimg, float2* exch) {
  constexpr int W = 64, CX = 30, CY = 8, CYB = F32_RR_CYB, HX = 2 * CYB + 3, HR = 2 * CYB + 1, NW = F32_BLOCK / 64;
  static_assert(HX <= NW && CY % CYB == 0, "one image row per wave");
  const int nx = A.nx, ny = A.ny, sx = nx + 1, sxc = A.nxc + 1;
  const int I0 = tx * CX, J0 = ty * CY + sub * CYB;
  const int i0 = 2 * I0 - 2, j0 = 2 * J0 - 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gi = i0 + lane, lj = wave, gj = j0 + lj;
  const bool in = lj < HX && gi >= 0 && gi <= nx && gj >= 0 && gj <= ny;
  const bool act = lane >= 1 && lane < W - 1;
  float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
  float2 rb = make_float2(0.f, 0.f), xa = make_float2(0.f, 0.f);
  float kv[7], mv[7];
  bool bc = false;
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    kv[t] = A.sc.kc[t];
    mv[t] = A.sc.mc[t];
  }
  if (in) {
    const int v = gj * sx + gi;
    xa = A.xf[v];
    bc = A.mask[v] != 0;
    if (lj <= HR) dq = A.Dq[v];
    if (lj >= 1 && lj <= HR) {
      rb = A.bf[v];
      if (!(A.sc.uniform && gi > 0 && gi < nx && gj > 0 && gj < ny)) {
#pragma unroll
        for (int t = 0; t < 7; ++t) {
          kv[t] = A.alpha * (float)A.K[(size_t)t * A.n + v];
          mv[t] = (float)A.M[(size_t)t * A.n + v];
        }
      }
    }
  }
  if (lj < HX) {
    ximg[lj * W + lane] = make_float2(bc ? 0.f : xa.x, xa.y);  // pre-masked image
    exch[lj * W + lane] = make_float2(dq.z, dq.w);
  }
  __syncthreads();
  if (lj >= 1 && lj <= HR) {
    const float d2 = lane_shr1f(dq.y), d4 = exch[(lj - 1) * W + lane].x, d6 = exch[(lj - 1) * W + lane - 1].y;
    const int q = lj * W + lane;
    const float2 x0 = ximg[q], x1 = ximg[q + 1], x2 = ximg[q - 1], x3 = ximg[q + W], x4 = ximg[q - W], x5 = ximg[q + W + 1],
                 x6 = ximg[q - W - 1];
    float au = ((kv[0] * x0.x + kv[1] * x1.x) + (kv[2] * x2.x + kv[3] * x3.x)) + ((kv[4] * x4.x + kv[5] * x5.x) + kv[6] * x6.x) +
               (((mv[0] * x0.y + mv[1] * x1.y) + (mv[2] * x2.y + mv[3] * x3.y)) + ((mv[4] * x4.y + mv[5] * x5.y) + mv[6] * x6.y));
    const float ap = (((mv[0] * x0.x + mv[1] * x1.x) + (mv[2] * x2.x + mv[3] * x3.x)) + ((mv[4] * x4.x + mv[5] * x5.x) + mv[6] * x6.x)) -
                     (((dq.x * x0.y + dq.y * x1.y) + (d2 * x2.y + dq.z * x3.y)) + ((d4 * x4.y + dq.w * x5.y) + d6 * x6.y));
    if (bc) au = xa.x;  // Dirichlet row of u: r_u = b_u - u (the unmasked iterate)
    if (act) rimg[(lj - 1) * W + lane] = in ? make_float2(rb.x - au, rb.y - ap) : make_float2(0.f, 0.f);  // out-of-grid fine vertices hold 0
  }
  __syncthreads();
  if (wave < CYB && lane < CX) {
    const int I = I0 + lane, J = J0 + wave;
    if (I <= A.nxc && J <= A.nyc) {
      const int q = (2 * wave + 1) * W + 2 * lane + 2;
      const float2 r0 = rimg[q], r1 = rimg[q + 1], r2 = rimg[q - 1], r3 = rimg[q + W], r4 = rimg[q - W], r5 = rimg[q + W + 1],
                   r6 = rimg[q - W - 1];
      float su = r0.x + 0.5f * (((r1.x + r2.x) + (r3.x + r4.x)) + (r5.x + r6.x));
      const float sp = r0.y + 0.5f * (((r1.y + r2.y) + (r3.y + r4.y)) + (r5.y + r6.y));
      const int C = J * sxc + I;
      if (A.mask_c[C]) su = 0.f;
      if (CB64) {
        A.cb64u[C] = (double)su;
        A.cb64p[C] = (double)sp;
      } else {
        A.cbf[C] = make_float2(su, sp);
      }
    }
  }
}

template <bool CB64>
__global__ void __launch_bounds__(F32_BLOCK) k_f_resid_restrict(const FRrArgs A) {
  constexpr int W = 64, CY = 8, HX = 2 * CY + 3, HR = 2 * CY + 1, PAD = W + 1;
  __shared__ float2 ximg_[HX * W + 2 * PAD], rimg_[HR * W + 2 * PAD], exch_[HX * W + 2 * PAD];
  int b = blockIdx.x;
  if (b < A.nbnd) {
    constexpr int NSUB = 8 / F32_RR_CYB;
    const int sub = b % NSUB;
    b /= NSUB;
    int tx, ty;
    const RrGrid& g = A.g;
    const int side = g.ntx - g.nfx;
    if (b < g.ntx) {
      tx = b;
      ty = 0;
    } else if ((b -= g.ntx) < g.nfy * side) {
      ty = 1 + b / side;
      const int r = b % side;
      tx = r == 0 ? 0 : g.nfx + r;
    } else {
      b -= g.nfy * side;
      ty = g.nfy + 1 + b / g.ntx;
      tx = b % g.ntx;
    }
    f_rr_bnd<CB64>(tx, ty, sub, A, ximg_ + PAD, rimg_ + PAD, exch_ + PAD);
  } else {
    b = xcd_block(b - A.nbnd, gridDim.x - A.nbnd, A.remap);
    f_rr_fast<CB64>(1 + b % A.g.nfx, 1 + b / A.g.nfx, A, ximg_ + PAD, rimg_ + PAD, exch_ + PAD);
  }
}

void pgxk_f_resid_restrict(hipStream_t st, const GridLevel& L, double alpha, const float2* xf, const GridLevel& C, int remap,
                           float2* cbf, double* cb64u, double* cb64p) {
  constexpr int CX = 30, CY = 8;
  FRrArgs A;
  A.nx = L.nx;
  A.ny = L.ny;
  A.n = L.n;
  A.nxc = C.nx;
  A.nyc = C.ny;
  A.remap = remap;
  A.K = L.K;
  A.M = L.M;
  A.Dq = L.Dq;
  A.mask = L.mask;
  A.mask_c = C.mask;
  A.xf = xf;
  A.bf = L.bf;
  A.cbf = cbf;
  A.cb64u = cb64u;
  A.cb64p = cb64p;
  A.alpha = (float)alpha;
  A.sc = make_fconst(L, alpha);
  RrGrid& g = A.g;
  g.ntx = (C.nx + CX) / CX;
  g.nty = (C.ny + CY) / CY;
  // interior <=> every vertex of the x image is strictly inside the fine grid: 2 tx CX - 2 >= 1, 2 tx CX - 2 + 63 <= nx - 1,
  // 2 ty CY - 2 >= 1, 2 ty CY - 2 + (2 CY + 2) <= ny - 1
  g.nfx = (L.nx - 62) >= 2 * CX ? (L.nx - 62) / (2 * CX) : 0;
  g.nfy = (L.ny - 1 - 2 * CY) >= 2 * CY ? (L.ny - 1 - 2 * CY) / (2 * CY) : 0;
  g.nfx = std::min(g.nfx, g.ntx - 1);
  g.nfy = std::min(g.nfy, g.nty - 1);
  if (!L.interior_free || !C.interior_free || g.nfx <= 0 || g.nfy <= 0) g.nfx = g.nfy = 0;
  const int nfast = g.nfx * g.nfy;
  A.nbnd = (g.ntx * g.nty - nfast) * (CY / F32_RR_CYB);  // boundary tiles: sub-tiles of F32_RR_CYB coarse rows
  const dim3 grid(A.nbnd + nfast), block(F32_BLOCK);
  if (cb64u)
    hipLaunchKernelGGL(k_f_resid_restrict<true>, grid, block, 0, st, A);
  else
    hipLaunchKernelGGL(k_f_resid_restrict<false>, grid, block, 0, st, A);
}
