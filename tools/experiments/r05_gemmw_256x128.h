// experiment record: see tools/experiments/README.md (r05_gemmw_256x128)

// 256 x 128 tiles for the large Schur updates (round 5): eight waves x (64 x 64 = 4 x 4 MFMA tiles), so a workgroup moves
// (256 + 128) panel entries per 2 * 256 * 128 flops - 24 flop per byte staged instead of the 16 of the 128 x 128 tile -, two LDS
// stages with ONE barrier per k-chunk, and an XCD-aware tile order on a 1-D grid: workgroup w runs on XCD w & 7, and the 32
// workgroups an XCD holds at a time form one PATCH of 4 x 8 tiles (1024 x 1024 entries of C) whose panel strips are fetched
// once into that XCD's L2 and shared (6x less traffic out of L2 than 32 unrelated tiles).
// Grid: 1-D, nd_gemmw_grid(count, rows, cols) workgroups of 512 threads.
#define ND_WM 256
#define ND_WN 128
#define ND_WPI 4  // patch: tiles along i
#define ND_WPJ 8  // patch: tiles along j
static inline unsigned nd_gemmw_grid(int64_t count, int rows, int cols) {
  const int64_t nti = (rows + ND_WM - 1) / ND_WM, ntj = (cols + ND_WN - 1) / ND_WN;
  const int64_t npatch = ((nti + ND_WPI - 1) / ND_WPI) * ((ntj + ND_WPJ - 1) / ND_WPJ);
  const int64_t G = count * npatch;
  return (unsigned)(((G + 7) / 8) * 8 * (ND_WPI * ND_WPJ));
}
template <bool GATHER>
__global__ __launch_bounds__(512, 2) void k_nd_gemmw(double* __restrict__ arena, int64_t lev_off, int M, int r0g, int r1g,
                                                     int c0g, int c1g, int k0, int k1, int64_t store_off, int P, NdGatherCtx gc,
                                                     int count) {
  constexpr int TM = ND_WM, TN = ND_WN, NT = 512, LA = TM + 16, LB = ND_KC + 1;
  constexpr int NA = ND_KC * TM / NT, NB_ = ND_KC * TN / NT;  // 8 and 4 staged elements per thread and chunk
  __shared__ double As[2][ND_KC][LA];
  __shared__ double Bs[2][TN][LB];
  // tile of this workgroup: XCD-major patches
  const int nti = (r1g - r0g + TM - 1) / TM, ntj = (c1g - c0g + TN - 1) / TN;
  const int npi = (nti + ND_WPI - 1) / ND_WPI, npj = (ntj + ND_WPJ - 1) / ND_WPJ, npatch = npi * npj;
  const unsigned w = blockIdx.x, sq = w >> 3;
  const int64_t g = (int64_t)(sq / (ND_WPI * ND_WPJ)) * 8 + (w & 7);
  if (g >= (int64_t)count * npatch) return;
  const int front = (int)(g / npatch), pp = (int)(g - (int64_t)front * npatch), t = (int)(sq % (ND_WPI * ND_WPJ));
  const int ti = (pp % npi) * ND_WPI + (t % ND_WPI), tj = (pp / npi) * ND_WPJ + (t / ND_WPI);
  if (ti >= nti || tj >= ntj) return;
  const int r0 = r0g + TM * ti, c0 = c0g + TN * tj;
  const int rmax = r1g, cmax = c1g;
  double* F = arena + lev_off + (int64_t)front * M * M;  // C: working matrix
  const int64_t MP = (int64_t)M * P;
  const double* S = arena + store_off + (int64_t)front * (MP + (int64_t)P * (M - P));  // A, B: solved panels (compact store)
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
  const int wi = (wv & 3) * 64, wj = (wv >> 2) * 64;
  nd_v4d acc[4][4];  // [tj][ti]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (nd_v4d){0.0, 0.0, 0.0, 0.0};
  // staging roles: A element q of a thread is (i = tid & 255, k = 2 q + (tid >> 8)); B element q is (k = tid & 15, j = (tid >> 4) + 32 q)
  const int ai = tid & (TM - 1), ak = tid >> 8, bk = tid & (ND_KC - 1), bj = tid >> 4;
  const bool arow = r0 + ai < rmax;
  const double* Ap = S + (int64_t)ak * M + r0 + ai;
  const double* Bp[NB_];
  bool bcol[NB_];
#pragma unroll
  for (int q = 0; q < NB_; ++q) {
    const int cj = c0 + bj + 32 * q;
    bcol[q] = cj < cmax;
    Bp[q] = S + (cj < P ? (int64_t)cj * M : MP + (int64_t)(cj - P) * P) + bk;
  }
  double ra[NA], rb[NB_];
  auto fetch = [&](int kc) {
    const int kn = k1 - kc;  // >= 1
#pragma unroll
    for (int q = 0; q < NA; ++q) ra[q] = (arow && 2 * q + ak < kn) ? Ap[(int64_t)(kc + 2 * q) * M] : 0.0;
#pragma unroll
    for (int q = 0; q < NB_; ++q) rb[q] = (bcol[q] && bk < kn) ? Bp[q][kc] : 0.0;
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NA; ++q) As[buf][2 * q + ak][ai] = ra[q];
#pragma unroll
    for (int q = 0; q < NB_; ++q) Bs[buf][bj + 32 * q][bk] = rb[q];
  };
  fetch(k0);
  stash(0);
  if (k0 + ND_KC < k1) fetch(k0 + ND_KC);
  __syncthreads();
  int buf = 0;
  for (int kc = k0; kc < k1; kc += ND_KC, buf ^= 1) {
#pragma unroll
    for (int kk = 0; kk < ND_KC; kk += 4) {
      const int kq = kk + (l >> 4);
      double uf[4], lf[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        uf[tt] = Bs[buf][wj + 16 * tt + (l & 15)][kq];
        lf[tt] = As[buf][kq][wi + 16 * tt + (l & 15)];
      }
#pragma unroll
      for (int tj2 = 0; tj2 < 4; ++tj2)
#pragma unroll
        for (int ti2 = 0; ti2 < 4; ++ti2) acc[tj2][ti2] = __builtin_amdgcn_mfma_f64_16x16x4f64(uf[tj2], lf[ti2], acc[tj2][ti2], 0, 0, 0);
    }
    if (kc + ND_KC < k1) {
      stash(buf ^ 1);
      if (kc + 2 * ND_KC < k1) fetch(kc + 2 * ND_KC);
    }
    __syncthreads();
  }
  // D[m][n] = sum_k U[k][j=m] L[i=n][k]: lane l holds n = l&15 (row i of C), m = (l>>4) + 4*reg (column j of C)
  if (GATHER) {
    const NdGatherSrc gs = nd_gather_src(gc, arena, gc.f0 + front);
    int a0[4], a1[4];
#pragma unroll
    for (int ti2 = 0; ti2 < 4; ++ti2) {
      const int i = r0 + wi + 16 * ti2 + (l & 15);
      a0[ti2] = (gs.S0 && i < rmax) ? gs.I0[i] : -1;
      a1[ti2] = (gs.S1 && i < rmax) ? gs.I1[i] : -1;
    }
#pragma unroll
    for (int tj2 = 0; tj2 < 4; ++tj2)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int j = c0 + wj + 16 * tj2 + (l >> 4) + 4 * reg;
        if (j >= cmax) continue;
        const int b0 = gs.S0 ? gs.I0[j] : -1, b1 = gs.S1 ? gs.I1[j] : -1;
#pragma unroll
        for (int ti2 = 0; ti2 < 4; ++ti2) {
          const int i = r0 + wi + 16 * ti2 + (l & 15);
          if (i >= rmax) continue;
          double v = 0.0;
          if ((a0[ti2] | b0) >= 0) v = gs.S0[(int64_t)b0 * gs.M0 + a0[ti2]];
          if ((a1[ti2] | b1) >= 0) v += gs.S1[(int64_t)b1 * gs.M1 + a1[ti2]];
          F[(int64_t)j * M + i] = v - acc[tj2][ti2][reg];
        }
      }
    return;
  }
#pragma unroll
  for (int tj2 = 0; tj2 < 4; ++tj2)
#pragma unroll
    for (int ti2 = 0; ti2 < 4; ++ti2) {
      const int i = r0 + wi + 16 * ti2 + (l & 15);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int j = c0 + wj + 16 * tj2 + (l >> 4) + 4 * reg;
        if (i < rmax && j < cmax) F[(int64_t)j * M + i] -= acc[tj2][ti2][reg];
      }
    }
}
