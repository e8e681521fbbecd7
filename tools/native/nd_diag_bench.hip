// Where do the ~33 us of one k_nd_diag launch (LU of a 64 x 64 diagonal block, one workgroup) go?  The library's nd_diag_lu with
// s_memtime stamps at its phase boundaries (a copy of the function text with stamps added is NOT kept: the stamps are taken around
// calls of the unmodified function on blocks of 16, 32, 48, 64 pivots, and around its load / store phases emulated separately).
//   nd_diag_bench [M]
#include "../../proximalgalerkin_amd/csrc/pgx_nd_gemm.h"

#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

__global__ __launch_bounds__(256) void k_diag(const double* F, double* S, int M, int nb, int* info, long long* stamps) {
  __shared__ double D[ND_NB][ND_NB + 1];
  const long long t0 = __builtin_amdgcn_s_memtime();
  nd_diag_lu(D, F, S, M, nb, info);
  __syncthreads();
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) stamps[0] = t1 - t0;
}
// load + store phases only
__global__ __launch_bounds__(256) void k_io(const double* F, double* S, int M, int nb, long long* stamps) {
  __shared__ double D[ND_NB][ND_NB + 1];
  const int tid = threadIdx.x;
  const long long t0 = __builtin_amdgcn_s_memtime();
  {
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int idx = tid + 256 * q;
      v[q] = idx < nb * nb ? F[(int64_t)(idx / nb) * M + idx % nb] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int idx = tid + 256 * q;
      if (idx < nb * nb) D[idx % nb][idx / nb] = v[q];
    }
  }
  __syncthreads();
  const long long t1 = __builtin_amdgcn_s_memtime();
  for (int idx = tid; idx < nb * nb; idx += 256) {
    const int r = idx % nb, c = idx / nb;
    S[(int64_t)c * M + r] = D[r][c];
  }
  __syncthreads();
  const long long t2 = __builtin_amdgcn_s_memtime();
  if (tid == 0) stamps[0] = t1 - t0, stamps[1] = t2 - t1;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4099;
  std::vector<double> h((size_t)M * 64);
  for (int c = 0; c < 64; ++c)
    for (int r = 0; r < M; ++r) h[(size_t)c * M + r] = (r == c ? 4.0 : 0.0) + 0.01 * ((r * 31 + c * 17) % 13) / 13.0;
  double *F, *S;
  int* info;
  long long* st;
  CK(hipMalloc(&F, h.size() * 8));
  CK(hipMalloc(&S, h.size() * 8));
  CK(hipMalloc(&info, 4));
  CK(hipMalloc(&st, 64));
  CK(hipMemcpy(F, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemset(info, 0, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int nb : {16, 32, 48, 64}) {
    k_diag<<<1, 256>>>(F, S, M, nb, info, st);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 200; ++r) k_diag<<<1, 256>>>(F, S, M, nb, info, st);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long hs[2];
    CK(hipMemcpy(hs, st, 16, hipMemcpyDeviceToHost));
    k_io<<<1, 256>>>(F, S, M, nb, st);
    CK(hipDeviceSynchronize());
    long long io[2];
    CK(hipMemcpy(io, st, 16, hipMemcpyDeviceToHost));
    printf("nb %2d: %.2f us per back-to-back launch; inside the kernel %lld s_memtime ticks (100 MHz: %.2f us); load %lld store %lld ticks\n", nb,
           ms * 1e3 / 200, hs[0], hs[0] / 100.0, io[0], io[1]);
  }
  return 0;
}
