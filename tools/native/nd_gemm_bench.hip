// Standalone harness of the sparse LU's Schur-update GEMM kernels (proximalgalerkin_amd/csrc/pgx_nd_gemm.h): one front of
// pivot order P and border B, C[P:M, P:M] -= L21 U12 with K = P, each kernel timed and checked against the 64 x 64 kernel
// (and, for small fronts, against a host loop).   nd_gemm_bench P B [count]
#include "../../proximalgalerkin_amd/csrc/pgx_nd_gemm.h"

#include <cmath>
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

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 2048, B = argc > 2 ? atoi(argv[2]) : 8192, count = argc > 3 ? atoi(argv[3]) : 1;
  const int M = P + B;
  const int64_t fs = (int64_t)M * P + (int64_t)P * B, ws = (int64_t)M * M;
  const int64_t store_off = 0, lev_off = fs * count;
  const int64_t total = lev_off + ws * count;
  std::vector<double> h(total);
  unsigned long long s = 88172645463325252ull;
  for (auto& v : h) {
    s ^= s << 13, s ^= s >> 7, s ^= s << 17;
    v = (double)(s >> 11) / 9007199254740992.0 - 0.5;
  }
  double *arena, *ref;
  CK(hipMalloc(&arena, total * 8));
  CK(hipMalloc(&ref, ws * count * 8));
  NdGatherCtx gc{};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double flops = 2.0 * B * (double)B * P * count;
  auto reset = [&]() { CK(hipMemcpy(arena, h.data(), total * 8, hipMemcpyHostToDevice)); };
  auto timeit = [&](const char* name, auto launch, bool check) {
    reset();
    launch();
    CK(hipDeviceSynchronize());
    double err = -1;
    if (check) {
      std::vector<double> a(ws * count), b(ws * count);
      CK(hipMemcpy(a.data(), arena + lev_off, ws * count * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(b.data(), ref, ws * count * 8, hipMemcpyDeviceToHost));
      err = 0;
      for (int64_t i = 0; i < ws * count; ++i) err = std::max(err, std::fabs(a[i] - b[i]));
    }
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    printf("%-28s %9.3f ms  %7.2f TF/s  max|diff vs 64-tile| %.3e\n", name, best, flops / best / 1e9, err);
  };
  // reference: the 64 x 64 kernel
  reset();
  {
    dim3 grid(count, (B + 63) / 64, (B + 63) / 64);
    hipLaunchKernelGGL((k_nd_gemm<2, false>), grid, dim3(256), 0, 0, arena, lev_off, M, P, M, P, M, 0, P, store_off, P, gc, -1, 0, 0);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref, arena + lev_off, ws * count * 8, hipMemcpyDeviceToDevice));
  }
  if ((double)B * B * P < 3e8) {  // host check of the reference itself
    std::vector<double> c(ws);
    CK(hipMemcpy(c.data(), ref, ws * 8, hipMemcpyDeviceToHost));
    double err = 0;
    const double* S = h.data();
    const double* F = h.data() + lev_off;
    for (int j = P; j < M; ++j)
      for (int i = P; i < M; ++i) {
        double a = F[(int64_t)j * M + i];
        for (int k = 0; k < P; ++k) a -= S[(int64_t)k * M + i] * S[(int64_t)M * P + (int64_t)(j - P) * P + k];
        err = std::max(err, std::fabs(a - c[(int64_t)j * M + i]));
      }
    printf("64-tile kernel vs host loop: max|diff| %.3e\n", err);
  }
  for (int order = 0; order < 2; ++order) {  // 0: plain blockIdx -> tile, 1: XCD-chunked grouped tile order (pgx_nd_gemm.h)
    printf("tile order %d\n", order);
    timeit("k_nd_gemm<2> 64x64", [&]() {
      dim3 grid(count, (B + 63) / 64, (B + 63) / 64);
      hipLaunchKernelGGL((k_nd_gemm<2, false>), grid, dim3(256), 0, 0, arena, lev_off, M, P, M, P, M, 0, P, store_off, P, gc, -1, order, 0);
    }, true);
    timeit("k_nd_gemm8 128x128", [&]() {
      dim3 grid(count, (B + 127) / 128, (B + 127) / 128);
      hipLaunchKernelGGL((k_nd_gemm8<false>), grid, dim3(512), 0, 0, arena, lev_off, M, P, M, P, M, 0, P, store_off, P, gc, -1, order, 0);
    }, true);
  }
  return 0;
}
