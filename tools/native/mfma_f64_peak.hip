// What the fp64 matrix cores of one MI355X sustain: back-to-back v_mfma_f64_16x16x4_f64 on register operands, NACC independent
// accumulators per wave, W waves per SIMD, every CU busy.   hipcc --offload-arch=gfx950 -O3 mfma_f64_peak.hip -o mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
  v4d acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wg_per_cu, int iters) {
  double* out;
  const int nblk = 256 * wg_per_cu;
  hipMalloc(&out, (size_t)nblk * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  k<NACC><<<nblk, 256>>>(out, 10, 1.0, 1.0);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    k<NACC><<<nblk, 256>>>(out, iters, 1.0000001, 0.9999999);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double fl = (double)nblk * 4 * iters * NACC * 2048.0;
  printf("NACC %2d  waves/SIMD %d  %.3f ms  %.2f TF/s  (%.1f cycles per MFMA at 2.4 GHz)\n", NACC, wg_per_cu, best, fl / best / 1e9,
         best * 1e-3 * 2.4e9 / ((double)iters * NACC * wg_per_cu));
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<4>(w, 20000 / w);
    run<8>(w, 10000 / w);
    run<16>(w, 5000 / w);
  }
  return 0;
}
