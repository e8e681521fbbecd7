// Deterministic scatter-add of element contributions (the "CSR scatter" of an assembly loop) without atomics.
//
// Element kernels PARK their contributions in a stash, stash[element * width + local index]: coalesced, no conflicts.  One
// thread per destination then sums the stash entries that map to it, in ascending source order - a segmented reduction over
// the inverted destination table, built once on the host from the same per-element destination table the atomic version
// indexed.  Same flops, one extra streaming pass over the stash (8 B written + 8 B read per contribution; < 1 % of a Newton
// step next to the factorisation), and the result is BITWISE reproducible: run to run, and across the replicas of a
// distributed-LU handle (pgx_*_create_dist), whose ranks assemble redundantly and must agree to the last bit - with atomics
// their right-hand sides differed by rounding and iterative refinement stalled at exactly that difference (each rank's
// subtree corrected with respect to its own copy; tests/test_gpu_multiprocess.py found it on example 06).
// Reference counterpart: DOLFINx assemble_vector / assemble_matrix (src/lvpp/problem.py:61-63,76) are sequential per rank,
// hence deterministic by construction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <string>
#include <thread>
#include <utility>
#include <vector>

struct PgxScatter {
  int64_t nsrc = 0;   // stash length the table was built for
  int64_t ndst = 0;   // destinations that receive at least one contribution
  int32_t* dst = nullptr;  // [ndst] destination index, ascending (device)
  int32_t* ptr = nullptr;  // [ndst + 1] (device)
  int32_t* src = nullptr;  // [ptr[ndst]] stash indices, ascending within a destination (device)
};

// out[dst[j]] = (accumulate ? out[dst[j]] : 0) + scale * sum_k stash[src[k]],  k in [ptr[j], ptr[j+1])
static __global__ __launch_bounds__(256) void k_pgx_scatter(int64_t ndst, const int32_t* __restrict__ dst,
                                                             const int32_t* __restrict__ ptr, const int32_t* __restrict__ src,
                                                             const double* __restrict__ stash, double scale, int accumulate,
                                                             double* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= ndst) return;
  double s = 0.0;
  for (int32_t k = ptr[j]; k < ptr[j + 1]; ++k) s += stash[src[k]];
  const int32_t d = dst[j];
  out[d] = accumulate ? out[d] + scale * s : scale * s;
}

static inline void pgx_scatter_run(hipStream_t st, const PgxScatter& S, const double* stash, double scale, int accumulate,
                                   double* out) {
  if (S.ndst == 0) return;
  hipLaunchKernelGGL(k_pgx_scatter, dim3((unsigned)((S.ndst + 255) / 256)), dim3(256), 0, st, S.ndst, S.dst, S.ptr, S.src, stash,
                     scale, accumulate, out);
}

// dest[k] (host, k < nsrc): destination of stash entry k in an array of nout entries, or < 0 to drop the entry.
// Device arrays are allocated with hipMalloc and appended to `allocs` (freed by the owner).  Returns "" or an error text.
static inline std::string pgx_scatter_build(const int32_t* dest, int64_t nsrc, int64_t nout, std::vector<void*>& allocs,
                                            PgxScatter* S) {
  if (nsrc > 0x7fffffff || nout > 0x7fffffff) return "scatter table exceeds int32 indices";
  // Counting sort by destination, threads over DESTINATION ranges: every thread streams the whole table (sequential reads) but
  // counts / places only the entries of its own range, so the random writes of a thread stay inside its slice and no two threads
  // touch the same counter.  (Sequential version: 3 s for the 226 M-entry constant-block table of example 06 at 1024^2.)
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), (int64_t)16, nout / 65536 + 1}));
  std::vector<int32_t> cnt((size_t)nout + 1, 0);
  std::vector<int> bad(T, 0);
  auto range = [&](int t) { return std::make_pair((int64_t)t * nout / T, (int64_t)(t + 1) * nout / T); };
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        const auto [d0, d1] = range(t);
        for (int64_t k = 0; k < nsrc; ++k) {
          const int32_t d = dest[k];
          if (d >= nout) bad[t] = 1;
          if (d >= d0 && d < d1) cnt[(size_t)d + 1]++;
        }
      });
    for (auto& x : th) x.join();
  }
  for (int t = 0; t < T; ++t)
    if (bad[t]) return "scatter destination out of range";
  std::vector<int32_t> dst, ptr;
  int64_t total = 0;
  for (int64_t d = 0; d < nout; ++d)
    if (cnt[(size_t)d + 1]) {
      dst.push_back((int32_t)d);
      ptr.push_back((int32_t)total);
      total += cnt[(size_t)d + 1];
    }
  ptr.push_back((int32_t)total);
  // position of every destination's segment, then a stable fill: sources ascend within a segment
  std::vector<int32_t> pos((size_t)nout, -1);
  for (size_t j = 0; j < dst.size(); ++j) pos[(size_t)dst[j]] = ptr[j];
  std::vector<int32_t> src((size_t)total);
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        const auto [d0, d1] = range(t);
        for (int64_t k = 0; k < nsrc; ++k) {
          const int32_t d = dest[k];
          if (d >= d0 && d < d1) src[(size_t)pos[(size_t)d]++] = (int32_t)k;
        }
      });
    for (auto& x : th) x.join();
  }
  S->nsrc = nsrc;
  S->ndst = (int64_t)dst.size();
  auto up = [&](const std::vector<int32_t>& v, int32_t** p) -> bool {
    void* q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(v.size(), 1) * sizeof(int32_t)) != hipSuccess) return false;
    allocs.push_back(q);
    *p = (int32_t*)q;
    return v.empty() || hipMemcpy(q, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
  };
  if (!up(dst, &S->dst) || !up(ptr, &S->ptr) || !up(src, &S->src)) return "hipMalloc/hipMemcpy(scatter table) failed";
  return "";
}
