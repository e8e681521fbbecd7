// Scope helpers shared by the solver translation units: the per-solve guard and the optional roctx ranges.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>

// Tuning / A-B / test switches (kernel variants, tile sizes, thresholds, consistency checks).  libpgx.so NEVER reads them from the
// environment: the table is filled only through the C ABI (include/pgx.h: pgx_tuning_set), so a deployed library behaves the same
// whatever PGX_* variables happen to be set.  nullptr = not set.  The environment variables the library does read are the three
// documented run-time options PGX_COMM_TIMEOUT, PGX_ROCTX and PGX_ND_THREADS.
// The returned pointer stays valid for the life of the process (values are interned, never freed), so a concurrent
// pgx_tuning_set cannot pull it away from a caller.
const char* pgx_tune(const char* name);
// Number of pgx_tuning_set calls so far: lets the per-launch switches below cache their value without taking the table's lock.
int pgx_tune_gen();
// An integer switch read on a hot path (smoother / restriction launches): one relaxed load per call, the table lookup only
// after the table changed.
struct PgxTuneInt {
  const char* key;
  int def;
  std::atomic<int> gen{-1}, val{0};
  PgxTuneInt(const char* k, int d) : key(k), def(d) {}
  int get() {
    const int g = pgx_tune_gen();
    if (gen.load(std::memory_order_acquire) != g) {
      const char* e = pgx_tune(key);
      val.store(e ? atoi(e) : def, std::memory_order_relaxed);
      gen.store(g, std::memory_order_release);
    }
    return val.load(std::memory_order_relaxed);
  }
};

// roctx ranges around the solver phases (readable rocprofv3 --marker-trace timelines).  Off unless
// PGX_ROCTX=1; the marker library is opened with dlopen, so libpgx.so has no link-time dependency on the profiler.
//   pgx_roctx("name") pushes a range, pgx_roctx(nullptr) pops it.
static inline void pgx_roctx(const char* name) {
  typedef int (*push_t)(const char*);
  typedef int (*pop_t)();
  struct Fns {
    push_t push = nullptr;
    pop_t pop = nullptr;
    Fns() {
      const char* e = getenv("PGX_ROCTX");
      if (!e || atoi(e) == 0) return;
      for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
        void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        push = (push_t)dlsym(h, "roctxRangePushA");
        pop = (pop_t)dlsym(h, "roctxRangePop");
        if (push && pop) return;
        push = nullptr, pop = nullptr;
      }
    }
  };
  static const Fns f;  // thread-safe one-time initialisation
  if (!f.push) return;
  if (name)
    f.push(name);
  else
    f.pop();
}
struct PgxRange {
  explicit PgxRange(const char* name) { pgx_roctx(name); }
  ~PgxRange() { pgx_roctx(nullptr); }
  PgxRange(const PgxRange&) = delete;
  PgxRange& operator=(const PgxRange&) = delete;
};

// Wall-clock events of one Newton solve (profiling) + a flag to drop on EVERY exit path: the early `return rc` of a failed
// collective or HIP call must neither leak the events nor leave "precondition with the factorisation" set on the handle.
struct PgxSolveScope {
  hipStream_t st;
  hipEvent_t w0 = nullptr, w1 = nullptr;
  bool* flag;
  PgxSolveScope(hipStream_t s, bool prof, bool* f) : st(s), flag(f) {
    if (prof && hipEventCreate(&w0) == hipSuccess && hipEventCreate(&w1) == hipSuccess) hipEventRecord(w0, st);
  }
  float stop() {  // elapsed ms so far (0 when not profiling)
    float ms = 0;
    if (w0 && w1 && hipEventRecord(w1, st) == hipSuccess && hipEventSynchronize(w1) == hipSuccess) hipEventElapsedTime(&ms, w0, w1);
    return ms;
  }
  ~PgxSolveScope() {
    if (flag) *flag = false;
    if (w0) hipEventDestroy(w0);
    if (w1) hipEventDestroy(w1);
  }
  PgxSolveScope(const PgxSolveScope&) = delete;
  PgxSolveScope& operator=(const PgxSolveScope&) = delete;
};
