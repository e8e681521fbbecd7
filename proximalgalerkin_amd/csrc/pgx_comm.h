// Transport used by the sharded path (include/pgx.h, "Sharded path"): exchange of ghost vertex rows with the two
// strip neighbours and all-reduce(sum) of small packed device buffers.  Two implementations in pgx_comm.hip:
// RCCL (one process per GPU, everything enqueued on the caller's stream) and an in-process thread group.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include <string>

struct pgx_comm {
  int rank = 0, size = 1;
  std::string err;
  virtual ~pgx_comm() {}
  // nf arrays f[0..nf) share one layout.  Entries [send_lo, send_lo+n_send_lo) go to rank-1, [recv_lo, ..+n_recv_lo)
  // come from rank-1; the *_hi ranges talk to rank+1.  A rank without that neighbour ignores the pair.
  virtual int halo(hipStream_t st, double* const* f, int nf, size_t send_lo, size_t n_send_lo, size_t recv_lo,
                   size_t n_recv_lo, size_t send_hi, size_t n_send_hi, size_t recv_hi, size_t n_recv_hi) = 0;
  // in-place sum over all ranks of n doubles in device memory; identical result on every rank
  virtual int allreduce(hipStream_t st, double* dev, size_t n) = 0;
  // rank r > 0 sends n doubles to rank 0, which receives them at recv0 + r*n (slot 0 of recv0 is not touched)
  virtual int gather0(hipStream_t st, const double* send, size_t n, double* recv0) = 0;
  // rank 0 sends send0 + r*n to every rank r > 0, which receives n doubles at recv
  virtual int scatter0(hipStream_t st, const double* send0, size_t n, double* recv) = 0;
};
