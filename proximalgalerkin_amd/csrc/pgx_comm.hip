// Transports of the sharded path.  See pgx_comm.h and include/pgx.h ("Sharded path").
//
//  * RCCL: what a torch.distributed launch uses (one process per GPU).  The halo exchange is one grouped
//    ncclSend/ncclRecv batch to the two strip neighbours (point-to-point over the direct xGMI link; messages are
//    16 KB .. 300 KB, i.e. latency-bound), the reductions are ncclAllReduce on a few doubles.  Everything is enqueued
//    on the handle's stream: no host synchronisation is added to the solver.  librccl is opened with dlopen so that
//    libpgx.so itself has no link-time dependency on it (single-GPU users never touch it).
//  * local group: N communicators for N host threads of one process; same call sequence through host-synchronised
//    device copies.  Lets the complete sharded algorithm run on a one-GPU box (tests/test_gpu_sharded.py).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <thread>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/pgx.h"
#include "pgx_comm.h"

static thread_local std::string g_comm_error;
extern "C" const char* pgx_comm_last_error(void) { return g_comm_error.c_str(); }

// ------------------------------------------------------------------------------------------------
// RCCL
// ------------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // prefer a copy that is already in the process (torch ships its own librccl.so)
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* nm : names)
      if (!api.lib) api.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    for (const char* nm : names)
      if (!api.lib) api.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (!api.lib) api.lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!api.lib) {
      api.err = std::string("cannot open librccl: ") + dlerror();
      return;
    }
    auto sym = [&](const char* s) {
      void* p = dlsym(api.lib, s);
      if (!p && api.err.empty()) api.err = std::string("librccl lacks ") + s;
      return p;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  });
  return &api;
}

struct RcclComm : pgx_comm {
  RcclApi* a = nullptr;
  ncclComm_t c = nullptr;
  int device = 0;
  ~RcclComm() override {
    if (c) a->CommDestroy(c);
  }
  int fail(const char* what, ncclResult_t r) {
    err = std::string(what) + ": " + a->GetErrorString(r);
    return PGX_ECOMM;
  }
  int halo(hipStream_t st, double* const* f, int nf, size_t send_lo, size_t n_send_lo, size_t recv_lo, size_t n_recv_lo,
           size_t send_hi, size_t n_send_hi, size_t recv_hi, size_t n_recv_hi) override {
    const bool lo = rank > 0, hi = rank + 1 < size;
    if (!lo && !hi) return PGX_OK;
    ncclResult_t r = a->GroupStart();
    if (r != ncclSuccess) return fail("ncclGroupStart", r);
    for (int k = 0; k < nf && r == ncclSuccess; ++k) {
      if (lo) {
        r = a->Send(f[k] + send_lo, n_send_lo, ncclDouble, rank - 1, c, st);
        if (r == ncclSuccess) r = a->Recv(f[k] + recv_lo, n_recv_lo, ncclDouble, rank - 1, c, st);
      }
      if (hi && r == ncclSuccess) {
        r = a->Send(f[k] + send_hi, n_send_hi, ncclDouble, rank + 1, c, st);
        if (r == ncclSuccess) r = a->Recv(f[k] + recv_hi, n_recv_hi, ncclDouble, rank + 1, c, st);
      }
    }
    const ncclResult_t r2 = a->GroupEnd();
    if (r != ncclSuccess) return fail("ncclSend/ncclRecv", r);
    if (r2 != ncclSuccess) return fail("ncclGroupEnd", r2);
    return PGX_OK;
  }
  int allreduce(hipStream_t st, double* dev, size_t n) override {
    if (size == 1) return PGX_OK;
    const ncclResult_t r = a->AllReduce(dev, dev, n, ncclDouble, ncclSum, c, st);
    if (r != ncclSuccess) return fail("ncclAllReduce", r);
    return PGX_OK;
  }
  int gather0(hipStream_t st, const double* send, size_t n, double* recv0) override {
    if (size == 1 || n == 0) return PGX_OK;
    ncclResult_t r = a->GroupStart();
    if (r != ncclSuccess) return fail("ncclGroupStart", r);
    if (rank == 0) {
      for (int q = 1; q < size && r == ncclSuccess; ++q) r = a->Recv(recv0 + (size_t)q * n, n, ncclDouble, q, c, st);
    } else {
      r = a->Send(send, n, ncclDouble, 0, c, st);
    }
    const ncclResult_t r2 = a->GroupEnd();
    if (r != ncclSuccess) return fail("ncclSend/ncclRecv (gather0)", r);
    if (r2 != ncclSuccess) return fail("ncclGroupEnd", r2);
    return PGX_OK;
  }
  int scatter0(hipStream_t st, const double* send0, size_t n, double* recv) override {
    if (size == 1 || n == 0) return PGX_OK;
    ncclResult_t r = a->GroupStart();
    if (r != ncclSuccess) return fail("ncclGroupStart", r);
    if (rank == 0) {
      for (int q = 1; q < size && r == ncclSuccess; ++q) r = a->Send(send0 + (size_t)q * n, n, ncclDouble, q, c, st);
    } else {
      r = a->Recv(recv, n, ncclDouble, 0, c, st);
    }
    const ncclResult_t r2 = a->GroupEnd();
    if (r != ncclSuccess) return fail("ncclSend/ncclRecv (scatter0)", r);
    if (r2 != ncclSuccess) return fail("ncclGroupEnd", r2);
    return PGX_OK;
  }
};
}  // namespace

extern "C" int pgx_comm_rccl_unique_id(char id[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  RcclApi* a = rccl_api();
  if (!a->err.empty() || !id) {
    g_comm_error = id ? a->err : "null argument";
    return PGX_ECOMM;
  }
  ncclUniqueId u;
  const ncclResult_t r = a->GetUniqueId(&u);
  if (r != ncclSuccess) {
    g_comm_error = std::string("ncclGetUniqueId: ") + a->GetErrorString(r);
    return PGX_ECOMM;
  }
  memcpy(id, u.internal, 128);
  return PGX_OK;
}

extern "C" int pgx_comm_rccl_init(const char id[128], int rank, int size, int device, pgx_comm** out) {
  if (!id || !out || size < 1 || rank < 0 || rank >= size) {
    g_comm_error = "pgx_comm_rccl_init: bad argument";
    return PGX_EINVAL;
  }
  *out = nullptr;
  RcclApi* a = rccl_api();
  if (!a->err.empty()) {
    g_comm_error = a->err;
    return PGX_ECOMM;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_comm_error = "hipSetDevice failed";
    return PGX_ENODEV;
  }
  std::unique_ptr<RcclComm> c(new RcclComm());
  c->a = a;
  c->rank = rank;
  c->size = size;
  c->device = device;
  ncclUniqueId u;
  memcpy(u.internal, id, 128);
  const ncclResult_t r = a->CommInitRank(&c->c, size, u, rank);
  if (r != ncclSuccess) {
    g_comm_error = std::string("ncclCommInitRank: ") + a->GetErrorString(r);
    c->c = nullptr;
    return PGX_ECOMM;
  }
  *out = c.release();
  return PGX_OK;
}

// ------------------------------------------------------------------------------------------------
// in-process thread group
// ------------------------------------------------------------------------------------------------
namespace {
struct LocalShared {
  int n = 0;
  std::mutex m;
  std::condition_variable cv;
  int waiting = 0;
  unsigned long gen = 0;
  bool broken = false;
  struct Pub {
    double* f[8];
    int nf = 0;
    size_t send_lo = 0, n_send_lo = 0, send_hi = 0, n_send_hi = 0;
    std::vector<double> red;
    const double* p2p_src = nullptr;  // gather0: a rank's send buffer; scatter0: rank 0's send0
    double* p2p_dst = nullptr;        // scatter0: a rank's receive buffer
    size_t p2p_n = 0;
  };
  std::vector<Pub> pub;
  // all ranks arrive or the group is declared broken (a peer returned early with an error): no silent hang
  bool barrier() {
    std::unique_lock<std::mutex> lk(m);
    if (broken) return false;
    const unsigned long my = gen;
    if (++waiting == n) {
      waiting = 0;
      ++gen;
      cv.notify_all();
      return true;
    }
    if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return gen != my || broken; })) broken = true;
    if (broken) cv.notify_all();
    return !broken;
  }
};

struct LocalComm : pgx_comm {
  std::shared_ptr<LocalShared> s;
  int dead() {
    err = "local group: a peer rank did not arrive within 120 s (it failed, or the ranks made different calls)";
    return PGX_ECOMM;
  }
  int hipfail(hipError_t e) {
    err = std::string("local group: ") + hipGetErrorString(e);
    {
      std::lock_guard<std::mutex> lk(s->m);
      s->broken = true;
    }
    s->cv.notify_all();
    return PGX_EHIP;
  }
  int halo(hipStream_t st, double* const* f, int nf, size_t send_lo, size_t n_send_lo, size_t recv_lo, size_t n_recv_lo,
           size_t send_hi, size_t n_send_hi, size_t recv_hi, size_t n_recv_hi) override {
    if (size == 1) return PGX_OK;
    if (nf > 8) {
      err = "local group: more than 8 arrays per exchange";
      return PGX_EINVAL;
    }
    hipError_t e = hipStreamSynchronize(st);  // my rows are final before the neighbours read them
    if (e != hipSuccess) return hipfail(e);
    LocalShared::Pub& p = s->pub[rank];
    p.nf = nf;
    for (int k = 0; k < nf; ++k) p.f[k] = f[k];
    p.send_lo = send_lo;
    p.n_send_lo = n_send_lo;
    p.send_hi = send_hi;
    p.n_send_hi = n_send_hi;
    if (!s->barrier()) return dead();
    if (rank > 0) {
      const LocalShared::Pub& q = s->pub[rank - 1];
      if (q.nf != nf || q.n_send_hi != n_recv_lo) {
        err = "local group: halo layouts of neighbouring ranks disagree";
        return PGX_ECOMM;
      }
      for (int k = 0; k < nf && e == hipSuccess; ++k)
        e = hipMemcpyAsync(f[k] + recv_lo, q.f[k] + q.send_hi, n_recv_lo * sizeof(double), hipMemcpyDefault, st);
    }
    if (rank + 1 < size && e == hipSuccess) {
      const LocalShared::Pub& q = s->pub[rank + 1];
      if (q.nf != nf || q.n_send_lo != n_recv_hi) {
        err = "local group: halo layouts of neighbouring ranks disagree";
        return PGX_ECOMM;
      }
      for (int k = 0; k < nf && e == hipSuccess; ++k)
        e = hipMemcpyAsync(f[k] + recv_hi, q.f[k] + q.send_lo, n_recv_hi * sizeof(double), hipMemcpyDefault, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hipfail(e);
    if (!s->barrier()) return dead();  // the neighbours have read my rows: they may change again
    return PGX_OK;
  }
  int allreduce(hipStream_t st, double* dev, size_t n) override {
    if (size == 1) return PGX_OK;
    std::vector<double>& mine = s->pub[rank].red;
    mine.resize(n);
    hipError_t e = hipMemcpyAsync(mine.data(), dev, n * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hipfail(e);
    if (!s->barrier()) return dead();
    std::vector<double> sum(n, 0.0);
    for (int r = 0; r < size; ++r) {  // fixed rank order: bitwise identical on every rank
      const std::vector<double>& o = s->pub[r].red;
      if (o.size() != n) {
        err = "local group: all-reduce lengths of the ranks disagree";
        return PGX_ECOMM;
      }
      for (size_t i = 0; i < n; ++i) sum[i] += o[i];
    }
    e = hipMemcpyAsync(dev, sum.data(), n * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hipfail(e);
    if (!s->barrier()) return dead();
    return PGX_OK;
  }
};
}  // namespace

namespace {
struct LocalCommP2P : LocalComm {
  int gather0(hipStream_t st, const double* send, size_t n, double* recv0) override {
    if (size == 1 || n == 0) return PGX_OK;
    hipError_t e = hipStreamSynchronize(st);  // my buffer is final before rank 0 reads it
    if (e != hipSuccess) return hipfail(e);
    s->pub[rank].p2p_src = send;
    s->pub[rank].p2p_n = n;
    if (!s->barrier()) return dead();
    if (rank == 0) {
      for (int q = 1; q < size && e == hipSuccess; ++q) {
        if (s->pub[q].p2p_n != n) {
          err = "local group: gather0 lengths of the ranks disagree";
          return PGX_ECOMM;
        }
        e = hipMemcpyAsync(recv0 + (size_t)q * n, s->pub[q].p2p_src, n * sizeof(double), hipMemcpyDefault, st);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e != hipSuccess) return hipfail(e);
    }
    if (!s->barrier()) return dead();  // rank 0 has read every buffer
    return PGX_OK;
  }
  int scatter0(hipStream_t st, const double* send0, size_t n, double* recv) override {
    if (size == 1 || n == 0) return PGX_OK;
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hipfail(e);
    s->pub[rank].p2p_src = send0;
    s->pub[rank].p2p_dst = recv;
    s->pub[rank].p2p_n = n;
    if (!s->barrier()) return dead();
    if (rank != 0) {
      if (s->pub[0].p2p_n != n) {
        err = "local group: scatter0 lengths of the ranks disagree";
        return PGX_ECOMM;
      }
      e = hipMemcpyAsync(recv, s->pub[0].p2p_src + (size_t)rank * n, n * sizeof(double), hipMemcpyDefault, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e != hipSuccess) return hipfail(e);
    }
    if (!s->barrier()) return dead();
    return PGX_OK;
  }
};
}  // namespace

extern "C" int pgx_comm_local_group(int size, pgx_comm** out) {
  if (size < 1 || !out) {
    g_comm_error = "pgx_comm_local_group: bad argument";
    return PGX_EINVAL;
  }
  auto s = std::make_shared<LocalShared>();
  s->n = size;
  s->pub.resize(size);
  for (int r = 0; r < size; ++r) {
    LocalComm* c = new LocalCommP2P();
    c->rank = r;
    c->size = size;
    c->s = s;
    out[r] = c;
  }
  return PGX_OK;
}

// ------------------------------------------------------------------------------------------------
// inter-process transport through POSIX shared memory (host-staged)
// ------------------------------------------------------------------------------------------------
// One communicator per PROCESS, any mix of devices - in particular several processes on ONE GPU, which RCCL refuses.  A
// torch.distributed.run launch of bench.py with BENCH_COMM=shm therefore executes, on a one-GPU box, everything the real
// multi-GPU launch executes except RCCL's byte movement: process spawn, rendezvous, name broadcast, the collective call order
// of every rank, the watchdog (tests/test_gpu_multiprocess.py).  Layout of the segment: header (sense-reversing barrier on
// process-shared atomics, per-rank publication records) followed by one mailbox of `slot_bytes` per rank.  Every operation is
// "stage my part into my mailbox - barrier - read the peers' mailboxes - barrier"; payloads larger than a mailbox move in
// chunks.  host_mode (tests without a GPU): the buffers are host memory and the staging copies are memcpy.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>

namespace {
constexpr int kShmMaxRanks = 64;
struct ShmHeader {
  std::atomic<uint32_t> magic, attached, arrived, gen, broken;
  uint32_t size;
  uint64_t slot_bytes;
  struct Pub {
    uint64_t n_lo, n_hi, nf, n;
  } pub[kShmMaxRanks];
};
constexpr uint32_t kShmMagic = 0x70677863u;  // "pgxc"
constexpr size_t kShmHeaderBytes = (sizeof(ShmHeader) + 4095) / 4096 * 4096;

double now_s() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

struct ShmComm : pgx_comm {
  ShmHeader* hd = nullptr;
  char* base = nullptr;
  size_t map_bytes = 0;
  bool host_mode = false;
  double timeout_s = 120.0;
  std::string name;
  bool owner = false;
  ~ShmComm() override {
    if (base) munmap(base, map_bytes);
    if (owner) shm_unlink(name.c_str());
  }
  double* slot(int r) const { return (double*)(base + kShmHeaderBytes + (size_t)r * hd->slot_bytes); }
  size_t cap() const { return hd->slot_bytes / sizeof(double); }
  int dead() {
    err = "shm group: a peer rank did not arrive within the timeout (it failed, or the ranks made different calls)";
    return PGX_ECOMM;
  }
  int bad(const char* what) {
    hd->broken.store(1);
    err = std::string("shm group: ") + what;
    return PGX_ECOMM;
  }
  // all ranks arrive, or the group is declared broken: no silent hang
  bool barrier() {
    if (hd->broken.load()) return false;
    const uint32_t my = hd->gen.load();
    if (hd->arrived.fetch_add(1) + 1 == (uint32_t)size) {
      hd->arrived.store(0);
      hd->gen.fetch_add(1);
      return true;
    }
    const double t0 = now_s();
    for (unsigned spin = 0;; ++spin) {
      if (hd->gen.load() != my) return true;
      if (hd->broken.load()) return false;
      if (spin < 2000) {
        sched_yield();
      } else {
        timespec ts{0, 50000};
        nanosleep(&ts, nullptr);
        if ((spin & 1023) == 0 && now_s() - t0 > timeout_s) {
          hd->broken.store(1);
          return false;
        }
      }
    }
  }
  int d2h(hipStream_t st, double* host, const double* dev, size_t n) {
    if (!n) return PGX_OK;
    if (host_mode) {
      memcpy(host, dev, n * sizeof(double));
      return PGX_OK;
    }
    const hipError_t e = hipMemcpyAsync(host, dev, n * sizeof(double), hipMemcpyDeviceToHost, st);
    return e == hipSuccess ? PGX_OK : hipfail(e);
  }
  int h2d(hipStream_t st, double* dev, const double* host, size_t n) {
    if (!n) return PGX_OK;
    if (host_mode) {
      memcpy(dev, host, n * sizeof(double));
      return PGX_OK;
    }
    const hipError_t e = hipMemcpyAsync(dev, host, n * sizeof(double), hipMemcpyHostToDevice, st);
    return e == hipSuccess ? PGX_OK : hipfail(e);
  }
  int sync(hipStream_t st) {
    if (host_mode) return PGX_OK;
    const hipError_t e = hipStreamSynchronize(st);
    return e == hipSuccess ? PGX_OK : hipfail(e);
  }
  int hipfail(hipError_t e) {
    hd->broken.store(1);
    err = std::string("shm group: ") + hipGetErrorString(e);
    return PGX_EHIP;
  }
  int halo(hipStream_t st, double* const* f, int nf, size_t send_lo, size_t n_send_lo, size_t recv_lo, size_t n_recv_lo,
           size_t send_hi, size_t n_send_hi, size_t recv_hi, size_t n_recv_hi) override {
    if (size == 1) return PGX_OK;
    const bool lo = rank > 0, hi = rank + 1 < size;
    const size_t nlo = lo ? n_send_lo : 0, nhi = hi ? n_send_hi : 0, per = nlo + nhi;
    if ((size_t)nf * per > cap()) return bad("halo message exceeds the mailbox (raise the slot size)");
    int rc = PGX_OK;
    double* mine = slot(rank);
    for (int k = 0; k < nf && !rc; ++k) {
      rc = d2h(st, mine + (size_t)k * per, f[k] + send_lo, nlo);
      if (!rc) rc = d2h(st, mine + (size_t)k * per + nlo, f[k] + send_hi, nhi);
    }
    if (!rc) rc = sync(st);
    if (rc) return rc;
    hd->pub[rank] = {nlo, nhi, (uint64_t)nf, 0};
    if (!barrier()) return dead();
    if (lo) {
      const ShmHeader::Pub q = hd->pub[rank - 1];
      if (q.nf != (uint64_t)nf || q.n_hi != n_recv_lo) return bad("halo layouts of neighbouring ranks disagree");
      const double* src = slot(rank - 1);
      for (int k = 0; k < nf && !rc; ++k) rc = h2d(st, f[k] + recv_lo, src + (size_t)k * (q.n_lo + q.n_hi) + q.n_lo, n_recv_lo);
    }
    if (hi && !rc) {
      const ShmHeader::Pub q = hd->pub[rank + 1];
      if (q.nf != (uint64_t)nf || q.n_lo != n_recv_hi) return bad("halo layouts of neighbouring ranks disagree");
      const double* src = slot(rank + 1);
      for (int k = 0; k < nf && !rc; ++k) rc = h2d(st, f[k] + recv_hi, src + (size_t)k * (q.n_lo + q.n_hi), n_recv_hi);
    }
    if (!rc) rc = sync(st);
    if (rc) return rc;
    if (!barrier()) return dead();  // the neighbours have read my mailbox: it may be overwritten
    return PGX_OK;
  }
  int allreduce(hipStream_t st, double* dev, size_t n) override {
    if (size == 1) return PGX_OK;
    std::vector<double> sum;
    for (size_t off = 0; off < n; off += cap()) {
      const size_t m = std::min(cap(), n - off);
      int rc = d2h(st, slot(rank), dev + off, m);
      if (!rc) rc = sync(st);
      if (rc) return rc;
      hd->pub[rank].n = n;
      if (!barrier()) return dead();
      sum.assign(m, 0.0);
      for (int r = 0; r < size; ++r) {  // fixed rank order: bitwise identical on every rank
        if (hd->pub[r].n != n) return bad("all-reduce lengths of the ranks disagree");
        const double* o = slot(r);
        for (size_t i = 0; i < m; ++i) sum[i] += o[i];
      }
      rc = h2d(st, dev + off, sum.data(), m);
      if (!rc) rc = sync(st);
      if (rc) return rc;
      if (!barrier()) return dead();
    }
    return PGX_OK;
  }
  int gather0(hipStream_t st, const double* send, size_t n, double* recv0) override {
    if (size == 1 || n == 0) return PGX_OK;
    for (size_t off = 0; off < n; off += cap()) {
      const size_t m = std::min(cap(), n - off);
      int rc = PGX_OK;
      if (rank != 0) {
        rc = d2h(st, slot(rank), send + off, m);
        if (!rc) rc = sync(st);
        if (rc) return rc;
      }
      hd->pub[rank].n = n;
      if (!barrier()) return dead();
      if (rank == 0) {
        for (int q = 1; q < size && !rc; ++q) {
          if (hd->pub[q].n != n) return bad("gather0 lengths of the ranks disagree");
          rc = h2d(st, recv0 + (size_t)q * n + off, slot(q), m);
        }
        if (!rc) rc = sync(st);
        if (rc) return rc;
      }
      if (!barrier()) return dead();
    }
    return PGX_OK;
  }
  int scatter0(hipStream_t st, const double* send0, size_t n, double* recv) override {
    if (size == 1 || n == 0) return PGX_OK;
    const size_t chunk = cap() / (size_t)size;  // rank 0's mailbox holds one chunk per receiver
    if (!chunk) return bad("mailbox too small for scatter0");
    for (size_t off = 0; off < n; off += chunk) {
      const size_t m = std::min(chunk, n - off);
      int rc = PGX_OK;
      if (rank == 0) {
        for (int q = 1; q < size && !rc; ++q) rc = d2h(st, slot(0) + (size_t)q * chunk, send0 + (size_t)q * n + off, m);
        if (!rc) rc = sync(st);
        if (rc) return rc;
      }
      hd->pub[rank].n = n;
      if (!barrier()) return dead();
      if (rank != 0) {
        if (hd->pub[0].n != n) return bad("scatter0 lengths of the ranks disagree");
        rc = h2d(st, recv + off, slot(0) + (size_t)rank * chunk, m);
        if (!rc) rc = sync(st);
        if (rc) return rc;
      }
      if (!barrier()) return dead();
    }
    return PGX_OK;
  }
};
}  // namespace

extern "C" int pgx_comm_shm_init(const char* name, int rank, int size, uint64_t slot_bytes, int host_mode, pgx_comm** out) {
  if (!name || !out || size < 1 || size > kShmMaxRanks || rank < 0 || rank >= size || name[0] != '/') {
    g_comm_error = "pgx_comm_shm_init: bad argument (name must start with '/', size <= 64)";
    return PGX_EINVAL;
  }
  *out = nullptr;
  if (slot_bytes == 0) slot_bytes = (uint64_t)64 << 20;
  slot_bytes = (slot_bytes + 4095) / 4096 * 4096;
  const size_t bytes = kShmHeaderBytes + (size_t)size * slot_bytes;
  std::unique_ptr<ShmComm> c(new ShmComm());
  c->rank = rank;
  c->size = size;
  c->host_mode = host_mode != 0;
  c->name = name;
  if (const char* t = getenv("PGX_COMM_TIMEOUT")) c->timeout_s = std::max(1.0, atof(t));
  int fd = -1;
  if (rank == 0) {
    shm_unlink(name);
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) {
      g_comm_error = std::string("shm_open/ftruncate(") + name + "): " + strerror(errno);
      if (fd >= 0) close(fd);
      return PGX_ECOMM;
    }
    c->owner = true;
  } else {  // wait for rank 0 to create and size the segment
    const double t0 = now_s();
    while (true) {
      fd = shm_open(name, O_RDWR, 0600);
      if (fd >= 0) {
        struct stat sb;
        if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= bytes) {
          // A segment left behind by a crashed run under the same name (rank 0 unlinks the name only once every rank has attached)
          // would be attached to by a rank that arrives before this launch's rank 0 has recreated it.  Such a segment shows its
          // age: a complete or broken group, or a barrier generation beyond the first.  Keep polling until rank 0 replaces it.
          void* pm = mmap(nullptr, sizeof(ShmHeader), PROT_READ, MAP_SHARED, fd, 0);
          bool stale = false;
          if (pm != MAP_FAILED) {
            const ShmHeader* hh = (const ShmHeader*)pm;
            stale = hh->magic.load() == kShmMagic && (hh->gen.load() != 0 || hh->broken.load() != 0 || hh->attached.load() >= (uint32_t)size);
            munmap(pm, sizeof(ShmHeader));
          }
          if (!stale) break;
        }
        close(fd);
        fd = -1;
      }
      if (now_s() - t0 > c->timeout_s) {
        g_comm_error = std::string("shm segment ") + name + " did not appear (rank 0 missing?)";
        return PGX_ECOMM;
      }
      timespec ts{0, 2000000};
      nanosleep(&ts, nullptr);
    }
  }
  void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) {
    g_comm_error = std::string("mmap: ") + strerror(errno);
    return PGX_ECOMM;
  }
  c->base = (char*)m;
  c->map_bytes = bytes;
  c->hd = (ShmHeader*)m;
  if (rank == 0) {  // a fresh segment is zero-filled: counters start at 0
    c->hd->size = (uint32_t)size;
    c->hd->slot_bytes = slot_bytes;
    c->hd->magic.store(kShmMagic);
  } else {
    const double t0 = now_s();
    while (c->hd->magic.load() != kShmMagic) {
      if (now_s() - t0 > c->timeout_s) {
        g_comm_error = "shm segment was never initialised by rank 0";
        return PGX_ECOMM;
      }
      sched_yield();
    }
    if (c->hd->size != (uint32_t)size || c->hd->slot_bytes != slot_bytes) {
      g_comm_error = "shm segment was created with another size / slot size";
      return PGX_ECOMM;
    }
  }
  c->hd->attached.fetch_add(1);
  if (!c->barrier()) {
    g_comm_error = "shm group: not every rank attached";
    return PGX_ECOMM;
  }
  if (rank == 0) {  // everybody has mapped it: the name can go (the memory lives until the last unmap)
    shm_unlink(name);
    c->owner = false;
  }
  *out = c.release();
  return PGX_OK;
}

// Direct calls of a communicator's four operations (tests; `host` buffers need a communicator created with host_mode).
extern "C" int pgx_comm_allreduce(pgx_comm* c, double* buf, uint64_t n) {
  if (!c || !buf) return PGX_EINVAL;
  const int rc = c->allreduce(nullptr, buf, n);
  if (rc) g_comm_error = c->err;
  return rc;
}
extern "C" int pgx_comm_halo(pgx_comm* c, double* f0, double* f1, uint64_t send_lo, uint64_t n_send_lo, uint64_t recv_lo,
                             uint64_t n_recv_lo, uint64_t send_hi, uint64_t n_send_hi, uint64_t recv_hi, uint64_t n_recv_hi) {
  if (!c || !f0) return PGX_EINVAL;
  double* f[2] = {f0, f1};
  const int rc = c->halo(nullptr, f, f1 ? 2 : 1, send_lo, n_send_lo, recv_lo, n_recv_lo, send_hi, n_send_hi, recv_hi, n_recv_hi);
  if (rc) g_comm_error = c->err;
  return rc;
}
extern "C" int pgx_comm_gather0(pgx_comm* c, const double* send, uint64_t n, double* recv0) {
  if (!c) return PGX_EINVAL;
  const int rc = c->gather0(nullptr, send, n, recv0);
  if (rc) g_comm_error = c->err;
  return rc;
}
extern "C" int pgx_comm_scatter0(pgx_comm* c, const double* send0, uint64_t n, double* recv) {
  if (!c) return PGX_EINVAL;
  const int rc = c->scatter0(nullptr, send0, n, recv);
  if (rc) g_comm_error = c->err;
  return rc;
}

// Self-check of a communicator before the first solve (round 5): the two operations the sharded path
// enqueues thousands of times - the exchange of ghost entries with both strip neighbours and the packed all-reduce - run ONCE on a
// known pattern, are verified, and must complete within timeout_s.  A transport that is mis-wired (wrong device binding, two
// copies of librccl in one process, a peer that never joined, an xGMI link that does not come up) then ends the run with a
// message that NAMES the failing call instead of a watchdog kill minutes into the first solve.  Collective.  host != 0: the
// communicator was created in host mode (buffers are host memory).
extern "C" int pgx_comm_selfcheck(pgx_comm* c, int host, double timeout_s) {
  if (!c) return PGX_EINVAL;
  if (!(timeout_s > 0.0)) timeout_s = 10.0;
  const size_t N = 256, L = 4 * N + 8;
  const int rank = c->rank, size = c->size;
  auto pat = [](int r, int part, size_t i) { return 1000.0 * r + 100000.0 * part + (double)i; };  // part 1: sent down, 2: sent up
  std::vector<double> hbuf(L, -1.0);
  for (size_t i = 0; i < N; ++i) hbuf[N + i] = pat(rank, 1, i), hbuf[2 * N + i] = pat(rank, 2, i);
  for (size_t i = 0; i < 8; ++i) hbuf[4 * N + i] = (double)(rank + 1) * (double)(i + 1);
  double* buf = hbuf.data();
  hipStream_t st = nullptr;
  double* dev = nullptr;
  auto fail = [&](const std::string& m) {
    c->err = "communicator self-check (rank " + std::to_string(rank) + " of " + std::to_string(size) + "): " + m;
    g_comm_error = c->err;
    return PGX_ECOMM;
  };
  if (!host) {
    if (hipMalloc(&dev, L * sizeof(double)) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
      return fail("cannot allocate the test buffer / stream on the device");
    if (hipMemcpy(dev, hbuf.data(), L * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail("upload of the pattern failed");
    buf = dev;
  }
  // a step = enqueue + bounded wait; on a timeout the stream still holds the stuck operation: nothing is freed, the caller exits
  auto finish = [&](const char* what, int rc) -> int {
    if (rc) return fail(std::string(what) + ": " + c->err);
    if (host) return PGX_OK;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t q = hipStreamQuery(st);
      if (q == hipSuccess) return PGX_OK;
      if (q != hipErrorNotReady) return fail(std::string(what) + ": " + hipGetErrorString(q));
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
        return fail(std::string(what) + " was enqueued but did not complete within " + std::to_string((int)timeout_s) +
                    " s - the peers are not reachable through this transport");
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
  };
  double* f[1] = {buf};
  int rc = finish("halo exchange with the strip neighbours (ncclSend / ncclRecv for the RCCL transport)",
                  c->halo(st, f, 1, N, N, 0, N, 2 * N, N, 3 * N, N));
  if (rc) return rc;
  rc = finish("packed all-reduce (ncclAllReduce for the RCCL transport)", c->allreduce(st, buf + 4 * N, 8));
  if (rc) return rc;
  if (!host) {
    if (hipMemcpy(hbuf.data(), dev, L * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return fail("download of the result failed");
    hipFree(dev);
    hipStreamDestroy(st);
  }
  for (size_t i = 0; i < N; ++i) {
    if (rank > 0 && hbuf[i] != pat(rank - 1, 2, i)) return fail("ghost entries from rank-1 arrived wrong (entry " + std::to_string(i) + ")");
    if (rank + 1 < size && hbuf[3 * N + i] != pat(rank + 1, 1, i)) return fail("ghost entries from rank+1 arrived wrong (entry " + std::to_string(i) + ")");
    if (hbuf[N + i] != pat(rank, 1, i) || hbuf[2 * N + i] != pat(rank, 2, i)) return fail("the exchange overwrote entries it only had to send");
  }
  const double tri = 0.5 * size * (size + 1);
  for (size_t i = 0; i < 8; ++i)
    if (hbuf[4 * N + i] != tri * (double)(i + 1)) return fail("all-reduce returned a wrong sum (entry " + std::to_string(i) + ")");
  return PGX_OK;
}

extern "C" void pgx_comm_free(pgx_comm* c) { delete c; }
