// pgx_nd.hip - geometric nested-dissection multifrontal LU on the GPU (C ABI: include/pgx_nd.h).
//
// Replaces PETSc "pc_type lu / pc_factor_mat_solver_type mumps" of the reference (obstacle_pg.py:129-131,
// gradient_constraint_dolfinx.py:118-121, signorini_dolfinx.py:271-279, thermoforming_dolfinx.py:105-107) for the Newton
// systems of the LVPP examples.  DESIGN.md section 9 has the measurements.
//
// Host (symbolic, once per pattern): node graph -> recursive coordinate bisection (separator = nodes of one half - the lighter choice -
// adjacent to the other half, leaves of 16 nodes) -> postorder, border ("struct") sets.  Fronts are grouped into BATCHES =
// one tree depth x one size class (<= 4 classes per depth, chosen to minimise padded flops + storage); every front of a
// batch is padded to the batch's pivot order P and border B, so a batch is one strided launch set.  Assembly destinations
// and child -> parent maps are precomputed.
// Device memory: the FACTORS of a front live compactly ([M x P: L11\\U11 over L21][P x B: U12]); the M x M matrix a front is
// assembled and eliminated in lives in one of two WORKING buffers chosen by the parity of its tree depth and reused two
// depths further up, once the parents have absorbed the Schur blocks - 2-D problems need ~1/2 of the storage of keeping
// every front whole (ex 06 1024^2: 77 -> 42 GB, ex 02 70^3: 139 -> 76 GB).
// Device (numeric, every Newton step), depth by depth, deepest first.  Assembly is PARENT-CENTRIC: every entry of a front is
// written once as the sum of its two children's Schur entries, found through inverse index maps (no zero fill, no
// read-modify-write, no atomics -> bitwise reproducible) - k_nd_gather does this for the FRAME of a front (pivot columns and
// pivot rows), where the CSR values are then added (k_nd_scatter_add; the assembly list is sorted by depth); the border block
// is assembled by the Schur-update GEMM itself, which takes its C operand through the same maps (GATHER variants).  The
// childless fronts of the deepest depth are assembled and eliminated by one wave each (k_nd_leaf: LDS tile, L columns in
// registers, pivot rows by v_readlane).  (PGX_ND_GATHER=0 / PGX_ND_LEAF_FUSED=0: zero fill + scatter + push-style
// k_nd_extend_add in two conflict-free passes, and the batched kernels for the leaves - the subtree cut of very large
// factorisations still uses the push path at the cut depth.)  Then for every batch of the depth - on forked HIP streams,
// joined per depth - a two-level blocked partial LU without pivoting across nodes; solved panels are written to the compact
// store only, which is also where the trailing updates read their operands:
//   k_nd_diag   LU of one <= 64-wide diagonal block in LDS (8-column panels by one wave with lane shuffles)
//   k_nd_panel_m both triangular panel solves against that block on the matrix cores (16-pivot blocked substitution; k_nd_panel:
//               the LDS-blocked scalar version)
//   k_nd_gemm   C -= A B on the fp64 matrix cores (v_mfma_f64_16x16x4_f64): 64^2 tiles by 4 waves x (2x2) MFMA tiles
//               (k_nd_gemm<2>) and 128^2 tiles by 8 waves x (4x2) MFMA tiles at 4 waves per SIMD (k_nd_gemm8), operands swapped (D^T = B^T A^T) so that the 16 lanes of an MFMA row write 128 contiguous bytes;
//               rank-64 updates touch only the strips of the current 256-pivot outer block, the rest of the trailing
//               matrix gets one rank-256 update per outer block, the Schur block F22 ONE update with K = P.
// Solve: the same tree walk on per-front vectors (forward leaves -> root, backward root -> leaves): k_nd_trsv on slabs of
// 256 pivots (64 x 64 triangles by wave shuffles), everything outside the slab by k_nd_gemv over many workgroups.
// Distributed (pgx_nd_create_dist): the tree is cut at depth log2(ranks); one subtree per rank, the levels above on rank 0
// with ghost copies of the other subtree roots; pgx_comm::gather0 / scatter0 / allreduce carry the Schur blocks, border
// vectors and the solution.
// (A first version used rocSOLVER getrf_npvt / rocBLAS trsm+gemm strided-batched: 859 ms at 1024^2 against 40 ms now,
// dominated by 4e5 tiny Tensile launches; profiles/r01_nd_rocblas_baseline_kernel_stats.csv.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/pgx.h"
#include "../../include/pgx_nd.h"
#include "pgx_comm.h"
#include "pgx_scope.h"

static thread_local std::string g_nd_error;

struct NdLevel {  // one BATCH: the fronts of one tree depth and one size class, padded to a common (P, B)
  int64_t start = 0, count = 0;
  int depth = 0;
  int P = 0, B = 0;
  int64_t off = 0;   // offset in the VIRTUAL arena (every front a full M x M matrix; symbolic maps, exports, tests)
  int64_t voff = 0;  // vector arena offset
  // device layout: the factors of a front are kept COMPACT ([M x P: L11\\U11 over L21][P x B: U12]) at poff + i * (MP + PB);
  // the M x M matrix it is assembled and factorised in lives in one of two WORKING buffers (tree depth parity) at
  // woff + i * M*M, which the fronts two depths further up reuse once the parents have absorbed the Schur blocks
  int64_t poff = 0, woff = 0;
};

struct pgx_nd {
  int device = -1;
  hipStream_t st = nullptr;
  bool own_stream = false;
  std::string err;
  int64_t n = 0, nnz = 0, nfronts = 0;
  std::vector<NdLevel> lev;
  std::vector<int32_t> fp, fb, parent, slot01, child0, child1, flevel;
  std::vector<int64_t> dof_ptr, rel_ptr, dest, fbase;
  std::vector<int32_t> own_dofs, rel;
  pgx_nd_stats stats{};
  int64_t arena_len = 0, vec_len = 0;  // arena_len: device layout = compact factors + both working buffers
  int64_t virt_len = 0;                 // virtual arena (full fronts)
  // GROUPS = units of the factorisation schedule: all batches of one tree depth, or - below the cut depth kcut of a
  // large factorisation - the batches of one depth inside ONE of the subtrees hanging at depth kcut.  The subtrees are
  // factorised one after the other (each level by level), so the working buffers of the deep levels only ever hold one
  // subtree's fronts; the levels from kcut upwards follow as before.
  struct Group {
    int depth = 0, sub = -1;   // sub < 0: every front of the depth
    int l0 = 0, l1 = 0;        // batches lev[l0 .. l1)
    int64_t w_off = 0, w_len = 0;  // its range of a working buffer
    int64_t nz0 = 0, nz1 = 0;      // its range of the group-sorted assembly list
    bool leaf_fused = false;       // deepest depth, small fronts: assembled + eliminated by k_nd_leaf (no prep, no diag/panel/gemm)
    bool frame_fused = false;      // small fronts WITH children: frame gathered + eliminated by k_nd_leaf<.,true>, border block by the GATHER GEMM
  };
  std::vector<Group> groups;
  std::vector<int> gfirst;  // groups of depth d: groups[gfirst[d]] .. groups[gfirst[d+1] - 1]
  int kcut = -1, nsub = 0;
  // fused leaves: per-front lists of the matrix entries (front-local position c*M + r, index of the entry), CSR over ALL fronts
  // (empty rows for the fronts of other groups)
  int64_t* d_leaf_ptr = nullptr;
  int32_t *d_leaf_loc = nullptr, *d_leaf_src = nullptr;
  bool leaf_fuse = true;                // PGX_ND_LEAF_FUSED=0: the level-batched kernels for the leaves too (A/B)
  // parent-centric assembly (k_nd_gather): every entry of a front = sum of its two children's Schur entries, WRITTEN once - no zero
  // fill, no read-modify-write.  inv[0|1][vbase[f] + p] = index in child 0|1's border of the parent-local index p, or -1.
  int32_t* d_inv[2] = {nullptr, nullptr};
  bool gather = true;                   // PGX_ND_GATHER=0: zero fill + push-style extend-add (A/B)
  int64_t* d_sdest = nullptr;           // assembly list sorted by tree depth: destination in the working buffer ...
  int32_t* d_ssrc = nullptr;            // ... and index of the matrix entry
  // device
  double *arena = nullptr, *vec = nullptr, *d_vals = nullptr, *d_b = nullptr;
  int64_t *d_dof_ptr = nullptr, *d_rel_ptr = nullptr, *d_fbase = nullptr, *d_vbase = nullptr;
  int32_t *d_fp = nullptr, *d_fb = nullptr, *d_parent = nullptr, *d_slot01 = nullptr, *d_child0 = nullptr,
          *d_child1 = nullptr, *d_own_dofs = nullptr, *d_rel = nullptr, *d_fM = nullptr, *d_fP = nullptr;
  int* d_info = nullptr;  // [0] = number of (near-)zero pivots met by the last factorisation
  int* h_info = nullptr;  // pinned copy, filled by an async D2H at the end of pgx_nd_factor
  hipEvent_t ev_info = nullptr;
  bool info_pending = false;
  // distributed factorisation (pgx_nd_create_dist): the 2^kdist subtrees below tree depth kdist live on one rank each,
  // the levels above on rank 0, which also holds GHOST copies of the other ranks' subtree-root fronts (identity pivot
  // block; their Schur block / border vector arrives through pgx_comm::gather0, leaves through scatter0)
  std::vector<int> dfirst;  // batches of tree depth d: lev[dfirst[d]] .. lev[dfirst[d+1] - 1]
  // the batches (size classes) of one tree depth are independent: their launch chains run on side streams, forked from
  // and joined into the main stream per depth (few large fronts near the root cannot fill 256 CUs one class at a time)
  hipStream_t side[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  // the working buffer of depth d - 1 (zero fill, matrix entries, padding) is prepared on its own stream while depth d is
  // eliminated: it is the buffer depth d + 1 has just vacated (extend-add of depth d + 1 is enqueued before)
  hipStream_t prep_st = nullptr;
  hipEvent_t ev_prep_go = nullptr, ev_prep_done = nullptr;
  bool prep_ahead = true;  // PGX_ND_PREP_AHEAD=0: everything on the main stream
  pgx_comm* comm = nullptr;
  int kbatch = 0;  // distributed: the (single) batch holding the subtree roots at depth kdist
  int rank = 0, size = 1, kdist = 0;
  int root_slot = -1;             // this rank's subtree-root front (depth kdist)
  std::vector<int> ghost_slot;    // rank 0: slot of rank j's subtree-root front (index j; [0] unused)
  double *d_xbuf = nullptr, *d_vbuf = nullptr;
  bool factored = false;
  bool timing = false;
  int outer = 512;          // PGX_ND_OUTER: pivots per outer block = rank of the trailing updates (multiple of 64).  512 since round 5: with
                            // left-looking steps inside the block, ex 02 at 70^3 990 -> 965 ms, ex 06 at 1024^2 100.1 -> 99.5 ms against 256
                            // on the same box (768 / 1024: within 0.5 %); with round 2`s right-looking strips 512 had lost on ex 06
  // SYMMETRIC MODE (round 5, pgx_nd_set_symmetric): the caller's matrix is symmetric (A = A^T; indefinite is fine - there is no pivoting
  // across blocks either way).  The factorisation is then L D L^T in LU clothing: U12 = D11 L21^T is written by the panel kernel as the
  // scaled transpose of the L panel it has just solved - no U panel solve, no assembly of the pivot rows right of the pivot block -, the
  // updates of the pivot block run on its lower part (block columns instead of L-shaped regions) and the Schur updates on the tiles on
  // and below the diagonal; every gather of a child's Schur block reads (row, col) at (max, min).  Half the flops of the factorisation;
  // the compact store and the solve phase are unchanged (U is materialised).  PGX_ND_SYM=0 ignores the request (A/B).
  int sym = 0;
  bool sym_allowed = true;
  int tile_order = 1;       // PGX_ND_TILEORDER=0: plain blockIdx -> tile mapping in the GEMM kernels (A/B; pgx_nd_gemm.h nd_block_tile)
  bool leftlook = true;     // PGX_ND_LEFTLOOK=0: right-looking rank-64 strip updates inside an outer block (A/B)
  bool trsv_big = true;     // PGX_ND_TRSV_BIG=0: k_nd_trsv for the batches of few large fronts too (A/B)
  bool lshape = true;       // PGX_ND_LSHAPE=0: the trailing update of an outer block as three rectangles (A/B)
  bool solve_small = true;  // PGX_ND_SOLVE_SMALL=0: the three-launch path for small fronts too (A/B)
  int panel_kind = 0;  // PGX_ND_PANEL: 0 MFMA (default), 1 LDS-blocked scalar, 2 register-column scalar panel kernel
  double factor_ms = 0, solve_ms = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  // per-depth device time (pgx_nd_depth_profile; diagnostic): events on the main stream at the depth boundaries of the
  // factorisation and of both sweeps of the solve - [phase][depth], phase 0 factor, 1 forward, 2 backward
  bool dprof = false;
  std::vector<hipEvent_t> dp_ev;
  std::vector<std::pair<int, int>> dp_tag;  // (phase, depth) of the interval that ENDS at event i + 1
  std::vector<double> dp_ms[3];
  int dp_calls[3] = {0, 0, 0};
  std::vector<void*> allocs;
};

extern "C" const char* pgx_nd_last_error(const pgx_nd* s) { return s ? s->err.c_str() : g_nd_error.c_str(); }

// ------------------------------------------------------------------------------------------------------------------
// symbolic phase (host)
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct TNode {
  int parent = -1, child[2] = {-1, -1}, depth = 0;
  std::vector<int32_t> own;     // graph nodes eliminated at this tree node
  std::vector<int32_t> border;  // graph nodes of ancestors coupled to the subtree, sorted by elimination position
};

struct Symbolic {
  const pgx_nd_matrix* A;
  int64_t n;
  int32_t nn, dim, leaf;
  std::vector<int64_t> nd_ptr;
  std::vector<int32_t> nd_dofs;
  std::vector<int64_t> gptr;
  std::vector<int32_t> gadj;
  std::vector<TNode> T;
  std::vector<int32_t> mark;
  int32_t stamp = 0;
  std::vector<double> keys;

  void build_graph() {
    const int32_t* nod = A->node_of_dof;
    nd_ptr.assign(nn + 1, 0);
    for (int64_t i = 0; i < n; ++i) nd_ptr[nod[i] + 1]++;
    for (int32_t g = 0; g < nn; ++g) nd_ptr[g + 1] += nd_ptr[g];
    nd_dofs.resize(n);
    std::vector<int64_t> fill(nd_ptr.begin(), nd_ptr.end() - 1);
    for (int64_t i = 0; i < n; ++i) nd_dofs[fill[nod[i]]++] = (int32_t)i;
    gptr.assign(nn + 1, 0);
    std::vector<int32_t> seen(nn, -1);
    for (int pass = 0; pass < 2; ++pass) {
      std::fill(seen.begin(), seen.end(), -1);
      for (int32_t g = 0; g < nn; ++g) {
        int64_t cnt = 0;
        for (int64_t q = nd_ptr[g]; q < nd_ptr[g + 1]; ++q) {
          int32_t i = nd_dofs[q];
          for (int32_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k) {
            int32_t h = nod[A->col[k]];
            if (h != g && seen[h] != g) {
              seen[h] = g;
              if (pass) gadj[gptr[g] + cnt] = h;
              ++cnt;
            }
          }
        }
        if (!pass) gptr[g + 1] = cnt;
      }
      if (!pass) {
        for (int32_t g = 0; g < nn; ++g) gptr[g + 1] += gptr[g];
        gadj.resize(gptr[nn]);
      }
    }
  }

  int bisect(int32_t* V, int64_t cnt, int parent, int depth) {
    int tid = (int)T.size();
    T.emplace_back();
    T[tid].parent = parent;
    T[tid].depth = depth;
    if (cnt <= leaf) {
      T[tid].own.assign(V, V + cnt);
      return tid;
    }
    const double* X = A->node_coords;
    int ax = 0;
    double best = -1;
    for (int d = 0; d < dim; ++d) {
      double lo = 1e300, hi = -1e300;
      for (int64_t k = 0; k < cnt; ++k) {
        double c = X[(int64_t)V[k] * dim + d];
        lo = std::min(lo, c);
        hi = std::max(hi, c);
      }
      if (hi - lo > best) best = hi - lo, ax = d;
    }
    keys.resize(cnt);
    for (int64_t k = 0; k < cnt; ++k) keys[k] = X[(int64_t)V[k] * dim + ax];
    std::nth_element(keys.begin(), keys.begin() + cnt / 2, keys.begin() + cnt);
    const double med = keys[cnt / 2];
    auto key_of = [&](int32_t g) { return X[(int64_t)g * dim + ax]; };
    int32_t* mid = std::stable_partition(V, V + cnt, [&](int32_t g) { return key_of(g) < med; });
    int64_t na = mid - V;
    if (na == 0 || na == cnt) {  // more than half of the nodes share the extreme coordinate: split by count
      std::stable_sort(V, V + cnt, [&](int32_t a, int32_t b) {
        for (int d = 0; d < dim; ++d) {
          int dd = (ax + d) % dim;
          double ca = X[(int64_t)a * dim + dd], cb = X[(int64_t)b * dim + dd];
          if (ca != cb) return ca < cb;
        }
        return a < b;
      });
      na = cnt / 2;
    }
    // separator: the nodes of ONE half that are adjacent to the other half - whichever side gives fewer dofs.  (On the P2 / Q2
    // lattices only every second node line is a one-line separator: with the median on a vertex line the lower half's boundary
    // is TWO lines thick - the edge-midpoint line and the vertex line below it, 1.5x the dofs, 3.4x the flops of that front -
    // while the upper half's boundary is the vertex line itself.)
    auto weight = [&](int32_t g) { return nd_ptr[g + 1] - nd_ptr[g]; };
    auto touches = [&](int32_t g) {
      for (int64_t q = gptr[g]; q < gptr[g + 1]; ++q)
        if (mark[gadj[q]] == stamp) return true;
      return false;
    };
    ++stamp;
    for (int64_t k = na; k < cnt; ++k) mark[V[k]] = stamp;
    int64_t w_lo = 0, w_hi = 0;
    for (int64_t k = 0; k < na; ++k)
      if (touches(V[k])) w_lo += weight(V[k]);
    ++stamp;
    for (int64_t k = 0; k < na; ++k) mark[V[k]] = stamp;
    for (int64_t k = na; k < cnt; ++k)
      if (touches(V[k])) w_hi += weight(V[k]);
    int64_t n_first, n_second;  // children: V[0, n_first) and V[n_first, n_first + n_second); the separator follows them
    if (w_hi < w_lo) {          // upper side: [lower | interior of the upper half | separator]
      int32_t* sep_begin = std::stable_partition(V + na, V + cnt, [&](int32_t g) { return !touches(g); });
      n_first = na;
      n_second = (sep_begin - V) - na;
    } else {  // lower side: [interior of the lower half | separator | upper] -> rotate the separator to the end
      ++stamp;
      for (int64_t k = na; k < cnt; ++k) mark[V[k]] = stamp;
      int32_t* sep_begin = std::stable_partition(V, V + na, [&](int32_t g) { return !touches(g); });
      n_first = sep_begin - V;
      n_second = cnt - na;
      std::rotate(sep_begin, V + na, V + cnt);
    }
    T[tid].own.assign(V + n_first + n_second, V + cnt);
    int nc = 0;
    if (n_first > 0) {
      int c = bisect(V, n_first, tid, depth + 1);
      T[tid].child[nc++] = c;
    }
    if (n_second > 0) {
      int c = bisect(V + n_first, n_second, tid, depth + 1);
      T[tid].child[nc++] = c;
    }
    return tid;
  }
};
}  // namespace

static int nd_symbolic(pgx_nd* s, const pgx_nd_matrix* A) {
  const int drank = s->rank, dsize = s->size;
  Symbolic S;
  S.A = A;
  S.n = A->n;
  S.nn = A->n_nodes;
  S.dim = A->dim;
  S.leaf = A->leaf_nodes > 0 ? A->leaf_nodes : 16;  // 1024^2 P1 sweep: 8: 57 ms, 16: 60, 32: 62, 64: 75, 128: 89 per factorisation
  const int64_t n = S.n;
  const int32_t nn = S.nn;
  for (int64_t i = 0; i < n; ++i)
    if (A->node_of_dof[i] < 0 || A->node_of_dof[i] >= nn) {
      s->err = "node_of_dof out of range";
      return PGX_EINVAL;
    }
  const bool ptime = pgx_tune("PGX_ND_TIMING") != nullptr;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  auto t_0 = tnow();
  S.build_graph();
  auto t_1 = tnow();
  S.mark.assign(nn, 0);
  std::vector<int32_t> ids(nn);
  std::iota(ids.begin(), ids.end(), 0);
  S.T.reserve(4 * (size_t)(nn / S.leaf + 2));
  S.bisect(ids.data(), nn, -1, 0);
  auto t_2 = tnow();
  std::vector<TNode>& T = S.T;
  const int nt = (int)T.size();
  // postorder
  std::vector<int> post;
  post.reserve(nt);
  {
    std::vector<std::pair<int, int>> stack;
    stack.push_back({0, 0});
    while (!stack.empty()) {
      auto [t, k] = stack.back();
      stack.pop_back();
      if (k < 2 && T[t].child[k] >= 0) {
        stack.push_back({t, k + 1});
        stack.push_back({T[t].child[k], 0});
      } else if (k < 2) {
        stack.push_back({t, 2});
      } else {
        post.push_back(t);
      }
    }
  }
  std::vector<int> order_of(nt);
  for (int k = 0; k < nt; ++k) order_of[post[k]] = k;
  std::vector<int32_t> tnode(nn);
  std::vector<int64_t> node_pos(nn);
  {
    int64_t k = 0;
    for (int t : post)
      for (int32_t g : T[t].own) tnode[g] = t, node_pos[g] = k++;
    if (k != nn) {
      s->err = "nested dissection lost nodes";
      return PGX_EINVAL;
    }
  }
  // border sets, bottom-up
  {
    std::vector<int32_t> seen(nn, -1);
    for (int t : post) {
      std::vector<int32_t>& b = T[t].border;
      auto consider = [&](int32_t h) {
        if (order_of[tnode[h]] > order_of[t] && seen[h] != t) seen[h] = t, b.push_back(h);
      };
      for (int32_t g : T[t].own)
        for (int64_t q = S.gptr[g]; q < S.gptr[g + 1]; ++q) consider(S.gadj[q]);
      for (int c = 0; c < 2; ++c)
        if (T[t].child[c] >= 0)
          for (int32_t h : T[T[t].child[c]].border) consider(h);
      std::sort(b.begin(), b.end(), [&](int32_t a, int32_t c) { return node_pos[a] < node_pos[c]; });
    }
  }
  // distribution: state[t] = 0 not on this rank, 1 owned, 2 ghost (rank 0's copy of another rank's subtree root)
  int kd = 0;
  while ((1 << kd) < dsize) ++kd;
  if ((1 << kd) != dsize) {
    s->err = "distributed factorisation needs a power-of-two number of ranks";
    return PGX_EINVAL;
  }
  s->kdist = kd;
  std::vector<uint8_t> state(nt, 1);
  if (dsize > 1) {
    std::vector<int> owner(nt, -1);
    int next_sub = 0;
    for (int t : post) {  // postorder visits the depth-kd nodes left to right
      if (T[t].depth < kd && (T[t].child[0] < 0 || T[t].child[1] < 0)) {
        s->err = "matrix too small to distribute over this many ranks (the dissection tree is shallower than log2(size))";
        return PGX_EINVAL;
      }
      if (T[t].depth == kd) owner[t] = next_sub++;
    }
    if (next_sub != dsize) {
      s->err = "distributed factorisation: unexpected number of subtrees";
      return PGX_EINVAL;
    }
    for (int t = 0; t < nt; ++t)  // tree ids are assigned parent-before-child
      if (T[t].depth > kd) owner[t] = owner[T[t].parent];
    s->ghost_slot.assign(dsize, -1);
    for (int t = 0; t < nt; ++t) {
      if (T[t].depth < kd)
        state[t] = drank == 0 ? 1 : 0;
      else if (owner[t] == drank)
        state[t] = 1;
      else
        state[t] = (drank == 0 && T[t].depth == kd) ? 2 : 0;
    }
  }
  // Batches: the fronts of one tree depth, split into up to four SIZE CLASSES so that padding every front of a batch to
  // the batch's largest (P, B) wastes little (one class per depth executed 1.35x the algorithmic flops on the 3-D and
  // 1.31x on the 2-D Newton matrices).  Classes are chosen on ALL fronts of the depth (identical on every rank of a
  // distributed factorisation); slots are postorder inside a batch; only fronts living on this rank get a slot.
  int maxd = 0;
  for (auto& t : T) maxd = std::max(maxd, t.depth);
  std::vector<int> slot_of(nt, -1);
  auto ndofs = [&](int32_t g) { return (int)(S.nd_ptr[g + 1] - S.nd_ptr[g]); };
  std::vector<int> tp(nt), tb(nt), tp_true(nt), tbatch(nt, 0);
  int nloc = 0;
  for (int t = 0; t < nt; ++t) {
    int p = 0, b = 0;
    for (int32_t g : T[t].own) p += ndofs(g);
    for (int32_t g : T[t].border) b += ndofs(g);
    tp_true[t] = p;
    tp[t] = state[t] == 2 ? 0 : p;  // a ghost eliminates nothing here
    tb[t] = b;
    if (state[t]) ++nloc;
  }
  auto cost = [](double P, double B) { return 2.0 / 3 * P * P * P + 2 * P * P * B + 2 * P * B * B + 30.0 * (P + B) * (P + B); };
  // subtree of every node below the cut depth (see pgx_nd::Group); the cut is tried only when the uncut layout is large
  int kcut = -1;
  std::vector<int> sub_of(nt, -1);
  int nsub = 0;
  int64_t off = 0, voff = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
  s->lev.clear();
  s->groups.clear();
  s->gfirst.assign(maxd + 2, 0);
  s->dfirst.assign(maxd + 2, 0);
  std::fill(tbatch.begin(), tbatch.end(), 0);
  {
    std::vector<std::vector<int>> by_depth(maxd + 1);
    for (int t : post) by_depth[T[t].depth].push_back(t);
    for (int d = 0; d <= maxd; ++d) {
      s->dfirst[d] = (int)s->lev.size();
      s->gfirst[d] = (int)s->groups.size();
      const int nparts = (kcut >= 0 && d > kcut) ? nsub : 1;
      for (int part = 0; part < nparts; ++part) {
      pgx_nd::Group G;
      G.depth = d;
      G.sub = nparts > 1 ? part : -1;
      G.l0 = (int)s->lev.size();
      std::vector<int> order;  // postorder
      if (nparts == 1)
        order = by_depth[d];
      else
        for (int t : by_depth[d])
          if (sub_of[t] == part) order.push_back(t);
      std::vector<std::vector<int>> classes(1, order);
      if (order.empty()) classes.clear();
      if (!(dsize > 1 && d == kd) && order.size() > 1) {
        std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return tp_true[a] + tb[a] > tp_true[c] + tb[c]; });
        classes.assign(1, order);
        for (int round = 0; round < 2; ++round) {  // each class may split once per round: at most 4 classes
          std::vector<std::vector<int>> next;
          for (auto& cl : classes) {
            const int m = (int)cl.size();
            const int mins = m > 32 ? 4 : 1;  // few, large fronts near the root: even one front per batch pays
            if (m < 2 * mins) {
              next.push_back(cl);
              continue;
            }
            std::vector<int> pP(m), pB(m), sP(m), sB(m);
            for (int i = 0; i < m; ++i) {
              pP[i] = std::max(i ? pP[i - 1] : 0, tp_true[cl[i]]);
              pB[i] = std::max(i ? pB[i - 1] : 0, tb[cl[i]]);
            }
            for (int i = m - 1; i >= 0; --i) {
              sP[i] = std::max(i + 1 < m ? sP[i + 1] : 0, tp_true[cl[i]]);
              sB[i] = std::max(i + 1 < m ? sB[i + 1] : 0, tb[cl[i]]);
            }
            const double whole = m * cost(pP[m - 1], pB[m - 1]);
            double best = whole;
            int cut = -1;
            for (int i = mins - 1; i + mins < m; ++i) {  // first class cl[0..i], second cl[i+1..]
              const double c2 = (i + 1) * cost(pP[i], pB[i]) + (m - i - 1) * cost(sP[i + 1], sB[i + 1]);
              if (c2 < best) best = c2, cut = i;
            }
            if (cut >= 0 && best < 0.93 * whole) {
              next.emplace_back(cl.begin(), cl.begin() + cut + 1);
              next.emplace_back(cl.begin() + cut + 1, cl.end());
            } else {
              next.push_back(cl);
            }
          }
          classes.swap(next);
        }
        for (auto& cl : classes)  // back to postorder inside a class (spatial locality of the slots)
          std::sort(cl.begin(), cl.end(), [&](int a, int c) { return order_of[a] < order_of[c]; });
      }
      for (auto& cl : classes) {
        NdLevel Lv;
        Lv.depth = d;
        for (int t : cl) {
          Lv.P = std::max(Lv.P, tp_true[t]);
          Lv.B = std::max(Lv.B, tb[t]);
          tbatch[t] = (int)s->lev.size();
          if (state[t]) Lv.count++;
        }
        if (dsize > 1 && d == kd) s->kbatch = (int)s->lev.size();
        s->lev.push_back(Lv);
      }
      G.l1 = (int)s->lev.size();
      s->groups.push_back(G);
      }  // part
    }
    s->dfirst[maxd + 1] = (int)s->lev.size();
    s->gfirst[maxd + 1] = (int)s->groups.size();
  }
  const int L = (int)s->lev.size();
  int64_t start = 0;
  off = 0, voff = 0;
  s->stats = pgx_nd_stats();
  for (int l = 0; l < L; ++l) {
    NdLevel& Lv = s->lev[l];
    if (Lv.P == 0) Lv.P = 1;  // degenerate (only empty separators): keep a 1x1 identity pivot
    Lv.start = start;
    Lv.off = off;
    Lv.voff = voff;
    int64_t M = Lv.P + Lv.B;
    off += Lv.count * M * M;
    voff += Lv.count * M;
    start += Lv.count;
    double P = Lv.P, B = Lv.B;
    s->stats.flops_padded += Lv.count * (2.0 / 3 * P * P * P + 2 * P * P * B + 2 * P * B * B);
    if (Lv.count) s->stats.max_front = std::max<int64_t>(s->stats.max_front, M);
  }
  // k_nd_extend_add, k_nd_gather and k_nd_leaf index a front with 32-bit arithmetic (row * M + column)
  if (s->stats.max_front >= 65536) {
    s->err = "pgx_nd: a front of " + std::to_string(s->stats.max_front) + " rows exceeds the 65535 the assembly kernels index (32-bit front offsets)";
    return PGX_EINVAL;
  }
  s->virt_len = off;
  s->vec_len = voff;
  s->nfronts = nloc;
  s->stats.n_fronts = nloc;
  s->stats.n_levels = L;
  {  // device layout: compact factors first, then the working buffers: two (depth parity) for the groups that span a whole
     // depth, two smaller ones for the groups of the subtrees below the cut
    int64_t poff = 0, wmax[2][2] = {{0, 0}, {0, 0}};  // [deep?][parity]
    for (auto& G : s->groups) {
      int64_t w = 0;
      for (int l = G.l0; l < G.l1; ++l) {
        NdLevel& Lv = s->lev[l];
        const int64_t M = Lv.P + Lv.B;
        Lv.poff = poff;
        Lv.woff = w;  // relative to the group's buffer for now
        poff += Lv.count * (M * Lv.P + (int64_t)Lv.P * Lv.B);
        w += Lv.count * M * M;
      }
      G.w_len = w;
      int64_t& m = wmax[G.sub >= 0][G.depth & 1];
      m = std::max(m, w);
    }
    const int64_t base[2][2] = {{poff, poff + wmax[0][0]},
                                {poff + wmax[0][0] + wmax[0][1], poff + wmax[0][0] + wmax[0][1] + wmax[1][0]}};
    for (auto& G : s->groups) {
      G.w_off = base[G.sub >= 0][G.depth & 1];
      for (int l = G.l0; l < G.l1; ++l) s->lev[l].woff += G.w_off;
    }
    s->arena_len = poff + wmax[0][0] + wmax[0][1] + wmax[1][0] + wmax[1][1];
  }
  s->kcut = kcut;
  s->nsub = nsub;
  {
    const char* e = pgx_tune("PGX_ND_PANEL");
    s->panel_kind = e ? atoi(e) : 0;
  }
  // large factorisation on one GPU: cut the tree at depth 3 and factorise the (up to) 8 subtrees below one after the other
  {
    // threshold in GB of device storage; 0 = always, < 0 = never.  Default 160 of the 288 GB: the uncut schedule is the faster one
    // while it fits (2048^2 P2, 115 GB: factorisations -4 %, solves -31 % against the cut at 96 GB; tools/p2_cut_ab.py)
    const char* e = pgx_tune("PGX_ND_CUT_GB");
    const double thr = e ? atof(e) : 160.0;
    const int kc = 3;
    if (attempt == 0 && dsize == 1 && thr >= 0 && maxd >= kc + 3 && (double)s->arena_len * 8 > thr * 1e9) {
      nsub = 0;
      for (int t : post)
        if (T[t].depth == kc) sub_of[t] = nsub++;
      for (int t = 0; t < nt; ++t)  // tree ids are assigned parent-before-child
        if (T[t].depth > kc) sub_of[t] = sub_of[T[t].parent];
      bool ok = nsub >= 2;
      for (int t = 0; t < nt; ++t)
        if (T[t].depth > kc && sub_of[t] < 0) ok = false;  // a branch that ends above the cut
      if (ok) {
        kcut = kc;
        continue;
      }
      nsub = 0;
    }
  }
  break;
  }  // attempt
  const int L = (int)s->lev.size();
  s->stats.arena_doubles = s->arena_len;
  {
    std::vector<int64_t> next(L);
    for (int l = 0; l < L; ++l) next[l] = s->lev[l].start;
    for (int t : post)
      if (state[t]) slot_of[t] = (int)next[tbatch[t]]++;
  }
  s->fp.assign(nloc, 0);
  s->fb.assign(nloc, 0);
  s->parent.assign(nloc, -1);
  s->slot01.assign(nloc, 0);
  s->child0.assign(nloc, -1);
  s->child1.assign(nloc, -1);
  s->flevel.assign(nloc, 0);
  s->fbase.assign(nloc, 0);
  s->dof_ptr.assign(nloc + 1, 0);
  s->rel_ptr.assign(nloc + 1, 0);
  for (int t = 0; t < nt; ++t) {
    if (!state[t]) continue;
    int f = slot_of[t];
    s->fp[f] = tp[t];
    s->fb[f] = tb[t];
    s->flevel[f] = tbatch[t];
    const NdLevel& Lv = s->lev[tbatch[t]];
    int64_t M = Lv.P + Lv.B;
    s->fbase[f] = Lv.off + (f - Lv.start) * M * M;
    if (T[t].parent >= 0 && slot_of[T[t].parent] >= 0) s->parent[f] = slot_of[T[t].parent];
    for (int c = 0; c < 2; ++c)
      if (T[t].child[c] >= 0 && slot_of[T[t].child[c]] >= 0) {
        (c == 0 ? s->child0 : s->child1)[f] = slot_of[T[t].child[c]];
        s->slot01[slot_of[T[t].child[c]]] = c;
      }
    s->dof_ptr[f + 1] = tp[t];
    s->rel_ptr[f + 1] = tb[t];
    if (dsize > 1 && T[t].depth == kd && state[t] == 1) s->root_slot = f;
    if (state[t] == 1) {
      double p = tp[t], b = tb[t];
      s->stats.flops += 2.0 / 3 * p * p * p + 2 * p * p * b + 2 * p * b * b;
      s->stats.factor_nnz += (int64_t)(p * p + 2 * p * b);
    }
  }
  if (dsize > 1 && drank == 0) {
    int j = 0;
    for (int t : post)
      if (T[t].depth == kd) s->ghost_slot[j++] = slot_of[t];
  }
  for (int f = 0; f < nloc; ++f) s->dof_ptr[f + 1] += s->dof_ptr[f], s->rel_ptr[f + 1] += s->rel_ptr[f];
  s->own_dofs.resize(s->dof_ptr[nloc]);
  s->rel.resize(s->rel_ptr[nloc]);
  s->nnz = A->rowptr[n];
  auto t_3 = tnow();
  s->dest.assign(s->nnz, -1);
  // local indices, child -> parent maps, assembly destinations
  auto find_entry = [&](int32_t row, int32_t colv) -> int64_t {
    const int32_t* b = A->col + A->rowptr[row];
    const int32_t* e = A->col + A->rowptr[row + 1];
    const int32_t* it = std::lower_bound(b, e, colv);
    return (it != e && *it == colv) ? (int64_t)(it - A->col) : -1;
  };
  // One front at a time needs the local index of every dof it holds: `loc` / `loc_owner`, indexed by dof.  The fronts are
  // independent (every output range - own_dofs, rel, the destinations of the entries of its own rows and of the mirrored entries -
  // belongs to one front), so worker threads take them in chunks, each with its private pair of index arrays.
  std::vector<int> work;
  work.reserve(post.size());
  for (int t : post)
    if (state[t] == 1) work.push_back(t);  // ghosts are neither assembled nor eliminated here; their maps into the parent are built by the parent
  const char* sym_err = nullptr;
  std::mutex err_mu;
  std::atomic<int64_t> next{0};
  auto worker = [&]() {
    std::vector<int32_t> loc(n, -1), loc_owner(n, -1);
    for (;;) {
      const int64_t w0 = next.fetch_add(256);
      if (w0 >= (int64_t)work.size()) break;
      for (int64_t wi = w0; wi < std::min<int64_t>(w0 + 256, (int64_t)work.size()); ++wi) {
        const int t = work[wi];
        const int f = slot_of[t];
        const NdLevel& Lv = s->lev[tbatch[t]];
        const int64_t M = Lv.P + Lv.B;
        int k = 0;
        int64_t w = s->dof_ptr[f];
        for (int32_t g : T[t].own)
          for (int64_t q = S.nd_ptr[g]; q < S.nd_ptr[g + 1]; ++q) {
            int32_t i = S.nd_dofs[q];
            loc[i] = k++;
            loc_owner[i] = t;
            s->own_dofs[w++] = i;
          }
        k = Lv.P;
        for (int32_t g : T[t].border)
          for (int64_t q = S.nd_ptr[g]; q < S.nd_ptr[g + 1]; ++q) {
            int32_t i = S.nd_dofs[q];
            loc[i] = k++;
            loc_owner[i] = t;
          }
        for (int c = 0; c < 2; ++c) {
          int ct = T[t].child[c];
          if (ct < 0 || slot_of[ct] < 0) continue;
          int64_t r = s->rel_ptr[slot_of[ct]];
          for (int32_t g : T[ct].border)
            for (int64_t q = S.nd_ptr[g]; q < S.nd_ptr[g + 1]; ++q) {
              int32_t i = S.nd_dofs[q];
              if (loc_owner[i] != t) {
                std::lock_guard<std::mutex> lk(err_mu);
                sym_err = "symbolic inconsistency: child border not contained in the parent front";
                return;
              }
              s->rel[r++] = loc[i];
            }
        }
        for (int32_t g : T[t].own)
          for (int64_t q = S.nd_ptr[g]; q < S.nd_ptr[g + 1]; ++q) {
            const int32_t i = S.nd_dofs[q];
            const int64_t r = loc[i];
            for (int32_t e = A->rowptr[i]; e < A->rowptr[i + 1]; ++e) {
              const int32_t j = A->col[e];
              const int tj = tnode[A->node_of_dof[j]];
              if (order_of[tj] < order_of[t]) continue;  // assembled from the other side (earlier front)
              if (loc_owner[j] != t) {
                std::lock_guard<std::mutex> lk(err_mu);
                sym_err = "symbolic inconsistency: coupled dof missing from the front";
                return;
              }
              const int64_t c = loc[j];
              s->dest[e] = s->fbase[f] + c * M + r;
              if (tj != t) {
                int64_t et = find_entry(j, i);
                if (et >= 0) s->dest[et] = s->fbase[f] + r * M + c;
              }
            }
          }
      }
    }
  };
  {
    // private index arrays cost 8 bytes per dof and thread: at most 8 threads, fewer on small problems
    int nthr = (int)std::min<int64_t>(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 8u),
                                     std::max<int64_t>(1, (int64_t)work.size() / 2048));
    if (const char* e = getenv("PGX_ND_THREADS")) nthr = std::max(1, atoi(e));
    std::vector<std::thread> pool;
    for (int i = 1; i < nthr; ++i) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
  }
  if (sym_err) {
    s->err = sym_err;
    return PGX_EINVAL;
  }
  if (dsize == 1)  // (on a distributed handle the entries of other ranks' fronts legitimately stay unassigned)
    for (int64_t e = 0; e < s->nnz; ++e)
      if (s->dest[e] < 0) {
        s->err = "matrix pattern is not structurally symmetric (or columns are not sorted)";
        return PGX_EINVAL;
      }
  if (ptime) {
    auto t_4 = tnow();
    fprintf(stderr, "pgx_nd symbolic: node graph %.0f ms, dissection %.0f ms, fronts / batches / layout %.0f ms, maps + destinations %.0f ms\n",
            tms(t_0, t_1), tms(t_1, t_2), tms(t_2, t_3), tms(t_3, t_4));
  }
  return PGX_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// device kernels
// ------------------------------------------------------------------------------------------------------------------
__global__ void k_nd_scatter(int64_t t0, int64_t t1, const int64_t* __restrict__ sdest, const int32_t* __restrict__ ssrc,
                             const double* __restrict__ vals, double* __restrict__ arena) {
  int64_t t = t0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; t < t1; t += stride) arena[sdest[t]] = vals[ssrc[t]];
}

// identity on the padded part of every pivot block of the fronts [f0, f0 + gridDim.x)
__global__ void k_nd_pad(int64_t f0, const int32_t* __restrict__ fp, const int32_t* __restrict__ fP,
                         const int32_t* __restrict__ fM, const int64_t* __restrict__ fbase, double* __restrict__ arena) {
  const int64_t f = f0 + blockIdx.x;
  const int64_t M = fM[f];
  double* F = arena + fbase[f];
  for (int k = fp[f] + threadIdx.x; k < fP[f]; k += blockDim.x) F[k * M + k] = 1.0;
}

// extend-add of the Schur complements of the fronts [c0, c0+nc) (one level) whose child slot equals `pass`
__global__ void k_nd_extend_add(int64_t c0, int pass, int Pc, int Mc, const int32_t* __restrict__ fb,
                                const int32_t* __restrict__ slot01, const int32_t* __restrict__ parent,
                                const int32_t* __restrict__ fM, const int64_t* __restrict__ fbase,
                                const int64_t* __restrict__ rel_ptr, const int32_t* __restrict__ rel,
                                double* __restrict__ arena, int sym) {
  const int64_t f = c0 + blockIdx.x;
  if (slot01[f] != pass) return;
  const int b = fb[f];
  const int32_t* R = rel + rel_ptr[f];
  const double* src = arena + fbase[f] + (int64_t)Pc * Mc + Pc;
  const int pf = parent[f];
  if (pf < 0) return;  // distributed: the parent of a subtree root lives on rank 0
  const int64_t Mp = fM[pf];
  double* dst = arena + fbase[pf];
  // 32-bit index arithmetic (b < 65536: a border of that size would be a 34 GB front)
  const unsigned ub = (unsigned)b, total = ub * ub, step = gridDim.y * blockDim.x;
  for (unsigned idx = blockIdx.y * blockDim.x + threadIdx.x; idx < total; idx += step) {
    const unsigned c = idx / ub, r = idx - c * ub;
    // (symmetric mode: only the lower triangle of the child's block is valid - the parent gets both halves from it)
    dst[(int64_t)R[c] * Mp + R[r]] += (sym && r < c) ? src[(int64_t)r * Mc + c] : src[(int64_t)c * Mc + r];
  }
}

// parent-centric assembly of the fronts [f0, f0 + gridDim.x) of one batch: entry (r, c) = S0[inv0[r], inv0[c]] + S1[inv1[r], inv1[c]]
// with S0 / S1 the Schur blocks of the two children (in the other working buffer) - every entry of the M x M front is written,
// so the buffer needs no zero fill and no entry is read-modified-written; the matrix entries are ADDED afterwards (k_nd_scatter_add).
__global__ __launch_bounds__(256) void k_nd_gather(int64_t f0, int M, int P, const int32_t* __restrict__ child0,
                                                   const int32_t* __restrict__ child1, const int32_t* __restrict__ fM,
                                                   const int32_t* __restrict__ fP, const int64_t* __restrict__ fbase,
                                                   const int64_t* __restrict__ vbase, const int32_t* __restrict__ inv0,
                                                   const int32_t* __restrict__ inv1, double* __restrict__ arena, int sym) {
  const int64_t f = f0 + blockIdx.x;
  const int c0 = child0[f], c1 = child1[f];
  double* F = arena + fbase[f];
  const int32_t* I0 = inv0 + vbase[f];
  const int32_t* I1 = inv1 + vbase[f];
  int M0 = 0, M1 = 0;
  const double *S0 = nullptr, *S1 = nullptr;
  if (c0 >= 0) M0 = fM[c0], S0 = arena + fbase[c0] + (int64_t)fP[c0] * M0 + fP[c0];
  if (c1 >= 0) M1 = fM[c1], S1 = arena + fbase[c1] + (int64_t)fP[c1] * M1 + fP[c1];
  // the FRAME of the front only: the P pivot columns whole, then the pivot rows of the B border columns; the border block is
  // assembled by the Schur update itself (GATHER variants of the GEMM kernels)
  // (symmetric mode: the pivot columns only - the pivot rows right of the pivot block are never read, U12 = D L21^T - and a child's
  // entry (row, col) from (max, min): only the lower triangles of Schur blocks are valid)
  const unsigned uM = (unsigned)M, uP = (unsigned)P, mp = uM * uP, total = sym ? mp : mp + uP * (uM - uP), step = gridDim.y * blockDim.x;
  for (unsigned e = blockIdx.y * blockDim.x + threadIdx.x; e < total; e += step) {
    unsigned c, r;
    if (e < mp) {
      c = e / uM, r = e - c * uM;
    } else {
      const unsigned q = e - mp;
      c = uP + q / uP, r = q - (q / uP) * uP;
    }
    const unsigned idx = c * uM + r;
    if (sym && r + 64 <= c) continue;  // above every diagonal block (<= 64 pivots wide): the pivot block's upper part has no reader
    double v = 0.0;
    if (S0) {
      int a = I0[c], b = I0[r];
      if (sym && b < a) {
        const int t = a;
        a = b, b = t;
      }
      if ((a | b) >= 0) v = S0[(int64_t)a * M0 + b];
    }
    if (S1) {
      int a = I1[c], b = I1[r];
      if (sym && b < a) {
        const int t = a;
        a = b, b = t;
      }
      if ((a | b) >= 0) v += S1[(int64_t)a * M1 + b];
    }
    F[idx] = v;
  }
}

__global__ void k_nd_scatter_add(int64_t t0, int64_t t1, const int64_t* __restrict__ sdest, const int32_t* __restrict__ ssrc,
                                 const double* __restrict__ vals, double* __restrict__ arena) {
  int64_t t = t0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; t < t1; t += stride) arena[sdest[t]] += vals[ssrc[t]];  // one matrix entry per position: no atomics
}

// Schur block of one front <-> contiguous B x B buffer (exchange between ranks)
__global__ void k_nd_pack(const double* __restrict__ F, int M, int P, int B, double* __restrict__ buf, int unpack) {
  const int64_t total = (int64_t)B * B;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx / B), r = (int)(idx - (int64_t)c * B);
    double* e = const_cast<double*>(F) + (int64_t)(P + c) * M + P + r;
    if (unpack)
      *e = buf[idx];
    else
      buf[idx] = *e;
  }
}

// forward assembly of the per-front vectors of one level: own part from the right-hand side, border part from the children
__global__ void k_nd_fwd_assemble(int64_t f0, int P, int M, const int32_t* __restrict__ fp, const int32_t* __restrict__ fb,
                                  const int32_t* __restrict__ child0, const int32_t* __restrict__ child1,
                                  const int32_t* __restrict__ fP, const int64_t* __restrict__ vbase,
                                  const int64_t* __restrict__ dof_ptr, const int32_t* __restrict__ own_dofs,
                                  const int64_t* __restrict__ rel_ptr, const int32_t* __restrict__ rel,
                                  const double* __restrict__ b, double* __restrict__ vec) {
  const int64_t f = f0 + blockIdx.x;
  double* w = vec + vbase[f];
  const int p = fp[f];
  const int32_t* od = own_dofs + dof_ptr[f];
  for (int k = threadIdx.x; k < M; k += blockDim.x) w[k] = k < p ? b[od[k]] : 0.0;
  for (int pass = 0; pass < 2; ++pass) {
    const int c = pass == 0 ? child0[f] : child1[f];
    __syncthreads();
    if (c < 0) continue;
    const double* wc = vec + vbase[c] + fP[c];
    const int32_t* R = rel + rel_ptr[c];
    const int bc = fb[c];
    for (int k = threadIdx.x; k < bc; k += blockDim.x) w[R[k]] += wc[k];
  }
}

// backward: border values from the parent's vector
__global__ void k_nd_bwd_gather(int64_t f0, int P, const int32_t* __restrict__ fb, const int32_t* __restrict__ parent,
                                const int64_t* __restrict__ vbase, const int64_t* __restrict__ rel_ptr,
                                const int32_t* __restrict__ rel, double* __restrict__ vec) {
  const int64_t f = f0 + blockIdx.x;
  if (parent[f] < 0) return;  // subtree root of a distributed factorisation: its border values arrive from rank 0
  double* w = vec + vbase[f] + P;
  const double* wp = vec + vbase[parent[f]];
  const int32_t* R = rel + rel_ptr[f];
  const int b = fb[f];
  for (int k = threadIdx.x; k < b; k += blockDim.x) w[k] = wp[R[k]];
}

__global__ void k_nd_write_x(int64_t nfronts, const int32_t* __restrict__ fp, const int64_t* __restrict__ vbase,
                             const int64_t* __restrict__ dof_ptr, const int32_t* __restrict__ own_dofs,
                             const double* __restrict__ vec, double* __restrict__ x) {
  const int64_t f = blockIdx.x;
  if (f >= nfronts) return;
  const double* w = vec + vbase[f];
  const int32_t* od = own_dofs + dof_ptr[f];
  const int p = fp[f];
  for (int k = threadIdx.x; k < p; k += blockDim.x) x[od[k]] = w[k];
}


// ---- dense kernels, batched over the fronts of one level (blockIdx.x = front within the level) ---------------------
#include "pgx_nd_gemm.h"

// LU without pivoting of the nb x nb diagonal block at (kb,kb) of every front of the level (nd_diag_lu, pgx_nd_gemm.h).
__global__ __launch_bounds__(256) void k_nd_diag(double* __restrict__ arena, int64_t lev_off, int M, int kb, int nb,
                                                 int* __restrict__ info, int64_t store_off, int P) {
  __shared__ double D[ND_NB][ND_NB + 1];
  const double* F = arena + lev_off + (int64_t)blockIdx.x * M * M + (int64_t)kb * M + kb;
  double* S = arena + store_off + (int64_t)blockIdx.x * ((int64_t)M * P + (int64_t)P * (M - P)) + (int64_t)kb * M + kb;
  nd_diag_lu(D, F, S, M, nb, info);
}

// panel solves against the factored diagonal block: chunk c < nch : columns [o0, o0+64) of the row panel, X <- L^{-1} X;
// chunk >= nch: rows [o0, o0+64) of the column panel, X <- X U^{-1}.  Blocked by 8 pivots in LDS: the 8x8 triangle by
// one thread per column/row (registers), the rank-8 update of the rest by all 256 threads; the entries of the diagonal
// block are packed so that the 8 coefficients a thread needs are contiguous.  2 barriers per 8 pivots.
__global__ __launch_bounds__(256) void k_nd_panel(double* __restrict__ arena, int64_t lev_off, int M, int kb, int nb,
                                                  int64_t store_off, int P) {
  __shared__ double T[ND_NB * (ND_NB + 1) / 2];
  __shared__ double X[ND_NB][ND_TS + 1];
  double* F = arena + lev_off + (int64_t)blockIdx.x * M * M;
  // the solved panels are final factor entries: they go to the compact store [M x P, ld M][P x B, ld P] of the front, which
  // is also where the trailing updates (k_nd_gemm) read their operands; the working matrix keeps the unsolved values
  const int64_t MP = (int64_t)M * P;
  double* S = arena + store_off + (int64_t)blockIdx.x * (MP + (int64_t)P * (M - P));
  const int tid = threadIdx.x;
  const int R = M - kb - nb, nch = (R + ND_TS - 1) / ND_TS;
  const bool isL = (int)blockIdx.y >= nch;
  const int o0 = kb + nb + ND_TS * (isL ? (int)blockIdx.y - nch : (int)blockIdx.y);
  const int wd = min(ND_TS, M - o0);
  const double* Dg = S + (int64_t)kb * M + kb;
  const int j = tid & 63, g = tid >> 6;
  if (!isL) {
    // T: strict lower triangle, row-major packed: L[r][c] at r(r-1)/2 + c
    {  // all loads in flight before the first LDS write
      double v[16], u[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int idx = tid + 256 * q, r = idx % nb, c = idx / nb;
        v[q] = (idx < nb * nb && r > c) ? Dg[(int64_t)c * M + r] : 0.0;
        u[q] = (idx < nb * ND_TS && c < wd) ? F[(int64_t)(o0 + c) * M + kb + r] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int idx = tid + 256 * q, r = idx % nb, c = idx / nb;
        if (idx < nb * nb && r > c) T[r * (r - 1) / 2 + c] = v[q];
        if (idx < nb * ND_TS) X[r][c] = u[q];
      }
    }
    __syncthreads();
    for (int jb = 0; jb < nb; jb += 8) {
      const int w = min(8, nb - jb);
      double xs[8];
      if (g == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (i < w) {
            double v = X[jb + i][j];
            const double* Lr = T + (jb + i) * (jb + i - 1) / 2 + jb;
#pragma unroll
            for (int m = 0; m < 8; ++m)
              if (m < i) v -= Lr[m] * xs[m];
            xs[i] = v;
            X[jb + i][j] = v;
          }
        }
      }
      __syncthreads();
      if (jb + w < nb) {
#pragma unroll
        for (int m = 0; m < 8; ++m) xs[m] = m < w ? X[jb + m][j] : 0.0;
        // four rows per trip, every LDS read issued before the first use (a row-at-a-time loop with a run-time trip
        // count serialises read -> 8 fma -> write, one LDS latency per row)
        for (int r = jb + w + g; r < nb; r += 16) {
          double v[4], lc[4][8];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int rr = min(r + 4 * q, nb - 1);
            const double* Lr = T + rr * (rr - 1) / 2 + jb;
            v[q] = X[rr][j];
#pragma unroll
            for (int m = 0; m < 8; ++m) lc[q][m] = Lr[m];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int m = 0; m < 8; ++m)
              if (m < w) v[q] -= lc[q][m] * xs[m];
            if (r + 4 * q < nb) X[r + 4 * q][j] = v[q];
          }
        }
        __syncthreads();
      }
    }
    for (int idx = tid; idx < nb * ND_TS; idx += 256) {
      const int k = idx % nb, jj = idx / nb;
      if (jj < wd) {
        const int c = o0 + jj;
        S[c < P ? (int64_t)c * M + kb + k : MP + (int64_t)(c - P) * P + kb + k] = X[k][jj];
      }
    }
  } else {
    // T: upper triangle incl. diagonal, column-major packed: U[k][c] at c(c+1)/2 + k
    {  // all loads in flight before the first LDS write
      double v[16], u[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int idx = tid + 256 * q, k = idx % nb, c = idx / nb, i = idx % ND_TS, k2 = idx / ND_TS;
        v[q] = (idx < nb * nb && k <= c) ? Dg[(int64_t)c * M + k] : 0.0;
        u[q] = (idx < nb * ND_TS && i < wd) ? F[(int64_t)(kb + k2) * M + o0 + i] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int idx = tid + 256 * q, k = idx % nb, c = idx / nb, i = idx % ND_TS, k2 = idx / ND_TS;
        if (idx < nb * nb && k <= c) T[c * (c + 1) / 2 + k] = v[q];
        if (idx < nb * ND_TS) X[k2][i] = u[q];
      }
    }
    __syncthreads();
    for (int jb = 0; jb < nb; jb += 8) {
      const int w = min(8, nb - jb);
      double xs[8];
      if (g == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (i < w) {
            const int c = jb + i;
            const double* Uc = T + c * (c + 1) / 2 + jb;
            double v = X[c][j];
#pragma unroll
            for (int m = 0; m < 8; ++m)
              if (m < i) v -= xs[m] * Uc[m];
            v /= Uc[i];
            xs[i] = v;
            X[c][j] = v;
          }
        }
      }
      __syncthreads();
      if (jb + w < nb) {
#pragma unroll
        for (int m = 0; m < 8; ++m) xs[m] = m < w ? X[jb + m][j] : 0.0;
        for (int c = jb + w + g; c < nb; c += 16) {  // four columns per trip, reads first (see the row panel)
          double v[4], uc[4][8];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int cc = min(c + 4 * q, nb - 1);
            const double* Uc = T + cc * (cc + 1) / 2 + jb;
            v[q] = X[cc][j];
#pragma unroll
            for (int m = 0; m < 8; ++m) uc[q][m] = Uc[m];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int m = 0; m < 8; ++m)
              if (m < w) v[q] -= xs[m] * uc[q][m];
            if (c + 4 * q < nb) X[c + 4 * q][j] = v[q];
          }
        }
        __syncthreads();
      }
    }
    for (int idx = tid; idx < nb * ND_TS; idx += 256) {
      const int ii = idx % ND_TS, k = idx / ND_TS;
      if (ii < wd) {
        S[(int64_t)(kb + k) * M + o0 + ii] = X[k][ii];
      }
    }
  }
}

// Register variant of the panel solves (PGX_ND_PANEL=2; an experiment kept for the comparison in DESIGN.md section 9).  ONE WAVE per 64-wide
// chunk, the lane's column (row panel) / row (column panel) of the chunk in registers, right-looking substitution
//     for m < nb:  x[m] final (* 1/U[m][m] for the column panel);  x[i] -= C[m][i] x[m]  for i > m
// fully unrolled: the nb - m - 1 updates of a step are independent FMAs, their coefficients are WAVE-UNIFORM LDS reads at
// compile-time offsets (ds_read2_b64 broadcasts, no conflicts) and there is NO barrier inside the solve - the LDS kernel
// pays 2 barriers per 8 pivots with one of its four waves doing the 8x8 triangles.  Work per chunk is the nb^2/2 x 64 FMAs
// of the substitution itself.  A workgroup = 4 chunks of ONE panel type sharing the coefficient block C (zero padded to
// NB so that the unrolled updates past nb are no-ops).  The row panel is column-major along the pivots (a lane's column is
// contiguous in memory), so its chunks pass through a 16-row LDS slab per wave in both directions: global accesses stay
// 128-B segments, the slab (stride ND_PS = 66) is conflict-free on both sides.
#define ND_PS 66
template <int NB>
__global__ __launch_bounds__(256) void k_nd_panel_r(double* __restrict__ arena, int64_t lev_off, int M, int kb, int nb,
                                                    int64_t store_off, int P) {
  __shared__ __attribute__((aligned(16))) double C[NB * ND_PS];
  __shared__ double inv[64];
  __shared__ double Tb[4][16 * ND_PS];
  double* F = arena + lev_off + (int64_t)blockIdx.x * M * M;
  const int64_t MP = (int64_t)M * P;
  double* S = arena + store_off + (int64_t)blockIdx.x * (MP + (int64_t)P * (M - P));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = (int)gridDim.y >> 1;
  const bool isL = (int)blockIdx.y >= half;
  const int t = ((int)blockIdx.y - (isL ? half : 0)) * 4 + wave;
  const int o0 = kb + nb + ND_TS * t;
  const int wd = min(ND_TS, M - o0);  // <= 0 for the idle waves of the last workgroup
  const double* Dg = S + (int64_t)kb * M + kb;
  // coefficient block: C[m][i] multiplies x[m] in equation i > m.  Row panel: L[i][m]; column panel: U[m][i].
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int a = idx % NB, b = idx / NB;  // a runs along memory in both cases
    if (!isL) {
      C[b * ND_PS + a] = (a < nb && b < nb && a > b) ? Dg[(int64_t)b * M + a] : 0.0;  // m = b, i = a
    } else {
      C[a * ND_PS + b] = (a < nb && b < nb && b > a) ? Dg[(int64_t)b * M + a] : 0.0;  // m = a, i = b
    }
  }
  if (tid < 64) inv[tid] = (isL && tid < nb) ? 1.0 / Dg[(int64_t)tid * M + tid] : 1.0;
  double x[NB];
  double* tb = Tb[wave];
  const int rl = lane & 15, cg = lane >> 4;
  if (wd > 0) {
    if (isL) {
#pragma unroll
      for (int k = 0; k < NB; ++k) x[k] = (k < nb && lane < wd) ? F[(int64_t)(kb + k) * M + o0 + lane] : 0.0;
    } else {
#pragma unroll
      for (int s = 0; s < NB / 16; ++s) {
        double v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int col = 4 * j + cg, r = 16 * s + rl;
          v[j] = (16 * s < nb && r < nb && col < wd) ? F[(int64_t)(o0 + col) * M + kb + r] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) tb[rl * ND_PS + 4 * j + cg] = v[j];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) x[16 * s + rr] = tb[rr * ND_PS + lane];
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  __syncthreads();
  if (wd <= 0) return;
#pragma unroll
  for (int m = 0; m < NB; ++m) {
    if ((m & 7) == 0 && m >= nb) break;
    const double xm = x[m] * inv[m];  // inv = 1 for the row panel (unit lower triangle)
    x[m] = xm;
#pragma unroll
    for (int i = m + 1; i < NB; ++i) x[i] -= C[m * ND_PS + i] * xm;
  }
  if (isL) {
#pragma unroll
    for (int k = 0; k < NB; ++k)
      if (k < nb && lane < wd) S[(int64_t)(kb + k) * M + o0 + lane] = x[k];
  } else {
#pragma unroll
    for (int s = 0; s < NB / 16; ++s) {
      if (16 * s >= nb) break;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) tb[rr * ND_PS + lane] = x[16 * s + rr];
      __builtin_amdgcn_wave_barrier();
      double v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = tb[rl * ND_PS + 4 * j + cg];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int col = 4 * j + cg, r = 16 * s + rl, c = o0 + col;
        if (r < nb && col < wd) S[c < P ? (int64_t)c * M + kb + r : MP + (int64_t)(c - P) * P + kb + r] = v[j];
      }
    }
  }
}

// MFMA variant of the panel solves (default).  The substitution is blocked by 16 pivots,
//     X_a = inv(T_aa) (B_a - sum_{b<a} T_ab X_b),   a = 0..3,
// every product a chain of v_mfma_f64_16x16x4_f64 with the 16 x 16 coefficient block as the A operand (one ds_read_b64 per
// MFMA = 512 B of LDS per 2048 flops; the scalar substitutions above need 512 B per 128 flops and are LDS-bound on it) and
// the chunk's pivot block as B operand AND accumulator: for this instruction lane l holds B[k = l>>4][n = l&15] and
// D[m = (l>>4) + 4 reg][n = l&15], so accumulator register s of block b IS the B operand of k-slice s (pivots 16b + 4s +
// (l>>4)) and a solved block feeds the next products straight from registers - no LDS round trip, no barrier in the solve.
// The column panel X U^{-1} is the same recurrence on X^T with T = U^T, so one code path serves both with the pivot / line
// strides swapped.  T is staged NEGATED below the diagonal (the MFMAs accumulate), its four diagonal blocks are inverted
// in place, one per wave (16 lanes = 16 columns of the inverse by substitution, coefficients broadcast from LDS), and the
// padding beyond nb is the identity.  A workgroup = one 64-line chunk, wave w = its lines [16w, 16w+16).
// LDS layouts (r = equation, m = unknown): column panel r*68 + m, row panel m*80 + r - both fill from memory along lanes
// and give the A-operand reads (r = l&15, m = 4s + (l>>4)) the minimal two passes.
__global__ __launch_bounds__(256) void k_nd_panel_m(double* __restrict__ arena, int64_t lev_off, int M, int kb, int nb,
                                                    int64_t store_off, int P, int sym) {
  __shared__ double Lh[64 * 80];
  double* F = arena + lev_off + (int64_t)blockIdx.x * M * M;
  const int64_t MP = (int64_t)M * P;
  double* S = arena + store_off + (int64_t)blockIdx.x * (MP + (int64_t)P * (M - P));
  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6, n = l & 15, g = l >> 4;
  const int R = M - kb - nb, nch = (R + ND_TS - 1) / ND_TS;
  // symmetric mode: the grid holds the nch chunks of the COLUMN panel only; the row panel is its scaled transpose (written below)
  const bool isL = sym || (int)blockIdx.y >= nch;
  const int o0 = kb + nb + ND_TS * (sym ? (int)blockIdx.y : (isL ? (int)blockIdx.y - nch : (int)blockIdx.y));
  const int wd = min(ND_TS, M - o0);
  const double* Dg = S + (int64_t)kb * M + kb;
  const int sr = isL ? 68 : 1, sm = isL ? 1 : 80;
  const int nbr = (nb + 15) & ~15;
  {
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int idx = tid + 256 * q, a = idx & 63, b = idx >> 6;  // a runs along memory
      const bool in = a < nb && b < nb;
      const bool off = isL ? a < b : a > b;  // (r, m) = (b, a) for the column panel, (a, b) for the row panel
      v[q] = (in && (off || (isL && a == b))) ? Dg[(int64_t)b * M + a] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int idx = tid + 256 * q, a = idx & 63, b = idx >> 6;
      const bool in = a < nb && b < nb;
      const double w = a == b ? ((isL && in) ? v[q] : 1.0) : -v[q];
      if (4 * q < nbr) Lh[isL ? b * 68 + a : b * 80 + a] = w;  // b = 4q + wave: only the rows / columns of the blocks in use
    }
  }
  // the chunk in accumulator layout: block a, register q <-> pivot 16a + (l>>4) + 4q, line 16 wave + (l&15)
  nd_v4d acc[4];
  const int line = 16 * wave + n;
  const int64_t lbase = isL ? (int64_t)kb * M + o0 + line : (int64_t)(o0 + line) * M + kb;
  const int64_t pstr = isL ? M : 1;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = 16 * a + g + 4 * q;
      acc[a][q] = (p < nb && line < wd) ? F[lbase + p * pstr] : 0.0;
    }
  __syncthreads();
  if (16 * wave < nb) {  // inverse of diagonal block `wave`: lane n solves T y = e_n
    const int a0 = 16 * wave;
    double y[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = r == n ? 1.0 : 0.0;
    const double rd = isL ? 1.0 / Lh[(a0 + n) * (sr + sm)] : 1.0;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (isL) y[m] *= nd_bcast(rd, m);
#pragma unroll
      for (int r = m + 1; r < 16; ++r) y[r] += Lh[(a0 + r) * sr + (a0 + m) * sm] * y[m];
    }
    __builtin_amdgcn_wave_barrier();
    if (g == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) Lh[(a0 + r) * sr + (a0 + n) * sm] = y[r];
    }
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    if (16 * a >= nb) break;
    const double* La = Lh + (16 * a + n) * sr + g * sm;
#pragma unroll
    for (int b = 0; b < a; ++b)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(La[(16 * b + 4 * s) * sm], acc[b][s], acc[a], 0, 0, 0);
    const nd_v4d t = acc[a];
    nd_v4d x = (nd_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) x = __builtin_amdgcn_mfma_f64_16x16x4f64(La[(16 * a + 4 * s) * sm], t[s], x, 0, 0, 0);
    acc[a] = x;
  }
  if (line < wd) {
    const int c = o0 + line;
    const int64_t sbase = isL ? lbase : (c < P ? (int64_t)c * M + kb : MP + (int64_t)(c - P) * P + kb);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int p = 16 * a + g + 4 * q;
        if (p < nb) S[sbase + p * pstr] = acc[a][q];
      }
    if (sym) {  // U12 = D11 L21^T: entry (pivot kb + p, column c) = u_pp * L(c, kb + p), at the row panel's place in the store
      const int64_t ubase = c < P ? (int64_t)c * M + kb : MP + (int64_t)(c - P) * P + kb;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int p = 16 * a + g + 4 * q;
          if (p < nb) S[ubase + p] = Dg[(int64_t)p * M + p] * acc[a][q];
        }
    }
  }
}

static void nd_launch_panel(int kind, hipStream_t q, unsigned count, unsigned nch, double* arena, int64_t woff, int M, int kb,
                            int nb, int64_t poff, int P, int sym = 0) {
  if (kind == 0) {
    hipLaunchKernelGGL(k_nd_panel_m, dim3(count, sym ? nch : 2 * nch), dim3(256), 0, q, arena, woff, M, kb, nb, poff, P, sym);
    return;
  }
  if (kind == 1) {
    hipLaunchKernelGGL(k_nd_panel, dim3(count, 2 * nch), dim3(256), 0, q, arena, woff, M, kb, nb, poff, P);
    return;
  }
  const dim3 grid(count, 2 * ((nch + 3) / 4));
  if (nb <= 16)
    hipLaunchKernelGGL(k_nd_panel_r<16>, grid, dim3(256), 0, q, arena, woff, M, kb, nb, poff, P);
  else if (nb <= 32)
    hipLaunchKernelGGL(k_nd_panel_r<32>, grid, dim3(256), 0, q, arena, woff, M, kb, nb, poff, P);
  else if (nb <= 48)
    hipLaunchKernelGGL(k_nd_panel_r<48>, grid, dim3(256), 0, q, arena, woff, M, kb, nb, poff, P);
  else
    hipLaunchKernelGGL(k_nd_panel_r<64>, grid, dim3(256), 0, q, arena, woff, M, kb, nb, poff, P);
}

// in-place triangular solve of the diagonal range [k0,k1) of every front's pivot block on w: upper == 0: unit lower L11;
// upper != 0: U11.  One workgroup per front, 64-wide blocks: the 64x64 triangle is solved by wave 0 with lane shuffles,
// the remaining rows OF THE RANGE are updated by all threads; rows outside the range are left to k_nd_gemv (many
// workgroups), so that the big fronts near the root do not stream their factors through a single CU.
__global__ __launch_bounds__(256) void k_nd_trsv(const double* __restrict__ arena, int64_t lev_off, int64_t fs,
                                                 double* __restrict__ vec, int64_t voff, int M, int k0, int k1, int upper) {
  __shared__ double Ds[64][65];
  __shared__ double ys[64];
  __shared__ double red[4][64];
  const double* F = arena + lev_off + (int64_t)blockIdx.x * fs;  // compact store: the first P columns keep leading dimension M
  double* w = vec + voff + (int64_t)blockIdx.x * M;
  const int tid = threadIdx.x;
  const int nblk = (k1 - k0 + 63) / 64;
  for (int bb = 0; bb < nblk; ++bb) {
    const int kb = k0 + (upper ? (nblk - 1 - bb) * 64 : bb * 64);
    const int nb = min(64, k1 - kb);
    {  // all loads in flight before the first LDS write (a load -> wait -> write loop costs 16 memory latencies)
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int idx = tid + 256 * q;
        v[q] = idx < nb * nb ? F[(int64_t)(kb + idx / nb) * M + kb + idx % nb] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int idx = tid + 256 * q;
        if (idx < nb * nb) Ds[idx % nb][idx / nb] = v[q];
      }
    }
    __syncthreads();
    if (tid < 64) {
      const int r = tid;
      double y = r < nb ? w[kb + r] : 0.0;
      // the lane's row of the triangle comes 16 coefficients at a time, read before the 16 dependent steps that use
      // them (one LDS latency per 16 steps instead of one per step: this loop is the serial core of the solve phase)
      if (!upper) {
        for (int kc = 0; kc < nb; kc += 16) {
          double d[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) d[q] = Ds[r][min(kc + q, 63)];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int k = kc + q;
            if (k < nb) {  // uniform
              const double yk = nd_bcast(y, k);
              if (r > k && r < nb) y -= d[q] * yk;
            }
          }
        }
      } else {
        const double dinv = r < nb ? 1.0 / Ds[r][r] : 0.0;
        for (int kc = ((nb - 1) / 16) * 16; kc >= 0; kc -= 16) {
          double d[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) d[q] = Ds[r][min(kc + q, 63)];
#pragma unroll
          for (int q = 15; q >= 0; --q) {
            const int k = kc + q;
            if (k < nb) {  // uniform
              const double xk = nd_bcast(y * dinv, k);
              if (r < k) y -= d[q] * xk;
              if (r == k) y = xk;
            }
          }
        }
      }
      ys[r] = y;
      if (r < nb) w[kb + r] = y;
    }
    __syncthreads();
    // rows of the slab outside the block: 64 rows x 4 column groups per pass (16 independent loads per thread instead of
    // a 64-long chain), partial sums combined through LDS
    const int lo = upper ? k0 : kb + nb, hi = upper ? kb : k1;
    for (int rbase = lo; rbase < hi; rbase += 64) {
      const int r = rbase + (tid & 63), cg = tid >> 6;
      double a = 0.0;
      if (r < hi) {
        const int kend = min(cg * 16 + 16, nb);
#pragma unroll 8
        for (int k = cg * 16; k < kend; ++k) a += F[(int64_t)(kb + k) * M + r] * ys[k];
      }
      red[cg][tid & 63] = a;
      __syncthreads();
      if (tid < 64 && r < hi) w[r] -= (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      __syncthreads();
    }
  }
}

// The same solve for the levels of FEW LARGE fronts (round 5; batches of <= ND_BIG_COUNT fronts): near the root the solve phase is a
// chain of slab launches - triangle, then everything outside it - and k_nd_trsv spends a full memory latency per 64-pivot block
// (triangle -> LDS -> wave solve -> update loads -> reduce: ~35 us per 256-pivot slab).  Here ONE workgroup of 16 waves holds the whole
// slab: waves 0-3 bring the four 64 x 64 triangles into LDS, waves 4-15 keep the six off-diagonal blocks in registers (two waves per
// block: lane = row, 32 columns each), every load goes out before anything is computed, and the four block steps then run out of LDS
// and registers (wave solve, partial products, combination: three barriers per step).
#define ND_BIG_COUNT 256
#define ND_BIG_LDS ((4 * 64 * 65 + 256 + 12 * 64) * sizeof(double))
__global__ __launch_bounds__(1024) void k_nd_trsv_big(const double* __restrict__ arena, int64_t lev_off, int64_t fs,
                                                      double* __restrict__ vec, int64_t voff, int M, int k0, int k1, int upper) {
  extern __shared__ double nd_sm[];
  double(*Ds)[64][65] = reinterpret_cast<double(*)[64][65]>(nd_sm);
  double* ys = nd_sm + 4 * 64 * 65;
  double(*red)[64] = reinterpret_cast<double(*)[64]>(ys + 256);
  const double* F = arena + lev_off + (int64_t)blockIdx.x * fs;
  double* w = vec + voff + (int64_t)blockIdx.x * M;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nbt = k1 - k0, nblk = (nbt + 63) / 64;
  // off-diagonal blocks (block row, block column) in the order their block column is processed
  const int prl[6] = {1, 2, 3, 2, 3, 3}, pcl[6] = {0, 0, 0, 1, 1, 2};
  const int pru[6] = {2, 1, 0, 1, 0, 0}, pcu[6] = {3, 3, 3, 2, 2, 1};
  double off[32];
  int br = -1, bc = -1, half = 0;
  if (wave >= 4) {
    const int p = (wave - 4) >> 1;
    half = (wave - 4) & 1;
    br = upper ? pru[p] : prl[p];
    bc = upper ? pcu[p] : pcl[p];
    const int r = k0 + 64 * br + lane;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const int c = k0 + 64 * bc + 32 * half + q;
      off[q] = (br < nblk && bc < nblk && r < k1 && c < k1) ? F[(int64_t)c * M + r] : 0.0;
    }
  } else if (wave < nblk) {
    const int kb = k0 + 64 * wave, nb = min(64, k1 - kb);
    for (int c0 = 0; c0 < 64; c0 += 16) {
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = (lane < nb && c0 + q < nb) ? F[(int64_t)(kb + c0 + q) * M + kb + lane] : 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) Ds[wave][lane][c0 + q] = v[q];
    }
  }
  if (tid < 256) ys[tid] = (k0 + tid < k1) ? w[k0 + tid] : 0.0;
  __syncthreads();
  for (int step = 0; step < nblk; ++step) {
    const int b = upper ? nblk - 1 - step : step;
    if (wave == 0) {
      const int r = lane, nb = min(64, nbt - 64 * b);
      double y = r < nb ? ys[64 * b + r] : 0.0;
      if (!upper) {
        for (int kc = 0; kc < nb; kc += 16) {
          double d[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) d[q] = Ds[b][r][min(kc + q, 63)];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int k = kc + q;
            if (k < nb) {  // uniform
              const double yk = nd_bcast(y, k);
              if (r > k && r < nb) y -= d[q] * yk;
            }
          }
        }
      } else {
        const double dinv = r < nb ? 1.0 / Ds[b][r][r] : 0.0;
        for (int kc = ((nb - 1) / 16) * 16; kc >= 0; kc -= 16) {
          double d[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) d[q] = Ds[b][r][min(kc + q, 63)];
#pragma unroll
          for (int q = 15; q >= 0; --q) {
            const int k = kc + q;
            if (k < nb) {  // uniform
              const double xk = nd_bcast(y * dinv, k);
              if (r < k) y -= d[q] * xk;
              if (r == k) y = xk;
            }
          }
        }
      }
      if (r < nb) ys[64 * b + r] = y;
    }
    __syncthreads();
    const bool mine = wave >= 4 && bc == b && br < nblk;  // this wave's block takes the block just solved
    if (mine) {
      const double* yb = ys + 64 * b + 32 * half;
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int q = 0; q < 32; q += 2) {
        a0 += off[q] * yb[q];
        a1 += off[q + 1] * yb[q + 1];
      }
      red[wave - 4][lane] = a0 + a1;
    }
    __syncthreads();
    if (mine && half == 0) ys[64 * br + lane] -= red[wave - 4][lane] + red[wave - 3][lane];
    __syncthreads();
  }
  if (tid < 256 && k0 + tid < k1) w[k0 + tid] = ys[tid];
}

// w[r0:r1) -= F[r0:r1, c0:c1) w[c0:c1).  A workgroup takes ND_GR = 64 rows; its 4 waves split the columns (wave g takes
// the columns c0 + g, c0 + g + 4, ...: 4x more workgroups and 4x shorter load chains than one thread per row with 256 rows
// per block - the mid and top levels of the tree have few fronts and were latency-bound), partial sums meet in LDS.
// Column c of the operand starts at F + cbase + (c - c0) * ld (compact store: ld = M inside the first P columns, the U12
// block has its own base and ld = P).
#define ND_GR 64
#define ND_GW 4  // waves per workgroup = column groups (8 waves, 16 loads per batch: measured SLOWER, ex 06 1024^2 solves 10.5 -> 12.2 ms)
__global__ __launch_bounds__(64 * ND_GW) void k_nd_gemv(const double* __restrict__ arena, int64_t lev_off, int64_t fs,
                                                        double* __restrict__ vec, int64_t voff, int M, int r0, int r1, int c0, int c1,
                                                        int64_t cbase, int ld) {
  __shared__ double xs[256];
  __shared__ double red[ND_GW][ND_GR];
  const double* F = arena + lev_off + (int64_t)blockIdx.x * fs + cbase;
  double* w = vec + voff + (int64_t)blockIdx.x * M;
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int r = r0 + blockIdx.y * ND_GR + lane;
  double a = 0.0;
  for (int k0 = c0; k0 < c1; k0 += 256) {
    const int kn = min(256, c1 - k0);
    __syncthreads();
    if ((int)threadIdx.x < kn) xs[threadIdx.x] = w[k0 + threadIdx.x];
    __syncthreads();
    if (r < r1) {
      const double* col = F + (int64_t)(k0 - c0) * ld + r;
#pragma unroll 8
      for (int k = g; k < kn; k += ND_GW) a += col[(int64_t)k * ld] * xs[k];
    }
  }
  red[g][lane] = a;
  __syncthreads();
  if (g == 0 && r < r1) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < ND_GW; ++q) t += red[q][lane];
    w[r] -= t;
  }
}

// ---- solve phase of SMALL fronts (round 5): P <= 64, M <= 256 - ONE WAVE per front does what k_nd_fwd_assemble + k_nd_trsv + k_nd_gemv
// (forward) resp. k_nd_bwd_gather + k_nd_gemv + k_nd_trsv (backward) did in three launches of 256-thread workgroups.  The deep levels
// of a 2-D tree are hundreds of thousands of fronts of 50-150 rows: a launch per step and four waves per front left them at 0.7-2.5
// TB/s (ex 06 at 1024^2: depth 17, 130 052 fronts of 5 + 48: 1.4 ms for 1 GB of factors).  Lane = row (rows lane + 64 i), the
// column of L / U for the next 8 pivots is requested before the 8 dependent steps, pivot values travel by v_readlane.
template <int NR>
__global__ __launch_bounds__(64) void k_nd_fwd_small(const double* __restrict__ arena, int64_t lev_off, int64_t fs, int64_t f0, int P,
                                                     int M, const int32_t* __restrict__ fp, const int32_t* __restrict__ fb,
                                                     const int32_t* __restrict__ child0, const int32_t* __restrict__ child1,
                                                     const int32_t* __restrict__ fP, const int64_t* __restrict__ vbase,
                                                     const int64_t* __restrict__ dof_ptr, const int32_t* __restrict__ own_dofs,
                                                     const int64_t* __restrict__ rel_ptr, const int32_t* __restrict__ rel,
                                                     const double* __restrict__ b, double* __restrict__ vec) {
  __shared__ double ws[64 * NR];
  const int64_t f = f0 + blockIdx.x;
  const int lane = threadIdx.x;
  const double* L = arena + lev_off + (int64_t)blockIdx.x * fs;
  double* w = vec + vbase[f];
  const int p = fp[f];
  const int32_t* od = own_dofs + dof_ptr[f];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = lane + 64 * i;
    ws[r] = r < p ? b[od[r]] : 0.0;
  }
  __syncthreads();
  for (int pass = 0; pass < 2; ++pass) {  // the children's border contributions (a child's map is injective: no conflicts)
    const int c = pass == 0 ? child0[f] : child1[f];
    if (c < 0) continue;
    const double* wc = vec + vbase[c] + fP[c];
    const int32_t* R = rel + rel_ptr[c];
    const int bc = fb[c];
    for (int k = lane; k < bc; k += 64) ws[R[k]] += wc[k];
    __syncthreads();
  }
  double wr[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) wr[i] = ws[lane + 64 * i];
  for (int k0 = 0; k0 < p; k0 += 8) {
    double col[8][NR];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = k0 + q;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int r = lane + 64 * i;
        col[q][i] = (k < p && r > k && r < M) ? L[(int64_t)k * M + r] : 0.0;
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = k0 + q;
      if (k < p) {  // uniform
        const double yk = nd_bcast(wr[0], k);
#pragma unroll
        for (int i = 0; i < NR; ++i) wr[i] -= col[q][i] * yk;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = lane + 64 * i;
    if (r < M) w[r] = wr[i];
  }
}

// backward sweep of a small front: border values from the parent's vector, y <- y - U12 x_border (lanes = (pivot row, column group),
// partial sums folded across the groups), then the U11 back substitution in registers.  gather == 0: the border values are already
// in the front's vector (subtree roots of a distributed factorisation).
template <int NR>
__global__ __launch_bounds__(64) void k_nd_bwd_small(const double* __restrict__ arena, int64_t lev_off, int64_t fs, int64_t f0, int P,
                                                     int M, int pplog, const int32_t* __restrict__ fp, const int32_t* __restrict__ fb,
                                                     const int32_t* __restrict__ parent, const int64_t* __restrict__ vbase,
                                                     const int64_t* __restrict__ rel_ptr, const int32_t* __restrict__ rel,
                                                     double* __restrict__ vec, int gather) {
  __shared__ double xb[64 * NR];
  const int64_t f = f0 + blockIdx.x;
  const int lane = threadIdx.x;
  const double* S = arena + lev_off + (int64_t)blockIdx.x * fs;
  double* w = vec + vbase[f];
  const int p = fp[f], b = fb[f], B = M - P;
  const int pf = parent[f];
  const bool take = gather && pf >= 0;
  const double* wp = take ? vec + vbase[pf] : nullptr;
  const int32_t* R = rel + rel_ptr[f];
  for (int j = lane; j < B; j += 64) {
    double v = 0.0;
    if (j < b) v = take ? wp[R[j]] : w[P + j];
    xb[j] = v;
    if (take && j < b) w[P + j] = v;  // the children of this front read it from here
  }
  __syncthreads();
  const int PP = 1 << pplog, G = 64 >> pplog;
  const int k = lane & (PP - 1), g = lane >> pplog;
  const double* U12 = S + (int64_t)M * P + k;
  double acc = 0.0;
  if (k < P) {
    int j = g;
    for (; j + 7 * G < b; j += 8 * G) {
      double u[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) u[q] = U12[(int64_t)(j + q * G) * P];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += u[q] * xb[j + q * G];
    }
    for (; j < b; j += G) acc += U12[(int64_t)j * P] * xb[j];
  }
  for (int off = PP; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
  double y = lane < P ? w[lane] - acc : 0.0;
  const double rd = lane < p ? 1.0 / S[(int64_t)lane * M + lane] : 1.0;
  for (int k0 = ((p - 1) / 8) * 8; k0 >= 0; k0 -= 8) {
    double col[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int kk = k0 + q;
      col[q] = (kk < p && lane < kk) ? S[(int64_t)kk * M + lane] : 0.0;
    }
#pragma unroll
    for (int q = 7; q >= 0; --q) {
      const int kk = k0 + q;
      if (kk < p) {  // uniform
        const double xk = nd_bcast(y * rd, kk);
        y -= col[q] * xk;
        if (lane == kk) y = xk;
      }
    }
  }
  if (lane < P) w[lane] = y;
}

// ------------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------------
#define NDHIP(call)                                                   \
  do {                                                                \
    hipError_t e_ = (call);                                           \
    if (e_ != hipSuccess) {                                           \
      s->err = std::string(#call) + ": " + hipGetErrorString(e_);     \
      return PGX_EHIP;                                                \
    }                                                                 \
  } while (0)

// ------------------------------------------------------------------------------------------------------------------
// Fused LEAF fronts (the deepest tree depth: no children).  The level-batched path moves a leaf's M x M working matrix
// seven times through HBM (zero fill, scatter, pad, diagonal block, two panel solves, Schur update) for P <= 32 pivots'
// worth of arithmetic: with hundreds of thousands of leaves that level alone was 11 % of a 2-D factorisation.  Here ONE
// WAVE owns a front: the matrix entries are assembled in an LDS tile, then every column is pulled into registers
// (lane = row; rows 64.. in a second register) and eliminated left-looking against the L columns found so far, which stay
// in registers (<= 32 pivot columns); the pivot-row entries travel by v_readlane.  A column is written once: L\U and L21
// to the compact store, U12 after them, the Schur block to the working matrix for the parent's extend-add.  HBM traffic:
// the entries in, the factors and the Schur block out.  P <= ND_LEAF_P, M <= 128; pivots as in k_nd_diag (no pivoting,
// static perturbation of zero pivots counted in info).
// ------------------------------------------------------------------------------------------------------------------
#define ND_LEAF_P 32
#define ND_LEAF_M 128
// CHILD: a front WITH children - its frame (same tile) is first gathered from the children's Schur blocks through the inverse maps,
// the matrix entries are added, and only L\\U, L21 and U12 are produced: the border block belongs to the GATHER Schur update.
template <bool TWO, bool CHILD>  // TWO: M > 64, a lane also owns row 64 + lane
__global__ __launch_bounds__(64) void k_nd_leaf(double* __restrict__ arena, int64_t lev_off, int64_t store_off, int64_t f0, int M, int P,
                                                const int32_t* __restrict__ fp, const int64_t* __restrict__ eptr,
                                                const int32_t* __restrict__ eloc, const int32_t* __restrict__ esrc,
                                                const double* __restrict__ vals, int* __restrict__ info, NdGatherCtx gc) {
  // tile = what a leaf is assembled from: [M x P pivot columns | P x B rows of the border columns]; the B x B block of a leaf
  // holds no matrix entry (an entry lives in the front that eliminates the earlier of its two dofs).  One wave per workgroup
  // and <= 17 KB of LDS per front: up to nine fronts per CU in independent phases, so the assembly latencies of one overlap
  // the arithmetic of the others.
  extern __shared__ double T[];
  const int lane = threadIdx.x;
  const int i = blockIdx.x;
  const int B = M - P, MP = M * P, tile = MP + P * B;
  const int64_t f = f0 + i;
  const int64_t e0 = eptr[f];
  const int ne = (int)(eptr[f + 1] - e0);
  const int npiv = fp[f];
  if (CHILD) {
    const NdGatherSrc g = nd_gather_src(gc, arena, f);
    for (int e = lane; e < tile; e += 64) {
      int c, r;
      if (e < MP) {
        c = e / M, r = e - c * M;
      } else {
        const int q = e - MP;
        c = P + q / P, r = q - (q / P) * P;
      }
      double v = 0.0;
      if (g.S0) {
        int a = g.I0[c], b = g.I0[r];
        if (gc.sym && b < a) {  // symmetric mode: (row, col) of a child's Schur block from (max, min)
          const int t = a;
          a = b, b = t;
        }
        if ((a | b) >= 0) v = g.S0[(int64_t)a * g.M0 + b];
      }
      if (g.S1) {
        int a = g.I1[c], b = g.I1[r];
        if (gc.sym && b < a) {
          const int t = a;
          a = b, b = t;
        }
        if ((a | b) >= 0) v += g.S1[(int64_t)a * g.M1 + b];
      }
      T[e] = v;
    }
  } else {
    for (int q = lane; q < tile; q += 64) T[q] = 0.0;
  }
  __syncthreads();
  // eight entries per lane and round, every load of a round issued before the first use: a round costs two memory
  // latencies (list, value) instead of two per entry
  for (int base = 0; base < ne; base += 512) {
    int loc[8], src[8];
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int t = base + 64 * q + lane;
      const bool ok = t < ne;
      loc[q] = ok ? eloc[e0 + t] : -1;
      src[q] = ok ? esrc[e0 + t] : 0;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = loc[q] >= 0 ? vals[src[q]] : 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (loc[q] >= 0) T[loc[q]] = CHILD ? T[loc[q]] + v[q] : v[q];  // one entry per position
  }
  for (int k = npiv + lane; k < P; k += 64) T[k * M + k] = 1.0;  // identity on the padded pivots (k_nd_pad)
  __syncthreads();
  const int r = lane;
  double* S = arena + store_off + (int64_t)i * ((int64_t)MP + (int64_t)P * B);
  double* F = arena + lev_off + (int64_t)i * M * M;
  // L columns: row r in L0 (0 on and above the diagonal, so the updates below need no row mask), row 64 + r in L1
  double L0[ND_LEAF_P], L1[TWO ? ND_LEAF_P : 1];
  int bad = 0;
  // pivot columns: column j against columns 0..j-1, then the multipliers of column j
#pragma unroll
  for (int j = 0; j < ND_LEAF_P; ++j) {
    if (j < P) {
      double x0 = r < M ? T[j * M + r] : 0.0, x1 = 0.0;
      if (TWO) x1 = r + 64 < M ? T[j * M + r + 64] : 0.0;
#pragma unroll
      for (int k = 0; k < j; ++k) {
        const double xk = nd_bcast(x0, k);
        x0 = fma(-L0[k], xk, x0);
        if (TWO) x1 = fma(-L1[k], xk, x1);
      }
      if (r == j && fabs(x0) < 1e-300) {
        x0 = x0 < 0 ? -1e-300 : 1e-300;
        bad = 1;
      }
      const double piv = nd_bcast(x0, j);
      L0[j] = r > j ? x0 / piv : 0.0;
      if (TWO) L1[j] = x1 / piv;
      if (r < M) S[(int64_t)j * M + r] = r > j ? L0[j] : x0;
      if (TWO && r + 64 < M) S[(int64_t)j * M + r + 64] = L1[j];
    } else {
      L0[j] = 0.0;
      if (TWO) L1[j] = 0.0;
    }
  }
  // border columns, two at a time (two independent dependency chains per lane): U12 = L11^-1 A12 on the first P rows,
  // Schur complement 0 - L21 U12 below.  Pivot steps in blocks of eight (steps beyond P multiply by L = 0).
  const double* TB = T + MP;
  for (int j = 0; j < B; j += 2) {
    const bool two = j + 1 < B;
    const int jb = two ? j + 1 : j;
    double x0 = r < P ? TB[j * P + r] : 0.0, y0 = r < P ? TB[jb * P + r] : 0.0, x1 = 0.0, y1 = 0.0;
#pragma unroll
    for (int k8 = 0; k8 < ND_LEAF_P; k8 += 8) {
      if (k8 < P) {
#pragma unroll
        for (int k = k8; k < k8 + 8; ++k) {
          const double xk = nd_bcast(x0, k), yk = nd_bcast(y0, k);
          x0 = fma(-L0[k], xk, x0);
          y0 = fma(-L0[k], yk, y0);
          if (TWO && !CHILD) {
            x1 = fma(-L1[k], xk, x1);
            y1 = fma(-L1[k], yk, y1);
          }
        }
      }
    }
    if (r < P) {
      S[MP + (int64_t)j * P + r] = x0;
      if (two) S[MP + (int64_t)jb * P + r] = y0;
    } else if (!CHILD && r < M) {
      F[(int64_t)(P + j) * M + r] = x0;
      if (two) F[(int64_t)(P + jb) * M + r] = y0;
    }
    if (!CHILD && TWO && r + 64 < M) {
      F[(int64_t)(P + j) * M + r + 64] = x1;
      if (two) F[(int64_t)(P + jb) * M + r + 64] = y1;
    }
  }
  if (bad) atomicAdd(info, 1);
}

template <typename T>
static int nd_upload(pgx_nd* s, T** d, const std::vector<T>& h) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, std::max<size_t>(h.size(), 1) * sizeof(T));
  if (e != hipSuccess) {
    s->err = std::string("hipMalloc: ") + hipGetErrorString(e);
    return PGX_ENOMEM;
  }
  s->allocs.push_back(q);
  *d = (T*)q;
  if (!h.empty()) NDHIP(hipMemcpy(q, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return PGX_OK;
}
template <typename T>
static int nd_alloc(pgx_nd* s, T** d, size_t count) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
  if (e != hipSuccess) {
    s->err = std::string("hipMalloc of ") + std::to_string(count * sizeof(T)) + " bytes: " + hipGetErrorString(e);
    return PGX_ENOMEM;
  }
  s->allocs.push_back(q);
  *d = (T*)q;
  return PGX_OK;
}

static int nd_create_impl(const pgx_nd_matrix* A, pgx_comm* comm, int device, void* hip_stream, pgx_nd** out,
                          int sym_rank = 0, int sym_size = 1) {
  if (!A || !out || A->n <= 0 || !A->rowptr || !A->col || !A->node_of_dof || !A->node_coords || A->n_nodes <= 0 ||
      (A->dim != 2 && A->dim != 3)) {
    g_nd_error = "pgx_nd_create: bad arguments";
    return PGX_EINVAL;
  }
  pgx_nd* s = new pgx_nd();
  s->n = A->n;
  if (comm)
    s->comm = comm, s->rank = comm->rank, s->size = comm->size;
  else
    s->rank = sym_rank, s->size = sym_size;  // symbolic-only view of one rank of a distributed factorisation (tests)
  const bool ptime = pgx_tune("PGX_ND_TIMING") != nullptr;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  const auto c_0 = tnow();
  int rc = nd_symbolic(s, A);
  const auto c_1 = tnow();
  if (rc) {
    g_nd_error = s->err;
    delete s;
    return rc;
  }
  if (device < 0) {
    *out = s;
    return PGX_OK;
  }
  auto fail = [&](int code) {
    g_nd_error = s->err;
    pgx_nd_destroy(s);
    return code;
  };
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) {
    s->err = "pgx_nd_create: no usable GPU (the direct solver has no CPU fallback)";
    return fail(PGX_ENODEV);
  }
  s->device = device;
  if (hipSetDevice(device) != hipSuccess) {
    s->err = "hipSetDevice failed";
    return fail(PGX_EHIP);
  }
  if (hip_stream) {
    s->st = (hipStream_t)hip_stream;
  } else {
    if (hipStreamCreate(&s->st) != hipSuccess) {
      s->err = "hipStreamCreate failed";
      return fail(PGX_EHIP);
    }
    s->own_stream = true;
  }
  std::vector<int32_t> fM(s->nfronts), fP(s->nfronts);
  std::vector<int64_t> vbase(s->nfronts);
  for (int64_t f = 0; f < s->nfronts; ++f) {
    const NdLevel& Lv = s->lev[s->flevel[f]];
    fM[f] = Lv.P + Lv.B;
    fP[f] = Lv.P;
    vbase[f] = Lv.voff + (f - Lv.start) * (int64_t)(Lv.P + Lv.B);
  }
  {  // assembly list: virtual destination -> working-buffer destination, sorted (stably) by the tree depth of the front
    if (s->nnz > (int64_t)INT32_MAX) {
      s->err = "pgx_nd_create: more than 2^31 matrix entries";
      return fail(PGX_EINVAL);
    }
    const int L = (int)s->lev.size();
    std::vector<int64_t> voffs(L);
    for (int l = 0; l < L; ++l) voffs[l] = s->lev[l].off;
    auto batch_of = [&](int64_t v) {  // last batch with off <= v and count > 0 containing v
      int l = (int)(std::upper_bound(voffs.begin(), voffs.end(), v) - voffs.begin()) - 1;
      while (l > 0 && s->lev[l].count == 0) --l;
      return l;
    };
    const int ng = (int)s->groups.size();
    std::vector<int> group_of(L, 0);
    for (int g = 0; g < ng; ++g)
      for (int l = s->groups[g].l0; l < s->groups[g].l1; ++l) group_of[l] = g;
    // a parallel counting sort by group, stable in the entry index: contiguous chunks of entries, one histogram per chunk, the
    // chunks' offsets in chunk order - the lists do not depend on the number of threads
    std::vector<int64_t> gnz(ng + 1, 0);
    std::vector<int32_t> bl(s->nnz, -1);
    int nthr = (int)std::min<int64_t>(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 8u),
                                     std::max<int64_t>(1, s->nnz / (1 << 20)));
    if (const char* e = getenv("PGX_ND_THREADS")) nthr = std::max(1, atoi(e));
    const int64_t chunk = (s->nnz + nthr - 1) / nthr;
    std::vector<std::vector<int64_t>> hist(nthr, std::vector<int64_t>(ng, 0));
    auto run_threads = [&](const std::function<void(int)>& fn) {
      std::vector<std::thread> pool;
      for (int i = 1; i < nthr; ++i) pool.emplace_back(fn, i);
      fn(0);
      for (auto& th : pool) th.join();
    };
    run_threads([&](int ti) {
      std::vector<int64_t>& hg = hist[ti];
      for (int64_t k = ti * chunk; k < std::min(s->nnz, (ti + 1) * chunk); ++k)
        if (s->dest[k] >= 0) {
          bl[k] = batch_of(s->dest[k]);
          hg[group_of[bl[k]]]++;
        }
    });
    for (int g = 0; g < ng; ++g)
      for (int ti = 0; ti < nthr; ++ti) gnz[g + 1] += hist[ti][g];
    for (int g = 0; g < ng; ++g) gnz[g + 1] += gnz[g];
    for (int g = 0; g < ng; ++g) s->groups[g].nz0 = gnz[g], s->groups[g].nz1 = gnz[g + 1];
    for (int g = 0; g < ng; ++g) {  // hist[ti][g] <- first slot of chunk ti in group g
      int64_t at = gnz[g];
      for (int ti = 0; ti < nthr; ++ti) {
        const int64_t c = hist[ti][g];
        hist[ti][g] = at;
        at += c;
      }
    }
    std::vector<int64_t> sdest(gnz[ng]);
    std::vector<int32_t> ssrc(gnz[ng]);
    run_threads([&](int ti) {
      std::vector<int64_t>& cur = hist[ti];
      for (int64_t k = ti * chunk; k < std::min(s->nnz, (ti + 1) * chunk); ++k)
        if (bl[k] >= 0) {
          const NdLevel& Lv = s->lev[bl[k]];
          const int64_t t = cur[group_of[bl[k]]]++;
          sdest[t] = Lv.woff + (s->dest[k] - Lv.off);  // same front-local position: both layouts use M x M fronts here
          ssrc[t] = (int32_t)k;
        }
    });
    if ((rc = nd_upload(s, &s->d_sdest, sdest)) || (rc = nd_upload(s, &s->d_ssrc, ssrc))) return fail(rc);
    std::vector<int64_t>().swap(sdest);
    std::vector<int32_t>().swap(ssrc);
    for (int64_t f = 0; f < s->nfronts; ++f) {
      const NdLevel& Lv = s->lev[s->flevel[f]];
      s->fbase[f] = Lv.woff + (f - Lv.start) * (int64_t)(Lv.P + Lv.B) * (Lv.P + Lv.B);
    }
    // fused leaves (k_nd_leaf): the groups of the deepest depth whose fronts are small enough get per-front entry lists
    {
      const char* e = pgx_tune("PGX_ND_LEAF_FUSED");
      s->leaf_fuse = !e || atoi(e) != 0;
    }
    const int maxdepth = (int)s->dfirst.size() - 2;
    bool any = false;
    // ... and, with the parent-centric assembly, the groups of small fronts WITH children (not the push-mode depth of a subtree
    // cut): their frame is gathered and eliminated by the same kernel (PGX_ND_FRAME_FUSED=0: k_nd_gather + diag + panel)
    bool frame_fuse = s->leaf_fuse;
    {
      const char* e = pgx_tune("PGX_ND_GATHER");
      if (e && atoi(e) == 0) frame_fuse = false;
      const char* e2 = pgx_tune("PGX_ND_FRAME_FUSED");
      if (e2 && atoi(e2) == 0) frame_fuse = false;
    }
    for (int g = 0; g < ng && s->leaf_fuse; ++g) {
      pgx_nd::Group& G = s->groups[g];
      const bool deepest = G.depth == maxdepth;
      if (!deepest && (!frame_fuse || G.depth == s->kcut)) continue;
      bool ok = true, some = false;
      for (int l = G.l0; l < G.l1 && ok; ++l) {
        const NdLevel& Lv = s->lev[l];
        if (Lv.count == 0) continue;
        some = true;
        if (Lv.P > ND_LEAF_P || Lv.P + Lv.B > ND_LEAF_M || Lv.P < 1) ok = false;
        for (int64_t f = Lv.start; f < Lv.start + Lv.count && ok && deepest; ++f)
          if (s->child0[f] >= 0 || s->child1[f] >= 0) ok = false;
      }
      (deepest ? G.leaf_fused : G.frame_fused) = ok && some;
      any = any || (ok && some);
    }
    if (any) {
      std::vector<int64_t> lptr(s->nfronts + 1, 0);
      auto front_of = [&](int64_t k, int* loc) -> int64_t {  // front and front-local position of matrix entry k, or -1
        const int l = bl[k];  // batch of the entry (-1: assembled on another rank)
        if (l < 0 || !(s->groups[group_of[l]].leaf_fused || s->groups[group_of[l]].frame_fused)) return -1;
        const NdLevel& Lv = s->lev[l];
        const int M = Lv.P + Lv.B;
        const int64_t MM = (int64_t)M * M, q = s->dest[k] - Lv.off;
        const int lq = (int)(q % MM), c = lq / M, r = lq % M;  // column-major position in the M x M front
        // the kernel's tile: pivot columns whole, border columns on the pivot rows only; a leaf has no entry elsewhere
        *loc = c < Lv.P ? lq : (r < Lv.P ? M * Lv.P + (c - Lv.P) * Lv.P + r : -1);
        return Lv.start + q / MM;
      };
      int loc = 0;
      for (int64_t k = 0; k < s->nnz && any; ++k) {
        const int64_t f = front_of(k, &loc);
        if (f >= 0) lptr[f + 1]++;
        if (f >= 0 && loc < 0) any = false;  // an entry in a leaf's border block (not with these assembly maps): batched path
      }
      if (!any)
        for (auto& G : s->groups) G.leaf_fused = G.frame_fused = false;
      for (int64_t f = 0; f < s->nfronts; ++f) lptr[f + 1] += lptr[f];
      std::vector<int32_t> lloc(std::max<int64_t>(lptr[s->nfronts], 1)), lsrc(std::max<int64_t>(lptr[s->nfronts], 1));
      std::vector<int64_t> fill(lptr.begin(), lptr.end() - 1);
      for (int64_t k = 0; k < s->nnz && any; ++k) {
        const int64_t f = front_of(k, &loc);
        if (f >= 0) {
          const int64_t t = fill[f]++;
          lloc[t] = loc;
          lsrc[t] = (int32_t)k;
        }
      }
      if (any && ((rc = nd_upload(s, &s->d_leaf_ptr, lptr)) || (rc = nd_upload(s, &s->d_leaf_loc, lloc)) ||
                  (rc = nd_upload(s, &s->d_leaf_src, lsrc))))
        return fail(rc);  // (the tile is at most (128 x 32 + 32 x 96) doubles = 56 KB: no LDS attribute needed)
    }
  }
  const auto c_2 = tnow();
#define UP(d, h)                        \
  if ((rc = nd_upload(s, &s->d, h))) return fail(rc);
  UP(d_dof_ptr, s->dof_ptr) UP(d_rel_ptr, s->rel_ptr) UP(d_fbase, s->fbase) UP(d_fp, s->fp) UP(d_fb, s->fb)
  UP(d_parent, s->parent) UP(d_slot01, s->slot01) UP(d_child0, s->child0) UP(d_child1, s->child1) UP(d_own_dofs, s->own_dofs)
  UP(d_rel, s->rel)
#undef UP
  if ((rc = nd_upload(s, &s->d_fM, fM)) || (rc = nd_upload(s, &s->d_fP, fP)) || (rc = nd_upload(s, &s->d_vbase, vbase)))
    return fail(rc);
  {  // inverse child -> parent maps of the parent-centric assembly
    const char* e = pgx_tune("PGX_ND_GATHER");
    s->gather = !e || atoi(e) != 0;
    if (s->gather) {
      std::vector<int32_t> inv[2];
      inv[0].assign((size_t)std::max<int64_t>(s->vec_len, 1), -1);
      inv[1].assign((size_t)std::max<int64_t>(s->vec_len, 1), -1);
      for (int64_t c = 0; c < s->nfronts; ++c) {
        const int pf = s->parent[c];
        if (pf < 0) continue;
        std::vector<int32_t>& I = inv[s->slot01[c] ? 1 : 0];
        const int32_t* R = s->rel.data() + s->rel_ptr[c];
        const int64_t nb = s->rel_ptr[c + 1] - s->rel_ptr[c];
        for (int64_t k = 0; k < nb; ++k) I[vbase[pf] + R[k]] = (int32_t)k;
      }
      if ((rc = nd_upload(s, &s->d_inv[0], inv[0])) || (rc = nd_upload(s, &s->d_inv[1], inv[1]))) return fail(rc);
    }
  }
  // the maps live on the device from here on: release the host copies of the large ones (8 bytes per matrix entry)
  std::vector<int64_t>().swap(s->dest);
  std::vector<int32_t>().swap(s->own_dofs);
  std::vector<int32_t>().swap(s->rel);
  if ((rc = nd_alloc(s, &s->arena, (size_t)s->arena_len)) || (rc = nd_alloc(s, &s->vec, (size_t)s->vec_len)) ||
      (rc = nd_alloc(s, &s->d_vals, (size_t)s->nnz)) || (rc = nd_alloc(s, &s->d_b, (size_t)s->n)))
    return fail(rc);
  int64_t maxbatch = 1;
  for (auto& L : s->lev) maxbatch = std::max(maxbatch, L.count);
  if ((rc = nd_alloc(s, &s->d_info, (size_t)maxbatch))) return fail(rc);
  hipEventCreate(&s->e0);
  hipEventCreate(&s->e1);
  hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming);
  hipEventCreateWithFlags(&s->ev_info, hipEventDisableTiming);
  if (hipHostMalloc((void**)&s->h_info, sizeof(int)) != hipSuccess) s->h_info = nullptr;
  {
    const char* e = pgx_tune("PGX_ND_PREP_AHEAD");
    s->prep_ahead = !e || atoi(e) != 0;
    if (hipStreamCreateWithFlags(&s->prep_st, hipStreamNonBlocking) != hipSuccess) s->prep_st = nullptr;
    hipEventCreateWithFlags(&s->ev_prep_go, hipEventDisableTiming);
    hipEventCreateWithFlags(&s->ev_prep_done, hipEventDisableTiming);
  }
  for (int i = 0; i < 3; ++i) {
    if (hipStreamCreateWithFlags(&s->side[i], hipStreamNonBlocking) != hipSuccess) s->side[i] = nullptr;
    hipEventCreateWithFlags(&s->ev_join[i], hipEventDisableTiming);
  }
  if (s->size > 1) {  // exchange buffers: one Schur block / border vector per rank on rank 0, one on the others
    const NdLevel& Lk = s->lev[s->kbatch];
    const size_t nb = (size_t)Lk.B * Lk.B, mult = s->rank == 0 ? (size_t)s->size : 1;
    if ((rc = nd_alloc(s, &s->d_xbuf, mult * nb)) || (rc = nd_alloc(s, &s->d_vbuf, mult * (size_t)Lk.B))) return fail(rc);
  }
  s->dprof = pgx_tune("PGX_ND_DEPTHPROF") != nullptr;
  if (const char* e = pgx_tune("PGX_ND_SOLVE_SMALL")) s->solve_small = atoi(e) != 0;
  if (const char* e = pgx_tune("PGX_ND_LSHAPE")) s->lshape = atoi(e) != 0;
  if (const char* e = pgx_tune("PGX_ND_TRSV_BIG")) s->trsv_big = atoi(e) != 0;
  if (const char* e = pgx_tune("PGX_ND_LEFTLOOK")) s->leftlook = atoi(e) != 0;
  if (const char* e = pgx_tune("PGX_ND_TILEORDER")) s->tile_order = atoi(e) != 0;
  if (const char* e = pgx_tune("PGX_ND_SYM")) s->sym_allowed = atoi(e) != 0;
  if (const char* e = pgx_tune("PGX_ND_OUTER")) s->outer = std::max(64, (atoi(e) / 64) * 64);
  if (s->trsv_big && hipFuncSetAttribute((const void*)k_nd_trsv_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ND_BIG_LDS) != hipSuccess) {
    (void)hipGetLastError();
    s->trsv_big = false;  // (no 140 KB of LDS per workgroup on this device: the four-wave kernel everywhere)
  }  // per-depth table on stderr when the handle is destroyed
  if (ptime) {
    hipDeviceSynchronize();
    fprintf(stderr, "pgx_nd create: symbolic %.0f ms, assembly / leaf lists %.0f ms, maps, uploads, device allocations %.0f ms\n", tms(c_0, c_1),
            tms(c_1, c_2), tms(c_2, tnow()));
  }
  *out = s;
  return PGX_OK;
}

extern "C" int pgx_nd_create(const pgx_nd_matrix* A, int device, void* hip_stream, pgx_nd** out) {
  return nd_create_impl(A, nullptr, device, hip_stream, out);
}

extern "C" int pgx_nd_create_symbolic_dist(const pgx_nd_matrix* A, int rank, int size, pgx_nd** out) {
  if (size < 1 || rank < 0 || rank >= size) {
    g_nd_error = "pgx_nd_create_symbolic_dist: bad rank / size";
    return PGX_EINVAL;
  }
  return nd_create_impl(A, nullptr, -1, nullptr, out, rank, size);
}

extern "C" int pgx_nd_export_dist(const pgx_nd* s, int32_t* kdist, int32_t* kbatch, int32_t* root_slot, int32_t* ghost_slot) {
  if (!s) return PGX_EINVAL;
  if (kdist) *kdist = s->kdist;
  if (kbatch) *kbatch = s->kbatch;
  if (root_slot) *root_slot = s->root_slot;
  if (ghost_slot)
    for (size_t j = 0; j < s->ghost_slot.size(); ++j) ghost_slot[j] = s->ghost_slot[j];
  return PGX_OK;
}

extern "C" int pgx_nd_create_dist(const pgx_nd_matrix* A, pgx_comm* comm, int device, void* hip_stream, pgx_nd** out) {
  if (!comm) {
    g_nd_error = "pgx_nd_create_dist: null communicator";
    return PGX_EINVAL;
  }
  if (device < 0) {
    g_nd_error = "pgx_nd_create_dist needs a device";
    return PGX_EINVAL;
  }
  return nd_create_impl(A, comm, device, hip_stream, out);
}

// PGX_ND_DEPTHPROF: where the device time of this handle's factorisations and solves went, tree depth by tree depth
static void nd_dp_report(const pgx_nd* s) {
  if (!s->dprof || s->dp_calls[0] + s->dp_calls[1] == 0) return;
  const int nd = (int)s->dfirst.size() - 1;
  const double nf = std::max(1, s->dp_calls[0]), ns = std::max(1, s->dp_calls[1]);
  fprintf(stderr, "pgx_nd depth profile: n = %lld, %d factorisations, %d solves; per call, ms\n", (long long)s->n, s->dp_calls[0], s->dp_calls[1]);
  fprintf(stderr, "%5s %8s %8s %8s %9s %8s %8s %7s %7s  batches (count x P+B)\n", "depth", "factor", "fwd", "bwd", "GF(pad)", "TF/s", "GB fact", "TB/s f", "TB/s s");
  double tot[3] = {0, 0, 0}, totf = 0, totb = 0;
  for (int d = nd - 1; d >= 0; --d) {
    double fl = 0, by = 0;
    std::string bl;
    for (int l = s->dfirst[d]; l < s->dfirst[d + 1]; ++l) {
      const NdLevel& Lv = s->lev[l];
      if (!Lv.count) continue;
      const double P = Lv.P, B = Lv.B;
      fl += Lv.count * (2.0 / 3.0 * P * P * P + 2.0 * P * P * B + 2.0 * P * B * B);
      by += Lv.count * 8.0 * ((P + B) * P + P * B);
      char t[64];
      snprintf(t, sizeof t, " %lldx(%d+%d)", (long long)Lv.count, Lv.P, Lv.B);
      bl += t;
    }
    double m[3];
    for (int ph = 0; ph < 3; ++ph) m[ph] = d < (int)s->dp_ms[ph].size() ? s->dp_ms[ph][d] / (ph ? ns : nf) : 0.0, tot[ph] += m[ph];
    totf += fl, totb += by;
    fprintf(stderr, "%5d %8.3f %8.3f %8.3f %9.1f %8.2f %8.3f %7.2f %7.2f %s\n", d, m[0], m[1], m[2], fl / 1e9, m[0] > 0 ? fl / m[0] / 1e9 : 0.0, by / 1e9,
            m[0] > 0 ? by / m[0] / 1e9 : 0.0, m[1] + m[2] > 0 ? 2 * by / (m[1] + m[2]) / 1e9 : 0.0, bl.c_str());
  }
  fprintf(stderr, "%5s %8.3f %8.3f %8.3f %9.1f %8.2f %8.3f %7.2f %7.2f\n", "all", tot[0], tot[1], tot[2], totf / 1e9, tot[0] > 0 ? totf / tot[0] / 1e9 : 0.0,
          totb / 1e9, tot[0] > 0 ? totb / tot[0] / 1e9 : 0.0, tot[1] + tot[2] > 0 ? 2 * totb / (tot[1] + tot[2]) / 1e9 : 0.0);
}

extern "C" void pgx_nd_destroy(pgx_nd* s) {
  if (!s) return;
  if (s->device >= 0) {
    hipSetDevice(s->device);
    if (s->st) hipStreamSynchronize(s->st);
    nd_dp_report(s);
    for (void* p : s->allocs) hipFree(p);
    if (s->e0) hipEventDestroy(s->e0);
    if (s->e1) hipEventDestroy(s->e1);
    for (hipEvent_t e : s->dp_ev) hipEventDestroy(e);
    if (s->ev_fork) hipEventDestroy(s->ev_fork);
    if (s->ev_info) hipEventDestroy(s->ev_info);
    if (s->h_info) hipHostFree(s->h_info);
    if (s->prep_st) hipStreamSynchronize(s->prep_st), hipStreamDestroy(s->prep_st);
    if (s->ev_prep_go) hipEventDestroy(s->ev_prep_go);
    if (s->ev_prep_done) hipEventDestroy(s->ev_prep_done);
    for (int i = 0; i < 3; ++i) {
      if (s->side[i]) hipStreamSynchronize(s->side[i]), hipStreamDestroy(s->side[i]);
      if (s->ev_join[i]) hipEventDestroy(s->ev_join[i]);
    }
    if (s->own_stream && s->st) hipStreamDestroy(s->st);
  }
  delete s;
}

// number of perturbed pivots of the last factorisation, once its read-back has arrived (wait: block until it has)
static void nd_poll_info(pgx_nd* s, bool wait) {
  if (!s->info_pending || !s->h_info) return;
  if (wait ? hipEventSynchronize(s->ev_info) != hipSuccess : hipEventQuery(s->ev_info) != hipSuccess) return;
  s->info_pending = false;
  s->stats.perturbed_pivots = *s->h_info;
  if (*s->h_info > 0)
    s->err = "pgx_nd_factor: " + std::to_string(*s->h_info) + " (near-)zero pivot(s) replaced by +-1e-300: the matrix is singular to working "
             "precision in the elimination order (no pivoting across fronts); solves with this factorisation are unreliable";
}

extern "C" int pgx_nd_get_stats(const pgx_nd* s, pgx_nd_stats* st) {
  if (!s || !st) return PGX_EINVAL;
  nd_poll_info(const_cast<pgx_nd*>(s), true);
  *st = s->stats;
  return PGX_OK;
}

extern "C" int pgx_nd_set_symmetric(pgx_nd* s, int on) {
  if (!s) return PGX_EINVAL;
  // eligible: the parent-centric (gather) assembly and the MFMA panel kernel, on one rank or distributed, cut schedule or not (the
  // extend-add at the cut reads a subtree root's Schur block through (max, min) as well; the
  // Schur blocks of the subtree roots travel whole; rank 0 gathers from them through (max, min) like from any child) - everything the
  // configurations of BASELINE.json run; otherwise the request is ignored and the general LU runs (pgx_nd_is_symmetric tells)
  const bool ok = s->sym_allowed && s->device >= 0 && s->gather && s->d_inv[0] && s->d_inv[1] && s->panel_kind == 0;
  s->sym = (on && ok) ? 1 : 0;
  s->factored = false;
  return PGX_OK;
}
extern "C" int pgx_nd_is_symmetric(const pgx_nd* s) { return s ? s->sym : 0; }
// test hook (host arithmetic only, no GPU): the tile enumeration of the symmetric GEMM launches - number of tiles (tr, tc) of an
// nr x nc rectangle with tc <= tr + band, and tile number t of that list (pgx_nd_gemm.h)
extern "C" int pgx_nd_sym_tile_count(int nr, int nc, int band) { return nd_sym_tiles(nr, nc, band); }
extern "C" void pgx_nd_sym_tile_at(int t, int nr, int nc, int band, int* tr, int* tc) {
  int a = 0, b = 0;
  nd_sym_tile(t, nr, nc, band, a, b);
  *tr = a, *tc = b;
}

extern "C" int pgx_nd_timing(pgx_nd* s, int enable, double* factor_ms, double* solve_ms) {
  if (!s) return PGX_EINVAL;
  if (factor_ms) *factor_ms = s->factor_ms;
  if (solve_ms) *solve_ms = s->solve_ms;
  s->timing = enable != 0;
  s->factor_ms = s->solve_ms = 0;
  return PGX_OK;
}

// per-depth profile: mark() records an event on the main stream and tags the interval it closes; collect() turns the
// events of the finished call into milliseconds per (phase, depth)
static void nd_dp_mark(pgx_nd* s, size_t& used, int phase, int depth) {
  if (!s->dprof) return;
  if (used >= s->dp_ev.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return;
    s->dp_ev.push_back(e);
    s->dp_tag.push_back({0, 0});
  }
  hipEventRecord(s->dp_ev[used], s->st);
  s->dp_tag[used] = {phase, depth};
  ++used;
}
static void nd_dp_collect(pgx_nd* s, size_t used) {
  if (!s->dprof || used < 2) return;
  hipEventSynchronize(s->dp_ev[used - 1]);
  bool seen[3] = {false, false, false};
  for (size_t i = 1; i < used; ++i) {
    float ms = 0;
    hipEventElapsedTime(&ms, s->dp_ev[i - 1], s->dp_ev[i]);
    const int ph = s->dp_tag[i].first, d = s->dp_tag[i].second;
    if (ph < 0 || d < 0) continue;
    if ((int)s->dp_ms[ph].size() <= d) s->dp_ms[ph].resize(d + 1, 0.0);
    s->dp_ms[ph][d] += ms;
    seen[ph] = true;
  }
  for (int ph = 0; ph < 3; ++ph) s->dp_calls[ph] += seen[ph];
}

// Diagnostic: device time per tree depth.  enable != 0 switches the recording on (and clears the sums); with arrays of
// n_depths entries the accumulated milliseconds of the factorisations / forward / backward sweeps since then are returned
// (calls[3] = number of factorisations, forward and backward sweeps summed).  n_depths in: capacity, out: tree depths.
extern "C" int pgx_nd_depth_profile(pgx_nd* s, int enable, int32_t* n_depths, double* factor_ms, double* fwd_ms, double* bwd_ms,
                                    int32_t* calls) {
  if (!s) return PGX_EINVAL;
  const int nd = (int)s->dfirst.size() - 1;
  if (n_depths) {
    const int cap = *n_depths;
    *n_depths = nd;
    double* out[3] = {factor_ms, fwd_ms, bwd_ms};
    for (int ph = 0; ph < 3; ++ph)
      if (out[ph])
        for (int d = 0; d < std::min(cap, nd); ++d) out[ph][d] = d < (int)s->dp_ms[ph].size() ? s->dp_ms[ph][d] : 0.0;
    if (calls)
      for (int ph = 0; ph < 3; ++ph) calls[ph] = s->dp_calls[ph];
  }
  if (enable >= 0) {
    s->dprof = enable != 0;
    for (int ph = 0; ph < 3; ++ph) s->dp_ms[ph].clear(), s->dp_calls[ph] = 0;
  }
  return PGX_OK;
}

// stream of the idx-th non-empty batch of the current depth (idx 0: the main stream); nd_join waits for all of them
static hipStream_t nd_fork(pgx_nd* s, int idx) {
  if (idx == 0) return s->st;
  hipStream_t q = s->side[(idx - 1) % 3];
  if (!q) return s->st;
  if (idx <= 3) hipStreamWaitEvent(q, s->ev_fork, 0);
  return q;
}
static void nd_join(pgx_nd* s, int nused) {
  for (int i = 0; i < std::min(nused - 1, 3); ++i)
    if (s->side[i]) {
      hipEventRecord(s->ev_join[i], s->side[i]);
      hipStreamWaitEvent(s->st, s->ev_join[i], 0);
    }
}

// C -= A B on one rectangle of every front of a level: 128 x 128 tiles where both sides are long, 64 x 64 otherwise

// cgather: C is not read but assembled from the children's Schur blocks (the border block of a parent-centrically assembled front)
static void nd_launch_gemm(pgx_nd* s, hipStream_t q, const NdLevel& Lv, int r0, int r1, int c0, int c1, int k0, int k1,
                           bool cgather = false, int symskip = 0) {
  if (r1 <= r0 || c1 <= c0 || k1 <= k0) return;
  const int M = Lv.P + Lv.B;
  const bool big = (r1 - r0) >= 256 && (c1 - c0) >= 256;
  const int TS = big ? 128 : 64;
  dim3 grid((unsigned)Lv.count, (unsigned)((r1 - r0 + TS - 1) / TS), (unsigned)((c1 - c0 + TS - 1) / TS));
  if (symskip > 0) {  // symmetric mode: only the tiles on / below the (band) diagonal are launched (pgx_nd_gemm.h nd_sym_tile)
    const int nlow = nd_sym_tiles((int)grid.y, (int)grid.z, symskip - 1);
    if (nlow <= 65535)
      grid = dim3((unsigned)Lv.count, (unsigned)nlow, 1);
    else
      symskip = -symskip;
  }
  NdGatherCtx gc{Lv.start, s->d_child0, s->d_child1, s->d_fM, s->d_fP, s->d_inv[0], s->d_inv[1], s->d_fbase, s->d_vbase, s->sym};
  if (big && cgather)
    hipLaunchKernelGGL(k_nd_gemm8<true>, grid, dim3(512), 0, q, s->arena, Lv.woff, M, r0, r1, c0, c1, k0, k1, Lv.poff, Lv.P, gc, -1, s->tile_order, symskip);
  else if (big)
    hipLaunchKernelGGL(k_nd_gemm8<false>, grid, dim3(512), 0, q, s->arena, Lv.woff, M, r0, r1, c0, c1, k0, k1, Lv.poff, Lv.P, gc, -1, s->tile_order, symskip);
  else if (cgather)
    hipLaunchKernelGGL((k_nd_gemm<2, true>), grid, dim3(256), 0, q, s->arena, Lv.woff, M, r0, r1, c0, c1, k0, k1, Lv.poff, Lv.P, gc, -1, s->tile_order, symskip);
  else
    hipLaunchKernelGGL((k_nd_gemm<2, false>), grid, dim3(256), 0, q, s->arena, Lv.woff, M, r0, r1, c0, c1, k0, k1, Lv.poff, Lv.P, gc, -1, s->tile_order, symskip);
}

extern "C" int pgx_nd_factor(pgx_nd* s, const double* vals, int on_device) {
  if (!s || !vals) return PGX_EINVAL;
  if (s->device < 0) {
    s->err = "pgx_nd_factor: symbolic-only handle (created with device < 0); there is no CPU numeric phase";
    return PGX_ENODEV;
  }
  NDHIP(hipSetDevice(s->device));
  PgxRange range("pgx_nd:factor");
  const double* dv = vals;
  if (!on_device) {
    NDHIP(hipMemcpyAsync(s->d_vals, vals, (size_t)s->nnz * sizeof(double), hipMemcpyHostToDevice, s->st));
    dv = s->d_vals;
  }
  if (s->timing) hipEventRecord(s->e0, s->st);
  NDHIP(hipMemsetAsync(s->d_info, 0, sizeof(int), s->st));
  const int maxdepth = (int)s->dfirst.size() - 2;
  // the working buffer of a group: zero, matrix entries, identity on the padded pivots.  (Its previous tenants are
  // stored compactly and their Schur blocks have been absorbed by their parents.)
  auto prep_on = [&](const pgx_nd::Group& G, hipStream_t ps) -> int {
    if (G.w_len <= 0 || G.leaf_fused) return PGX_OK;  // fused leaves assemble in LDS and write every entry that is read later
    NDHIP(hipMemsetAsync(s->arena + G.w_off, 0, (size_t)G.w_len * sizeof(double), ps));
    if (G.nz1 > G.nz0) {
      int blocks = (int)std::min<int64_t>((G.nz1 - G.nz0 + 255) / 256, 256 * 64);
      hipLaunchKernelGGL(k_nd_scatter, dim3(blocks), dim3(256), 0, ps, G.nz0, G.nz1, s->d_sdest, s->d_ssrc, dv, s->arena);
    }
    const int64_t f0 = s->lev[G.l0].start, f1 = s->lev[G.l1 - 1].start + s->lev[G.l1 - 1].count;
    if (f1 > f0)
      hipLaunchKernelGGL(k_nd_pad, dim3((unsigned)(f1 - f0)), dim3(64), 0, ps, f0, s->d_fp, s->d_fP, s->d_fM, s->d_fbase,
                         s->arena);
    return PGX_OK;
  };
  auto prep = [&](const pgx_nd::Group& G) -> int { return prep_on(G, s->st); };
  // parent-centric assembly of a group whose children (depth + 1) have been eliminated: children's Schur blocks gathered and
  // written (k_nd_gather), matrix entries added, identity on the padded pivots - instead of prep + extend of the children
  auto gather = [&](const pgx_nd::Group& G) {
    if (G.frame_fused) return;  // k_nd_leaf<., true> gathers the frame itself
    for (int l = G.l0; l < G.l1; ++l) {
      const NdLevel& Lv = s->lev[l];
      if (Lv.count == 0) continue;
      const int M = Lv.P + Lv.B;
      const int64_t per = (int64_t)M * Lv.P + (int64_t)Lv.P * Lv.B;  // the frame; the border block belongs to the Schur update
      unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>((per + 2047) / 2048, 2048));
      while ((int64_t)gy * Lv.count > (int64_t)1 << 22 && gy > 1) gy /= 2;
      hipLaunchKernelGGL(k_nd_gather, dim3((unsigned)Lv.count, gy), dim3(256), 0, s->st, Lv.start, M, Lv.P, s->d_child0, s->d_child1, s->d_fM,
                         s->d_fP, s->d_fbase, s->d_vbase, s->d_inv[0], s->d_inv[1], s->arena, s->sym);
    }
    if (G.nz1 > G.nz0) {
      int blocks = (int)std::min<int64_t>((G.nz1 - G.nz0 + 255) / 256, 256 * 64);
      hipLaunchKernelGGL(k_nd_scatter_add, dim3(blocks), dim3(256), 0, s->st, G.nz0, G.nz1, s->d_sdest, s->d_ssrc, dv, s->arena);
    }
    const int64_t f0 = s->lev[G.l0].start, f1 = s->lev[G.l1 - 1].start + s->lev[G.l1 - 1].count;
    if (f1 > f0)
      hipLaunchKernelGGL(k_nd_pad, dim3((unsigned)(f1 - f0)), dim3(64), 0, s->st, f0, s->d_fp, s->d_fP, s->d_fM, s->d_fbase, s->arena);
  };
  const bool use_gather = s->gather && s->d_inv[0] && s->d_inv[1];
  // extend-add of the Schur complements of a (factorised) group into its parents' fronts (two conflict-free passes: first
  // children, second children)
  auto extend = [&](const pgx_nd::Group& G) {
    for (int cb = G.l0; cb < G.l1; ++cb) {
      const NdLevel& C = s->lev[cb];
      if (C.count == 0 || C.B == 0) continue;
      int64_t per = (int64_t)C.B * C.B;
      unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>((per + 2047) / 2048, 1024));
      while ((int64_t)gy * C.count > (int64_t)1 << 22 && gy > 1) gy /= 2;  // keep the grid bounded for very wide levels
      for (int pass = 0; pass < 2; ++pass)
        hipLaunchKernelGGL(k_nd_extend_add, dim3((unsigned)C.count, gy), dim3(256), 0, s->st, C.start, pass, C.P, C.P + C.B,
                           s->d_fb, s->d_slot01, s->d_parent, s->d_fM, s->d_fbase, s->d_rel_ptr, s->d_rel, s->arena, s->sym);
    }
  };
  auto eliminate = [&](const pgx_nd::Group& G, bool cgather = false) {  // the batches of a group on forked streams
    hipEventRecord(s->ev_fork, s->st);
    int used = 0;
    for (int l = G.l1 - 1; l >= G.l0; --l) {
      const NdLevel& Lv = s->lev[l];
      const int P = Lv.P, B = Lv.B, M = P + B;
      if (Lv.count == 0) continue;  // distributed: the levels above the subtrees live on rank 0
      hipStream_t q = nd_fork(s, used++);
      if (G.leaf_fused || (cgather && G.frame_fused)) {
        // one wave per front: assemble in LDS, eliminate in registers; a leaf writes factors + Schur block once, a front with
        // children its factors - its border block comes from the GATHER Schur update below
        const size_t lds = ((size_t)M * P + (size_t)P * B) * sizeof(double);
        const NdGatherCtx gc{Lv.start, s->d_child0, s->d_child1, s->d_fM, s->d_fP, s->d_inv[0], s->d_inv[1], s->d_fbase, s->d_vbase, s->sym};
#define ND_LEAF(TWO, CH)                                                                                                          \
  hipLaunchKernelGGL((k_nd_leaf<TWO, CH>), dim3((unsigned)Lv.count), dim3(64), lds, q, s->arena, Lv.woff, Lv.poff, Lv.start, M, P, \
                     s->d_fp, s->d_leaf_ptr, s->d_leaf_loc, s->d_leaf_src, dv, s->d_info, gc)
        if (G.leaf_fused) {
          if (M > 64)
            ND_LEAF(true, false);
          else
            ND_LEAF(false, false);
        } else {
          if (M > 64)
            ND_LEAF(true, true);
          else
            ND_LEAF(false, true);
          if (B > 0) nd_launch_gemm(s, q, Lv, P, M, P, M, 0, P, true, s->sym ? 1 : 0);
        }
#undef ND_LEAF
        continue;
      }
      // Two-level blocked partial LU of the batch.  Outer blocks of <= ND_OUTER pivots; inside one, <= 64-wide panels:
      // diagonal block, both panel solves, then rank-64 updates of the outer block's row and column STRIPS only.  The
      // rest of the trailing pivot block and panels gets ONE rank-ND_OUTER update per outer block (arithmetic intensity
      // ND_OUTER/8 flop/byte: MFMA-bound instead of HBM-bound), the Schur block F22 ONE update with K = P at the end.
      const int nouter = (P + s->outer - 1) / s->outer;
      int ob = 0;
      for (int ou = 0; ou < nouter; ++ou) {
        const int W = P / nouter + (ou < P % nouter ? 1 : 0), oe = ob + W;
        const int nsteps = (W + ND_NB - 1) / ND_NB;
        int kb = ob;
        for (int st = 0; st < nsteps; ++st) {
          const int nb = W / nsteps + (st < W % nsteps ? 1 : 0), ke = kb + nb;
          hipLaunchKernelGGL(k_nd_diag, dim3((unsigned)Lv.count), dim3(256), 0, q, s->arena, Lv.woff, M, kb, nb, s->d_info, Lv.poff, P);
          if (M - ke > 0) {
            const unsigned nch = (unsigned)((M - ke + ND_TS - 1) / ND_TS);
            nd_launch_panel(s->panel_kind, q, (unsigned)Lv.count, nch, s->arena, Lv.woff, M, kb, nb, Lv.poff, P, s->sym);
            if (s->sym) {
              // symmetric, left-looking: the next step reads its diagonal block and its block COLUMN only - rows [ke, M) x columns
              // [ke, kn), a rectangle (the block row is the scaled transpose the panel kernel writes)
              if (st + 1 < nsteps) {
                const int kn = ke + (W / nsteps + (st + 1 < W % nsteps ? 1 : 0));
                nd_launch_gemm(s, q, Lv, ke, M, ke, kn, ob, ke);
              }
            } else if (s->leftlook) {
              // LEFT-LOOKING inside the outer block (round 5): only what the NEXT 64-pivot step reads is brought up to date - its block
              // row [ke, kn) x [ke, M) and block column [kn, M) x [ke, kn), ONE launch over that L-shaped region - with ALL the pivots
              // of the outer block eliminated so far (K = [ob, ke): rank 64, 128, 192).  The right-looking form updated the whole
              // remaining strips of the outer block after every step (two launches, ~47 us of the ~96 us a step costs on the chain of
              // a level near the root); the flops are the same, the rows and columns beyond the next block wait for their turn.
              if (st + 1 < nsteps) {
                const int kn = ke + (W / nsteps + (st + 1 < W % nsteps ? 1 : 0));
                const dim3 grid((unsigned)Lv.count, (unsigned)nd_lshape_tiles(64, ke, M, kn), 1);
                NdGatherCtx gc{};
                hipLaunchKernelGGL((k_nd_gemm<2, false>), grid, dim3(256), 0, q, s->arena, Lv.woff, M, ke, M, ke, M, ob, ke, Lv.poff, P, gc, kn, s->tile_order, 0);
              }
            } else {
              nd_launch_gemm(s, q, Lv, ke, oe, ke, M, kb, ke);  // row strip of the outer block, all remaining columns
              nd_launch_gemm(s, q, Lv, oe, M, ke, oe, kb, ke);  // column strip of the outer block, rows below it
            }
          }
          kb = ke;
        }
        // trailing matrix beyond the outer block, without the Schur block: one launch over the L-shaped region
        if (s->sym) {
          // symmetric: the lower part of the pivot block and the columns below it - rows [oe, M) x columns [oe, P) without the tiles more
          // than one above the diagonal
          nd_launch_gemm(s, q, Lv, oe, M, oe, P, ob, oe, false, 2);
        } else if (s->lshape && P - oe > 0) {
          const bool big = (P - oe) >= 256;
          const int TS = big ? 128 : 64;
          const dim3 grid((unsigned)Lv.count, (unsigned)nd_lshape_tiles(TS, oe, M, P), 1);
          NdGatherCtx gc{};
          if (big)
            hipLaunchKernelGGL(k_nd_gemm8<false>, grid, dim3(512), 0, q, s->arena, Lv.woff, M, oe, M, oe, M, ob, oe, Lv.poff, P, gc, P, s->tile_order, 0);
          else
            hipLaunchKernelGGL((k_nd_gemm<2, false>), grid, dim3(256), 0, q, s->arena, Lv.woff, M, oe, M, oe, M, ob, oe, Lv.poff, P, gc, P, s->tile_order, 0);
        } else {
          nd_launch_gemm(s, q, Lv, oe, P, oe, P, ob, oe);
          nd_launch_gemm(s, q, Lv, oe, P, P, M, ob, oe);
          nd_launch_gemm(s, q, Lv, P, M, oe, P, ob, oe);
        }
        ob = oe;
      }
      if (B > 0) nd_launch_gemm(s, q, Lv, P, M, P, M, 0, P, cgather, (s->sym && cgather) ? 1 : 0);
    }
    nd_join(s, used);
  };
  auto grp = [&](int d, int sub) -> const pgx_nd::Group& { return s->groups[s->gfirst[d] + (sub < 0 ? 0 : sub)]; };
  int rcp = PGX_OK;
  const int kc = s->kcut;
  size_t dpn = 0;
  nd_dp_mark(s, dpn, -1, -1);
  if (kc >= 0) {
    // subtrees below the cut, one after the other; their roots' fronts (depth kc) wait in the other working buffer
    if ((rcp = prep(grp(kc, -1)))) return rcp;
    for (int g = 0; g < s->nsub; ++g) {
      for (int d = maxdepth; d > kc; --d) {
        const bool pc = use_gather && d < maxdepth;
        if (pc) {
          gather(grp(d, g));
        } else {
          if ((rcp = prep(grp(d, g)))) return rcp;
          if (d < maxdepth) extend(grp(d + 1, g));
        }
        eliminate(grp(d, g), pc);
        nd_dp_mark(s, dpn, 0, d);
      }
      extend(grp(kc + 1, g));
    }
    eliminate(grp(kc, -1));
    nd_dp_mark(s, dpn, 0, kc);
  }
  bool ahead = false;  // this depth's buffer has been prepared on prep_st
  for (int d = (kc >= 0 ? kc - 1 : maxdepth); d >= 0; --d) {
    const bool pc = use_gather && d < maxdepth;
    if (pc) {  // (nothing is prepared ahead in this mode: a front is written once, from its children)
      gather(grp(d, -1));
    } else {
      if (ahead)
        hipStreamWaitEvent(s->st, s->ev_prep_done, 0);
      else if ((rcp = prep(grp(d, -1))))
        return rcp;
      if (d < maxdepth) extend(grp(d + 1, -1));
    }
    ahead = false;
    if (!use_gather && d > 0 && s->prep_ahead && s->prep_st) {  // depth d + 1 has left the buffer depth d - 1 will use
      hipEventRecord(s->ev_prep_go, s->st);
      hipStreamWaitEvent(s->prep_st, s->ev_prep_go, 0);
      if ((rcp = prep_on(grp(d - 1, -1), s->prep_st))) return rcp;
      hipEventRecord(s->ev_prep_done, s->prep_st);
      ahead = true;
    }
    eliminate(grp(d, -1), pc);
    if (s->size > 1 && d == s->kdist) {  // Schur blocks of the subtree roots -> rank 0's ghost fronts
      const NdLevel& Lv = s->lev[s->kbatch];
      const int P = Lv.P, B = Lv.B, M = P + B;
      const size_t nb = (size_t)B * B;
      const unsigned pb = (unsigned)std::min<size_t>((nb + 255) / 256, 65535);
      if (s->rank != 0)
        hipLaunchKernelGGL(k_nd_pack, dim3(pb), dim3(256), 0, s->st, s->arena + s->fbase[s->root_slot], M, P, B, s->d_xbuf, 0);
      int rcx = s->comm->gather0(s->st, s->d_xbuf, nb, s->d_xbuf);
      if (rcx) {
        s->err = "distributed factorisation: " + s->comm->err;
        return rcx;
      }
      if (s->rank == 0)
        for (int j = 1; j < s->size; ++j)
          hipLaunchKernelGGL(k_nd_pack, dim3(pb), dim3(256), 0, s->st, s->arena + s->fbase[s->ghost_slot[j]], M, P, B,
                             s->d_xbuf + (size_t)j * nb, 1);
    }
    nd_dp_mark(s, dpn, 0, d);
  }
  nd_dp_collect(s, dpn);
  if (s->timing) {
    hipEventRecord(s->e1, s->st);
    hipEventSynchronize(s->e1);
    float ms = 0;
    hipEventElapsedTime(&ms, s->e0, s->e1);
    s->factor_ms += ms;
  }
  if (s->h_info) {
    NDHIP(hipMemcpyAsync(s->h_info, s->d_info, sizeof(int), hipMemcpyDeviceToHost, s->st));
    hipEventRecord(s->ev_info, s->st);
    s->info_pending = true;
  }
  NDHIP(hipGetLastError());
  if (!on_device) {
    NDHIP(hipStreamSynchronize(s->st));
    nd_poll_info(s, true);
  }
  s->factored = true;
  return PGX_OK;
}

extern "C" int pgx_nd_solve(pgx_nd* s, const double* b, double* x, int on_device) {
  if (!s || !b || !x) return PGX_EINVAL;
  if (s->device < 0) {
    s->err = "pgx_nd_solve: symbolic-only handle";
    return PGX_ENODEV;
  }
  if (!s->factored) {
    s->err = "pgx_nd_solve before pgx_nd_factor";
    return PGX_ESTATE;
  }
  NDHIP(hipSetDevice(s->device));
  PgxRange range("pgx_nd:solve");
  const double* db = b;
  double* dx = x;
  if (!on_device) {
    NDHIP(hipMemcpyAsync(s->d_b, b, (size_t)s->n * sizeof(double), hipMemcpyHostToDevice, s->st));
    db = s->d_b;
    dx = s->d_b;
  }
  if (s->timing) hipEventRecord(s->e0, s->st);
  const int maxdepth = (int)s->dfirst.size() - 2;
  size_t dpn = 0;
  nd_dp_mark(s, dpn, -1, -1);
  for (int d = maxdepth; d >= 0; --d) {  // forward: leaves to root; the batches of one depth run on forked streams
    hipEventRecord(s->ev_fork, s->st);
    int used = 0;
    for (int l = s->dfirst[d + 1] - 1; l >= s->dfirst[d]; --l) {
      const NdLevel& Lv = s->lev[l];
      const int P = Lv.P, B = Lv.B, M = P + B;
      const int64_t fs = (int64_t)M * P + (int64_t)P * B;
      if (Lv.count == 0) continue;
      hipStream_t q = nd_fork(s, used++);
      if (s->solve_small && P <= 64 && M <= 256) {  // one wave per front: assemble + triangle + border rows
#define ND_FS(NR)                                                                                                                    \
  hipLaunchKernelGGL(k_nd_fwd_small<NR>, dim3((unsigned)Lv.count), dim3(64), 0, q, s->arena, Lv.poff, fs, Lv.start, P, M, s->d_fp, s->d_fb, \
                     s->d_child0, s->d_child1, s->d_fP, s->d_vbase, s->d_dof_ptr, s->d_own_dofs, s->d_rel_ptr, s->d_rel, db, s->vec)
        if (M <= 64)
          ND_FS(1);
        else if (M <= 128)
          ND_FS(2);
        else
          ND_FS(4);
#undef ND_FS
        continue;
      }
      hipLaunchKernelGGL(k_nd_fwd_assemble, dim3((unsigned)Lv.count), dim3(256), 0, q, Lv.start, P, M, s->d_fp, s->d_fb,
                         s->d_child0, s->d_child1, s->d_fP, s->d_vbase, s->d_dof_ptr, s->d_own_dofs, s->d_rel_ptr, s->d_rel,
                         db, s->vec);
      // forward substitution in slabs of ND_SLAB pivots: triangle by one workgroup per front, everything below the slab
      // (rest of the pivot block AND the border rows) by a gemv over many workgroups
      for (int k0 = 0; k0 < P; k0 += ND_SLAB) {
        const int k1 = std::min(P, k0 + ND_SLAB);
        if (s->trsv_big && Lv.count <= ND_BIG_COUNT)
          hipLaunchKernelGGL(k_nd_trsv_big, dim3((unsigned)Lv.count), dim3(1024), ND_BIG_LDS, q, s->arena, Lv.poff, fs, s->vec, Lv.voff, M, k0, k1, 0);
        else
          hipLaunchKernelGGL(k_nd_trsv, dim3((unsigned)Lv.count), dim3(256), 0, q, s->arena, Lv.poff, fs, s->vec, Lv.voff, M, k0, k1, 0);
        if (k1 < M)
          hipLaunchKernelGGL(k_nd_gemv, dim3((unsigned)Lv.count, (unsigned)((M - k1 + ND_GR - 1) / ND_GR)), dim3(64 * ND_GW), 0, q, s->arena,
                             Lv.poff, fs, s->vec, Lv.voff, M, k1, M, k0, k1, (int64_t)k0 * M, M);
      }
    }
    nd_join(s, used);
    if (s->size > 1 && d == s->kdist && s->lev[s->kbatch].B > 0) {  // border contributions of the subtree roots -> ghosts
      const NdLevel& Lv = s->lev[s->kbatch];
      const int P = Lv.P, B = Lv.B, M = P + B;
      const int64_t vb = Lv.voff + (int64_t)(s->root_slot - Lv.start) * M + P;
      int rcx = s->comm->gather0(s->st, s->vec + vb, (size_t)B, s->d_vbuf);
      if (rcx) {
        s->err = "distributed solve: " + s->comm->err;
        return rcx;
      }
      if (s->rank == 0)
        for (int j = 1; j < s->size; ++j)
          NDHIP(hipMemcpyAsync(s->vec + Lv.voff + (int64_t)(s->ghost_slot[j] - Lv.start) * M + P, s->d_vbuf + (size_t)j * B,
                               sizeof(double) * B, hipMemcpyDeviceToDevice, s->st));
    }
    nd_dp_mark(s, dpn, 1, d);
  }
  for (int d = 0; d <= maxdepth; ++d) {  // backward: root to leaves
    const bool xchg = s->size > 1 && d == s->kdist && s->lev[s->kbatch].B > 0;
    if (xchg) {  // the subtree roots' border values: gathered from the parents on rank 0, then sent to their owners
      const NdLevel& Lv = s->lev[s->kbatch];
      const int P = Lv.P, B = Lv.B, M = P + B;
      if (Lv.count)
        hipLaunchKernelGGL(k_nd_bwd_gather, dim3((unsigned)Lv.count), dim3(256), 0, s->st, Lv.start, P, s->d_fb, s->d_parent,
                           s->d_vbase, s->d_rel_ptr, s->d_rel, s->vec);
      if (s->rank == 0)
        for (int j = 1; j < s->size; ++j)
          NDHIP(hipMemcpyAsync(s->d_vbuf + (size_t)j * B, s->vec + Lv.voff + (int64_t)(s->ghost_slot[j] - Lv.start) * M + P,
                               sizeof(double) * B, hipMemcpyDeviceToDevice, s->st));
      const int64_t vb = Lv.voff + (int64_t)(s->root_slot - Lv.start) * M + P;
      int rcx = s->comm->scatter0(s->st, s->d_vbuf, (size_t)B, s->vec + vb);
      if (rcx) {
        s->err = "distributed solve: " + s->comm->err;
        return rcx;
      }
    }
    hipEventRecord(s->ev_fork, s->st);
    int used = 0;
    for (int l = s->dfirst[d]; l < s->dfirst[d + 1]; ++l) {
      const NdLevel& Lv = s->lev[l];
      const int P = Lv.P, B = Lv.B, M = P + B;
      const int64_t fs = (int64_t)M * P + (int64_t)P * B;
      if (Lv.count == 0) continue;
      hipStream_t q = nd_fork(s, used++);
      if (s->solve_small && P <= 64 && M <= 256) {
        int pplog = 0;
        while ((1 << pplog) < P) ++pplog;
        const int gth = !(xchg && l == s->kbatch);
#define ND_BS(NR)                                                                                                                    \
  hipLaunchKernelGGL(k_nd_bwd_small<NR>, dim3((unsigned)Lv.count), dim3(64), 0, q, s->arena, Lv.poff, fs, Lv.start, P, M, pplog, s->d_fp, \
                     s->d_fb, s->d_parent, s->d_vbase, s->d_rel_ptr, s->d_rel, s->vec, gth)
        if (M <= 64)
          ND_BS(1);
        else if (M <= 128)
          ND_BS(2);
        else
          ND_BS(4);
#undef ND_BS
        continue;
      }
      if (B > 0) {
        if (!(xchg && l == s->kbatch))
          hipLaunchKernelGGL(k_nd_bwd_gather, dim3((unsigned)Lv.count), dim3(256), 0, q, Lv.start, P, s->d_fb, s->d_parent,
                             s->d_vbase, s->d_rel_ptr, s->d_rel, s->vec);
        hipLaunchKernelGGL(k_nd_gemv, dim3((unsigned)Lv.count, (unsigned)((P + ND_GR - 1) / ND_GR)), dim3(64 * ND_GW), 0, q, s->arena, Lv.poff,
                           fs, s->vec, Lv.voff, M, 0, P, P, M, (int64_t)M * P, P);
      }
      const int nsl = (P + ND_SLAB - 1) / ND_SLAB;
      for (int sl = nsl - 1; sl >= 0; --sl) {
        const int k0 = sl * ND_SLAB, k1 = std::min(P, k0 + ND_SLAB);
        if (s->trsv_big && Lv.count <= ND_BIG_COUNT)
          hipLaunchKernelGGL(k_nd_trsv_big, dim3((unsigned)Lv.count), dim3(1024), ND_BIG_LDS, q, s->arena, Lv.poff, fs, s->vec, Lv.voff, M, k0, k1, 1);
        else
          hipLaunchKernelGGL(k_nd_trsv, dim3((unsigned)Lv.count), dim3(256), 0, q, s->arena, Lv.poff, fs, s->vec, Lv.voff, M, k0, k1, 1);
        if (k0 > 0)
          hipLaunchKernelGGL(k_nd_gemv, dim3((unsigned)Lv.count, (unsigned)((k0 + ND_GR - 1) / ND_GR)), dim3(64 * ND_GW), 0, q, s->arena, Lv.poff,
                             fs, s->vec, Lv.voff, M, 0, k0, k0, k1, (int64_t)k0 * M, M);
      }
    }
    nd_join(s, used);
    nd_dp_mark(s, dpn, 2, d);
  }
  nd_dp_collect(s, dpn);
  if (s->size > 1) NDHIP(hipMemsetAsync(dx, 0, sizeof(double) * s->n, s->st));  // every rank writes its own dofs only
  hipLaunchKernelGGL(k_nd_write_x, dim3((unsigned)s->nfronts), dim3(128), 0, s->st, s->nfronts, s->d_fp, s->d_vbase,
                     s->d_dof_ptr, s->d_own_dofs, s->vec, dx);
  if (s->size > 1) {
    int rcx = s->comm->allreduce(s->st, dx, (size_t)s->n);
    if (rcx) {
      s->err = "distributed solve: " + s->comm->err;
      return rcx;
    }
  }
  if (s->timing) {
    hipEventRecord(s->e1, s->st);
    hipEventSynchronize(s->e1);
    float ms = 0;
    hipEventElapsedTime(&ms, s->e0, s->e1);
    s->solve_ms += ms;
  }
  NDHIP(hipGetLastError());
  if (!on_device) {
    NDHIP(hipMemcpyAsync(x, s->d_b, (size_t)s->n * sizeof(double), hipMemcpyDeviceToHost, s->st));
    NDHIP(hipStreamSynchronize(s->st));
  }
  nd_poll_info(s, !on_device);
  return PGX_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// symbolic export (tests)
// ------------------------------------------------------------------------------------------------------------------
extern "C" int pgx_nd_export_levels(const pgx_nd* s, int64_t* n_levels, int64_t* lev_start, int32_t* P, int32_t* B,
                                    int64_t* lev_off, int32_t* depth) {
  if (!s || !n_levels) return PGX_EINVAL;
  const int64_t L = (int64_t)s->lev.size();
  *n_levels = L;
  for (int64_t l = 0; l < L; ++l) {
    if (lev_start) lev_start[l] = s->lev[l].start;
    if (P) P[l] = s->lev[l].P;
    if (B) B[l] = s->lev[l].B;
    if (lev_off) lev_off[l] = s->lev[l].off;
    if (depth) depth[l] = s->lev[l].depth;
  }
  if (lev_start) lev_start[L] = s->nfronts;
  return PGX_OK;
}

extern "C" int pgx_nd_export_fronts(const pgx_nd* s, int64_t* n_fronts, int32_t* fp, int32_t* fb, int32_t* parent,
                                    int32_t* slot01, int64_t* dof_ptr, int32_t* own_dofs, int64_t* rel_ptr, int32_t* rel) {
  if (!s || !n_fronts) return PGX_EINVAL;
  if (s->device >= 0 && (own_dofs || rel)) return PGX_ESTATE;  // released after upload (see pgx_nd_export_dest)
  *n_fronts = s->nfronts;
  auto cp = [](auto* dst, const auto& v) {
    if (dst) std::copy(v.begin(), v.end(), dst);
  };
  cp(fp, s->fp);
  cp(fb, s->fb);
  cp(parent, s->parent);
  cp(slot01, s->slot01);
  cp(dof_ptr, s->dof_ptr);
  cp(own_dofs, s->own_dofs);
  cp(rel_ptr, s->rel_ptr);
  cp(rel, s->rel);
  return PGX_OK;
}

extern "C" int pgx_nd_export_dest(const pgx_nd* s, int64_t* nnz, int64_t* dest) {
  if (!s || !nnz) return PGX_EINVAL;
  if (s->device >= 0) return PGX_ESTATE;  // a device handle has released its host maps; export from a symbolic-only handle
  *nnz = s->nnz;
  if (dest) std::copy(s->dest.begin(), s->dest.end(), dest);
  return PGX_OK;
}
