"""Nested-dissection multifrontal sparse LU on the CPU (numpy + LAPACK/BLAS through scipy) - the oracle's `pc_type lu`.

TEST INFRASTRUCTURE: imported by tests/, tools/make_golden_*.py and bench.py's `cpu_baseline` leg only, never by the product.

What it restates.  The reference solves every Newton system exactly with a sparse direct solver:
`ksp_type preonly, pc_type lu, pc_factor_mat_solver_type mumps` (/root/reference/examples/01_obstacle_problem/obstacle_pg.py:129-131;
same dictionary in /root/reference/examples/06_gradient_constraints/gradient_constraint_dolfinx.py:118-121).  MUMPS is a multifrontal
LU on a nested-dissection-type ordering with partial pivoting INSIDE the fully summed block of each front.  That published algorithm
is what this file implements; `oracle/pg_oracle.py`'s default (SuperLU + COLAMD) is the same mathematics with a fill-in that grows
like N^3.2 on these saddle points (profiles/r02_cpu_ladder.json), which stopped the oracle at 512^2.  With this solver the oracle
reaches BASELINE config 2 itself (2048^2 P1, 8.4 M unknowns) and `cpu_baseline` is measured with an ordering a CPU user would choose.

Design (follows oracle/nd_proto.py, vectorised and blocked so that it scales):
* graph NODES = mesh entities; all dofs of a node (u first, then psi) are eliminated together.  For the Newton matrices
  [[aK, M],[M, -D(psi)]] every leading principal node block is nonsingular in any node order (symmetric quasi-definite argument,
  DESIGN.md section 9), so no pivot ever has to leave its front: `dgetrf` on the p x p pivot block is all the pivoting there is.
* ordering: recursive coordinate bisection; the separator is the set of nodes of ONE half adjacent to the other half, whichever of
  the two candidates carries fewer dofs.
* a front is kept as four Fortran-ordered blocks F11 (p x p), F12 (p x b), F21 (b x p), F22 (b x b) of one buffer; every numeric step
  is a LAPACK/BLAS call on whole blocks: dgetrf(F11), dlaswp + dtrsm(F12), dtrsm(F21), dgemm(F22 -= F21 F12).
* assembly destinations of every matrix entry and the children's positions in their parents are computed once per pattern.
"""
from __future__ import annotations

import time

import numpy as np
import scipy.sparse as sp
from scipy.linalg import blas, lapack


def _load_helper():
    """oracle/_build/libndhelper.so (oracle/csrc/nd_helper.c; compiled by __graft_entry__.build() or on first use with gcc).
    Without it the extend-add falls back to numpy fancy indexing - same results, several times slower."""
    import ctypes
    import os
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "_build", "libndhelper.so")
    src = os.path.join(here, "csrc", "nd_helper.c")
    try:
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            os.makedirs(os.path.dirname(so), exist_ok=True)
            subprocess.run(["gcc", "-O3", "-shared", "-fPIC", "-o", so, src], check=True)
        lib = ctypes.CDLL(so)
    except (OSError, subprocess.CalledProcessError):
        return None
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.nd_extend_add.argtypes = [vp, vp, vp, vp, i64, i64, vp, i64, vp, i64]
    lib.nd_extend_add.restype = None
    lib.nd_scatter.argtypes = [vp, vp, vp, vp, i64]
    lib.nd_scatter.restype = None
    return lib


_HELPER = _load_helper()


def _gather_rows(ptr, nodes):
    """Flat offsets of the CSR rows `nodes` and the start of each row in the flat array."""
    st = ptr[nodes]
    ln = ptr[nodes + 1] - st
    cs = np.cumsum(ln)
    starts = cs - ln
    off = np.repeat(st - starts, ln) + np.arange(int(cs[-1]) if len(cs) else 0)
    return off, starts


def _as_index(a):
    """A contiguous ascending index array becomes a slice (basic indexing: views instead of gather/scatter)."""
    if len(a) and int(a[-1]) - int(a[0]) + 1 == len(a):
        return slice(int(a[0]), int(a[-1]) + 1)
    return a


class NDLU:
    """Symbolic analysis of one sparsity pattern + repeated numeric factorisations / solves."""

    def __init__(self, A: sp.csr_matrix, node_of_dof, node_coords, leaf_nodes: int = 32, verbose: bool = False):
        t0 = time.perf_counter()
        A = A.tocsr()
        if not A.has_sorted_indices:
            A = A.sorted_indices()
        n = A.shape[0]
        self.n = n
        self._indptr = A.indptr.copy()
        self._indices = A.indices.copy()
        node_of_dof = np.asarray(node_of_dof, dtype=np.int64)
        node_coords = np.asarray(node_coords, dtype=np.float64)
        nn = int(node_of_dof.max()) + 1
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(A.indptr))
        gi, gj = node_of_dof[rows], node_of_dof[A.indices]
        key = np.unique(np.concatenate([gi * nn + gj, gj * nn + gi]))
        G = sp.csr_matrix((np.ones(len(key), dtype=np.int8), (key // nn, key % nn)), shape=(nn, nn))
        del key
        G.setdiag(1)
        G = G.tocsr()
        G.sort_indices()
        gp, ga = G.indptr.astype(np.int64), G.indices.astype(np.int64)
        ndof_of_node = np.bincount(node_of_dof, minlength=nn)

        # ---- dissection tree (ids parent-before-child) ----
        own, children = [], []
        side = np.zeros(nn, dtype=np.int8)
        stack = [(np.arange(nn, dtype=np.int64), -1)]
        while stack:
            V, par = stack.pop()
            tid = len(own)
            own.append(V)
            children.append([])
            if par >= 0:
                children[par].append(tid)
            if len(V) <= leaf_nodes:
                continue
            c = node_coords[V]
            ax = int(np.argmax(c.max(axis=0) - c.min(axis=0)))
            k = c[:, ax]
            med = np.partition(k, len(V) // 2)[len(V) // 2]
            inA = k < med
            if not inA.any():
                o = np.argsort(k, kind="stable")
                inA = np.zeros(len(V), dtype=bool)
                inA[o[: len(V) // 2]] = True
            Aset, Bset = V[inA], V[~inA]
            side[Bset] = 1
            off, st = _gather_rows(gp, Aset)
            sA = np.add.reduceat(side[ga[off]].astype(np.int32), st) > 0
            side[Bset] = 0
            side[Aset] = 1
            off, st = _gather_rows(gp, Bset)
            sB = np.add.reduceat(side[ga[off]].astype(np.int32), st) > 0
            side[Aset] = 0
            if ndof_of_node[Aset[sA]].sum() <= ndof_of_node[Bset[sB]].sum():
                own[tid], parts = Aset[sA], (Aset[~sA], Bset)
            else:
                own[tid], parts = Bset[sB], (Aset, Bset[~sB])
            for W in parts:
                if len(W):
                    stack.append((W, tid))
        nt = len(own)
        post = []
        st2 = [(0, 0)]
        while st2:
            t, k = st2.pop()
            if k < len(children[t]):
                st2.append((t, k + 1))
                st2.append((children[t][k], 0))
            else:
                post.append(t)
        post = np.array(post, dtype=np.int64)
        order_of = np.empty(nt, dtype=np.int64)
        order_of[post] = np.arange(nt)
        tnode = np.empty(nn, dtype=np.int64)
        node_pos = np.empty(nn, dtype=np.int64)
        k = 0
        for t in post:
            o = own[t]
            tnode[o] = t
            node_pos[o] = np.arange(k, k + len(o))
            k += len(o)
        depth = np.zeros(nt, dtype=np.int64)
        for t in range(nt):
            for c in children[t]:
                depth[c] = depth[t] + 1
        # ---- border (struct) node sets ----
        struct = [None] * nt
        node_ord = order_of[tnode]
        for t in post:
            off, _ = _gather_rows(gp, own[t])
            cand = [ga[off]] + [struct[c] for c in children[t]]
            cand = np.unique(np.concatenate(cand))
            cand = cand[node_ord[cand] > order_of[t]]
            struct[t] = cand[np.argsort(node_pos[cand], kind="stable")]
        # ---- dofs of nodes: ascending dof index inside a node (u before psi) ----
        dof_sort = np.argsort(node_of_dof, kind="stable")
        nd_ptr = np.concatenate(([0], np.cumsum(ndof_of_node)))

        def dofs(nodes):
            if not len(nodes):
                return np.zeros(0, dtype=np.int64)
            off, _ = _gather_rows(nd_ptr, nodes)
            return dof_sort[off]

        self.own_dofs = [dofs(own[t]) for t in range(nt)]
        self.struct_dofs = [dofs(struct[t]) for t in range(nt)]
        self.p = np.array([len(a) for a in self.own_dofs], dtype=np.int64)
        self.b = np.array([len(a) for a in self.struct_dofs], dtype=np.int64)
        self.post, self.children, self.depth, self.nt = post, children, depth, nt
        # ---- (front, dof) -> local position ----
        m = self.p + self.b
        fptr = np.concatenate(([0], np.cumsum(m)))
        all_idx = np.concatenate([np.concatenate([self.own_dofs[t], self.struct_dofs[t]]) for t in range(nt)])
        fkey = np.repeat(np.arange(nt, dtype=np.int64), m) * n + all_idx
        sorter = np.argsort(fkey, kind="stable")
        fkey_sorted = fkey[sorter]
        del fkey, all_idx

        def local(front, dof):
            pos = np.searchsorted(fkey_sorted, front * n + dof)
            return sorter[pos] - fptr[front]

        # children's border positions inside their parents
        self.cpos = [None] * nt
        for t in range(nt):
            for c in children[t]:
                lp = local(np.full(len(self.struct_dofs[c]), t, dtype=np.int64), self.struct_dofs[c])
                kk = int(np.searchsorted(lp, self.p[t]))
                assert np.all(np.diff(lp) > 0)
                self.cpos[c] = (_as_index(lp[:kk]), _as_index(lp[kk:] - self.p[t]), kk, np.ascontiguousarray(lp, dtype=np.int64))
        # ---- assembly destinations of the matrix entries ----
        dof_front = tnode[node_of_dof]  # front that eliminates each dof
        dof_ord = order_of[dof_front]
        cols = A.indices.astype(np.int64)
        fi, fj = dof_front[rows], dof_front[cols]
        f = np.where(dof_ord[rows] <= dof_ord[cols], fi, fj)
        del fi, fj
        li = local(f, rows)
        lj = local(f, cols)
        pf, bf = self.p[f], self.b[f]
        dst = np.where(li < pf,
                       np.where(lj < pf, li + lj * pf, pf * pf + li + (lj - pf) * pf),
                       pf * pf + pf * bf + (li - pf) + lj * bf)
        assert np.all((li < pf) | (lj < pf))
        del li, lj, pf, bf, rows, cols
        o = np.argsort(order_of[f], kind="stable")
        self.asm_src = o.astype(np.int64)
        self.asm_dst = dst[o].astype(np.int64)
        cnt = np.bincount(order_of[f], minlength=nt)
        self.asm_ptr = np.concatenate(([0], np.cumsum(cnt)))
        del f, dst, o
        self.flops = float(np.sum(2 / 3 * self.p**3 + 2 * self.p**2 * self.b + 2 * self.p * self.b**2))
        self.factor_entries = int(np.sum(self.p * self.p + 2 * self.p * self.b))
        self.max_front = int(m.max())
        self.symbolic_s = time.perf_counter() - t0
        self.fronts = None
        self.arena = None
        # numeric schedule: level by level, deepest first (the fronts of a level are independent); a level whose fronts are all
        # small runs with ONE BLAS thread (OpenBLAS's fork/join costs more than such a front's arithmetic), the others with all
        lv_order = np.argsort(-depth[post], kind="stable")
        self.sched = post[lv_order]
        self.sched_k = lv_order  # position of each scheduled front in the postorder-sorted assembly list
        dsorted = depth[self.sched]
        cuts = np.flatnonzero(np.diff(dsorted)) + 1
        self.level_ranges = list(zip(np.concatenate(([0], cuts)), np.concatenate((cuts, [nt]))))
        self.level_big = [bool(m[self.sched[a:b_]].max() >= 1024) for a, b_ in self.level_ranges]
        if verbose:
            print(f"NDLU symbolic: n={n} fronts={nt} depth={int(depth.max())} max front={self.max_front} "
                  f"factor entries={self.factor_entries / 1e9:.2f} G ({8 * self.factor_entries / 2**30:.1f} GiB) "
                  f"flops={self.flops / 1e9:.1f} GF in {self.symbolic_s:.1f} s", flush=True)

    # ------------------------------------------------------------------------------------------
    def same_pattern(self, A):
        return A.shape[0] == self.n and len(A.indices) == len(self._indices) and np.array_equal(A.indptr, self._indptr) \
            and np.array_equal(A.indices, self._indices)

    def factor(self, A: sp.csr_matrix, check_pattern: bool = True, workers: int = 0):
        """workers > 1: TREE-PARALLEL numeric phase - the subtrees hanging at tree depth ceil(log2(workers)) are factorised by
        forked worker processes (one BLAS thread each) into a SHARED arena, the levels above them by this process with all BLAS
        threads (what `mpirun -n N` + MUMPS does with a 2-D dissection: its fronts are too small for threaded BLAS alone to help,
        bench.py's `cpu_baseline.all_cores`).  Same algorithm as the serial schedule; the summation order inside threaded BLAS calls of
        the top levels may differ, so results agree to the solver's accuracy, not bitwise."""
        A = A.tocsr()
        if not A.has_sorted_indices:
            A = A.sorted_indices()
        if check_pattern and not self.same_pattern(A):
            raise ValueError("NDLU.factor: matrix pattern differs from the analysed one")
        data = A.data
        nt = self.nt
        upd = [None] * nt
        fronts = [None] * nt
        par = self._parallel_plan(workers) if workers and workers > 1 else None
        if par is not None:
            self.prepare_parallel(workers, touch=False)
        if self.arena is None:
            self.aoff = np.concatenate(([0], np.cumsum(self.p * self.p + 2 * self.p * self.b)))
            self.arena = np.zeros(int(self.aoff[-1]))
        else:
            self.arena.fill(0.0)
        src, dst, aptr = self.asm_src, self.asm_dst, self.asm_ptr
        if par is not None and hasattr(self, "pivs"):
            self._factor_parallel(par, workers, data, upd, fronts, src, dst, aptr)
            self.fronts = fronts
            return
        for (lo, hi), big in zip(self.level_ranges, self.level_big):
            with _threads(None if big else 1):
                self._factor_range(range(lo, hi), data, upd, fronts, src, dst, aptr)
        self.fronts = fronts

    def prepare_parallel(self, workers, touch=True):
        """The SHARED buffers of the tree-parallel factorisation (anonymous shared mappings: factors, pivots, the subtree roots' Schur
        blocks), allocated once; touch=True also writes every page (bench.py: factor storage is in place before the clock starts)."""
        par = self._parallel_plan(workers)
        if par is None or hasattr(self, "pivs"):
            return par
        import mmap

        self.aoff = np.concatenate(([0], np.cumsum(self.p * self.p + 2 * self.p * self.b)))
        self._mm = [mmap.mmap(-1, max(8, 8 * int(self.aoff[-1]))), mmap.mmap(-1, max(4, 4 * int(self.p.sum()))),
                    mmap.mmap(-1, max(8, 8 * int(par["uoff"][-1])))]
        self.arena = np.frombuffer(self._mm[0], dtype=np.float64)[: int(self.aoff[-1])]
        self.pivs = np.frombuffer(self._mm[1], dtype=np.int32)[: int(self.p.sum())]
        self.root_upd = np.frombuffer(self._mm[2], dtype=np.float64)[: int(par["uoff"][-1])]
        self.poff = np.concatenate(([0], np.cumsum(self.p)))
        if touch:
            self.arena.fill(1.0)
            self.root_upd.fill(1.0)
        return par

    def _parallel_plan(self, workers):
        """Cut of the elimination tree for `workers` processes: roots = the fronts at depth kc = ceil(log2(workers)) (fewer where a
        branch ends earlier), per-root schedules (deepest level first, as the serial schedule orders them), the schedule of the
        fronts above the cut, offsets of the roots' Schur blocks in the shared buffer.  None if the tree is too shallow."""
        if getattr(self, "_plan", None) is not None and self._plan["workers"] == workers:
            return self._plan
        kc = int(np.ceil(np.log2(workers)))
        depth = self.depth
        if int(depth.max()) < kc + 2:
            return None
        parent = np.full(self.nt, -1, dtype=np.int64)
        for t, ch in enumerate(self.children):
            for c in ch:
                parent[c] = t
        roots = [int(t) for t in range(self.nt) if depth[t] == kc]
        sub_of = np.full(self.nt, -1, dtype=np.int64)
        for i, r in enumerate(roots):
            sub_of[r] = i
        for t in sorted(range(self.nt), key=lambda t: depth[t]):  # parents before children
            if depth[t] > kc and parent[t] >= 0:
                sub_of[t] = sub_of[parent[t]]
        qsub = sub_of[self.sched]
        plan = {"workers": workers, "kc": kc, "roots": roots, "sub_q": [np.flatnonzero(qsub == i) for i in range(len(roots))],
                "top_q": np.flatnonzero(qsub < 0), "uoff": np.concatenate(([0], np.cumsum([int(self.b[r]) ** 2 for r in roots]))),
                "sub_fronts": [self.sched[np.flatnonzero(qsub == i)] for i in range(len(roots))]}
        self._plan = plan
        return plan

    def _front_views(self, t):
        p, b = int(self.p[t]), int(self.b[t])
        buf = self.arena[self.aoff[t]: self.aoff[t] + p * p + 2 * p * b]
        return (buf[: p * p].reshape((p, p), order="F"), self.pivs[self.poff[t]: self.poff[t] + p],
                buf[p * p: p * p + p * b].reshape((p, b), order="F"), buf[p * p + p * b:].reshape((b, p), order="F"))

    def _factor_parallel(self, par, workers, data, upd, fronts, src, dst, aptr):
        import os

        roots, uoff = par["roots"], par["uoff"]
        pending = list(range(len(roots)))
        running = {}
        failed = False
        while pending or running:
            while pending and len(running) < workers:
                i = pending.pop(0)
                pid = os.fork()
                if pid == 0:  # worker: its subtree, one BLAS thread, results into the shared buffers; never returns
                    code = 1
                    try:
                        with _threads(1):
                            self._factor_range(par["sub_q"][i], data, upd, fronts, src, dst, aptr, shared=True)
                        r = roots[i]
                        if upd[r] is not None:
                            self.root_upd[uoff[i]: uoff[i + 1]] = np.asarray(upd[r]).ravel(order="F")
                        code = 0
                    finally:
                        os._exit(code)
                running[pid] = i
            pid, status = os.wait()
            if pid in running:
                del running[pid]
                failed |= status != 0
        if failed:
            raise RuntimeError("NDLU: a worker of the tree-parallel factorisation failed (singular pivot block or out of memory)")
        for i, r in enumerate(roots):  # what the levels above need from below: the roots' Schur blocks; what the solves need: views
            b = int(self.b[r])
            upd[r] = self.root_upd[uoff[i]: uoff[i + 1]].reshape((b, b), order="F") if b else None
            for t in par["sub_fronts"][i]:
                fronts[int(t)] = self._front_views(int(t))
        with _threads(None):
            self._factor_range(par["top_q"], data, upd, fronts, src, dst, aptr, shared=True)

    def _factor_range(self, qs, data, upd, fronts, src, dst, aptr, shared=False):
        for q in qs:
            t, k = int(self.sched[q]), int(self.sched_k[q])
            p, b = int(self.p[t]), int(self.b[t])
            # the factor blocks live in one arena that is allocated once and reused by every factorisation (fresh pages cost more
            # than the arithmetic of a small front); the Schur block is a separate allocation that the parent frees
            buf = self.arena[self.aoff[t]: self.aoff[t] + p * p + 2 * p * b]
            s = slice(aptr[k], aptr[k + 1])
            buf[dst[s]] = data[src[s]]
            F11 = buf[: p * p].reshape((p, p), order="F")
            F12 = buf[p * p: p * p + p * b].reshape((p, b), order="F")
            F21 = buf[p * p + p * b:].reshape((b, p), order="F")
            F22 = np.zeros((b, b), order="F")
            for c in self.children[t]:
                ca, cb, kk, lp = self.cpos[c]
                U = upd[c]
                upd[c] = None
                if U is None:  # a child without border (a component the Dirichlet rows cut off): nothing to add
                    continue
                if _HELPER is not None:
                    _HELPER.nd_extend_add(F11.ctypes.data, F12.ctypes.data, F21.ctypes.data, F22.ctypes.data, p, b,
                                          U.ctypes.data, U.shape[0], lp.ctypes.data, kk)
                    continue
                if kk:
                    _add(F11, ca, ca, U[:kk, :kk])
                if kk and kk < U.shape[0]:
                    _add(F12, ca, cb, U[:kk, kk:])
                    _add(F21, cb, ca, U[kk:, :kk])
                if kk < U.shape[0]:
                    _add(F22, cb, cb, U[kk:, kk:])
            lu, piv, info = lapack.dgetrf(F11, overwrite_a=1)
            if info != 0:
                raise ZeroDivisionError(f"NDLU: singular pivot block in front {t} (dgetrf info {info})")
            if b:
                F12o, F21o = F12, F21
                F12 = lapack.dlaswp(F12, piv, overwrite_a=1)
                F12 = blas.dtrsm(1.0, lu, F12, side=0, lower=1, trans_a=0, diag=1, overwrite_b=1)
                F21 = blas.dtrsm(1.0, lu, F21, side=1, lower=0, trans_a=0, diag=0, overwrite_b=1)
                F22 = blas.dgemm(-1.0, F21, F12, beta=1.0, c=F22, overwrite_c=1)
                upd[t] = F22
                if shared:  # the factors must END UP in the arena (the wrappers work in place on these Fortran-ordered views; if one
                    if not np.shares_memory(F12, F12o):  # ever returned a copy, put it back)
                        F12o[...] = F12
                    if not np.shares_memory(F21, F21o):
                        F21o[...] = F21
            if shared:
                if not np.shares_memory(lu, F11):
                    F11[...] = lu
                self.pivs[self.poff[t]: self.poff[t] + p] = piv
                fronts[t] = self._front_views(t)
            else:
                fronts[t] = (lu, piv, F12, F21)

    def solve(self, rhs):
        with _threads(self.solve_threads):
            return self._solve(rhs)

    solve_threads = 1  # matrix-vector products of mostly small blocks: threading does not pay

    def _solve(self, rhs):
        x = np.array(rhs, dtype=np.float64)
        fr = self.fronts
        for t in self.post:
            lu, piv, F12, F21 = fr[t]
            od = self.own_dofs[t]
            y = lapack.dlaswp(x[od].reshape(-1, 1, order="F"), piv, overwrite_a=1)[:, 0]
            y = blas.dtrsv(lu, y, lower=1, trans=0, diag=1, overwrite_x=1)
            x[od] = y
            if F21.shape[0]:
                x[self.struct_dofs[t]] -= F21 @ y
        for t in self.post[::-1]:
            lu, piv, F12, F21 = fr[t]
            od = self.own_dofs[t]
            y = x[od]
            if F12.shape[1]:
                y = y - F12 @ x[self.struct_dofs[t]]
            x[od] = blas.dtrsv(lu, y, lower=0, trans=0, diag=0, overwrite_x=1)
        return x


class _threads:
    """BLAS thread limit for a block (threadpoolctl when present, otherwise a no-op); None = leave as is."""

    def __init__(self, n):
        self.n, self.ctx = n, None

    def __enter__(self):
        if self.n is not None:
            try:
                from threadpoolctl import threadpool_limits

                cap = MAX_THREADS if MAX_THREADS else self.n
                self.ctx = threadpool_limits(min(self.n, cap))
                self.ctx.__enter__()
            except ImportError:
                self.ctx = None
        return self

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


MAX_THREADS = 0  # 0 = whatever the BLAS uses by default; bench.py's cpu_baseline sets 1 (the reference runs OMP_NUM_THREADS=1)


def _add(F, r, c, U):
    if isinstance(r, slice) or isinstance(c, slice):
        F[r, c] += U  # at most one gathered axis
    else:
        F[np.ix_(r, c)] += U


class NDLinearSolve:
    """`linear_solve(J, rhs)` callback for pg_oracle.newton_solve: analyse once per pattern, factorise every call, solve with
    iterative refinement on the exact matrix (stops at a relative residual of 1e-15 or when it no longer improves)."""

    def __init__(self, node_of_dof, node_coords, leaf_nodes=32, verbose=False, max_refine=3, workers=0):
        self.node_of_dof, self.node_coords = node_of_dof, node_coords
        self.leaf_nodes, self.verbose, self.max_refine = leaf_nodes, verbose, max_refine
        self.workers = workers  # > 1: tree-parallel numeric factorisation (NDLU.factor)
        self.nd = None
        self.t_factor = self.t_solve = 0.0
        self.n_factor = 0
        self.last_relres = None

    def __call__(self, J, rhs):
        J = J.tocsr()
        if not J.has_sorted_indices:
            J = J.sorted_indices()
        if self.nd is None or not self.nd.same_pattern(J):
            self.nd = NDLU(J, self.node_of_dof, self.node_coords, self.leaf_nodes, self.verbose)
        t = time.perf_counter()
        self.nd.factor(J, check_pattern=False, workers=self.workers)
        t1 = time.perf_counter()
        x = self.nd.solve(rhs)
        bn = np.linalg.norm(rhs)
        r = rhs - J @ x
        rn = np.linalg.norm(r)
        for _ in range(self.max_refine):
            if rn <= 1e-15 * bn:
                break
            x2 = x + self.nd.solve(r)
            r2 = rhs - J @ x2
            rn2 = np.linalg.norm(r2)
            if rn2 >= rn:
                break
            x, r, rn = x2, r2, rn2
        self.last_relres = rn / bn if bn > 0 else 0.0
        self.t_factor += t1 - t
        self.t_solve += time.perf_counter() - t1
        self.n_factor += 1
        if self.verbose:
            print(f"  NDLU factor {t1 - t:.1f} s, solve+refine {time.perf_counter() - t1:.1f} s, relres {self.last_relres:.1e}", flush=True)
        return x


def nodes_of_problem(prob):
    """Node grouping of the oracle's problems.  Obstacle (pg_oracle): dof i of u and dof i of psi share node i (vertex or edge
    midpoint).  Gradient constraint (gc_oracle.GradientConstraintP2, layout u[n2] | psi_x[nv] | psi_y[nv]): a vertex node holds
    (u_v, psi_x_v, psi_y_v), an edge node its P2 midpoint value."""
    if hasattr(prob, "cverts") and hasattr(prob, "npsi"):  # sg_oracle.SignoriniP1: u_x | u_y | u_z per vertex, psi per contact vertex
        nv = prob.nv
        return np.concatenate([np.arange(nv), np.arange(nv), np.arange(nv), np.asarray(prob.cverts)]), np.asarray(prob.coords)
    if hasattr(prob, "n2") and hasattr(prob, "ntot"):
        nv, n2 = prob.nv, prob.n2
        return np.concatenate([np.arange(n2), np.arange(nv), np.arange(nv)]), np.asarray(prob.dof_coords)[:n2]
    n = prob.n
    coords = getattr(prob, "dof_coords", prob.coords)
    return np.concatenate([np.arange(n), np.arange(n)]), np.asarray(coords)[:n]
