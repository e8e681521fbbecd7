"""Prototype (numpy) of the geometric nested-dissection multifrontal LU that proximalgalerkin_amd/csrc/pgx_nd.hip
implements on the GPU.  TEST INFRASTRUCTURE / design study only - never imported by the product.

Replaces what the reference asks of PETSc: `ksp_type preonly, pc_type lu, pc_factor_mat_solver_type mumps`
(/root/reference/examples/01_obstacle_problem/obstacle_pg.py:129-131,
 /root/reference/examples/06_gradient_constraints/gradient_constraint_dolfinx.py:118-121).

Design
------
* graph NODES = mesh entities (vertex / edge midpoint); all dofs of one entity (u, psi[, psi_y]) are eliminated together,
  u first.  For the Newton matrices of this repo every leading principal NODE-block is nonsingular (K_SS, M_SS are SPD
  principal submatrices; the latent block is negative semi-definite), so an LU WITHOUT pivoting across nodes exists for
  any node ordering (symmetric quasi-definite argument); tiny pivots are replaced by a static perturbation and the
  factorisation is used inside iterative refinement / FGMRES on the exact operator.
* ordering = recursive coordinate bisection; separator = the nodes of the lower half adjacent to the upper half.
* one frontal matrix per tree node: [own dofs | struct dofs], struct = ancestors' dofs coupled to the subtree.
* numeric phase per tree LEVEL (deepest first): assemble original entries, extend-add the children's update matrices,
  factor the pivot block, update.  All fronts of a level are independent (batched on the GPU).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


class Tree:
    pass


def build_tree(A: sp.csr_matrix, node_of_dof, node_coords, leaf_nodes=48):
    n = A.shape[0]
    node_of_dof = np.asarray(node_of_dof)
    nn = int(node_of_dof.max()) + 1
    # node graph
    Ac = A.tocoo()
    gi, gj = node_of_dof[Ac.row], node_of_dof[Ac.col]
    G = sp.csr_matrix((np.ones(len(gi), dtype=np.int8), (gi, gj)), shape=(nn, nn))
    G = ((G + G.T) > 0).astype(np.int8).tocsr()
    gp, ga = G.indptr, G.indices
    own, children = [], []
    side = np.zeros(nn, dtype=np.int8)

    def rec(V):
        tid = len(own)
        own.append(None)
        children.append([])
        if len(V) <= leaf_nodes:
            own[tid] = V
            return tid
        c = node_coords[V]
        ext = c.max(axis=0) - c.min(axis=0)
        ax = int(np.argmax(ext))
        key = c[:, ax]
        med = np.partition(key, len(V) // 2)[len(V) // 2]
        inA = key < med
        if not inA.any() or inA.all():
            o = np.argsort(key, kind="stable")
            inA = np.zeros(len(V), dtype=bool)
            inA[o[: len(V) // 2]] = True
        Aset, Bset = V[inA], V[~inA]
        side[Bset] = 1
        # separator: nodes of A adjacent to B
        isS = np.zeros(len(Aset), dtype=bool)
        for k, g in enumerate(Aset):
            nb = ga[gp[g]: gp[g + 1]]
            if (side[nb] == 1).any():
                isS[k] = True
        side[Bset] = 0
        own[tid] = Aset[isS]
        Ain = Aset[~isS]
        for W in (Ain, Bset):
            if len(W):
                children[tid].append(rec(W))
        return tid

    import sys
    sys.setrecursionlimit(10000)
    rec(np.arange(nn))
    T = Tree()
    nt = len(own)
    # postorder
    post = []
    stack = [(0, 0)]
    while stack:
        t, k = stack.pop()
        if k < len(children[t]):
            stack.append((t, k + 1))
            stack.append((children[t][k], 0))
        else:
            post.append(t)
    order_of = np.empty(nt, dtype=np.int64)
    order_of[post] = np.arange(nt)
    tnode = np.empty(nn, dtype=np.int64)
    for t in range(nt):
        tnode[own[t]] = t
    # elimination position of nodes, dofs of node in increasing dof index
    dof_sort = np.argsort(node_of_dof, kind="stable")
    nd_ptr = np.concatenate(([0], np.cumsum(np.bincount(node_of_dof, minlength=nn))))
    perm = []  # new -> old dof
    node_pos = np.empty(nn, dtype=np.int64)
    k = 0
    for t in post:
        for g in own[t]:
            node_pos[g] = k
            k += 1
            perm.extend(dof_sort[nd_ptr[g]: nd_ptr[g + 1]])
    perm = np.array(perm)
    iperm = np.empty(n, dtype=np.int64)
    iperm[perm] = np.arange(n)
    struct = [None] * nt
    for t in post:
        cand = [ga[gp[g]: gp[g + 1]] for g in own[t]] + [struct[c] for c in children[t]]
        cand = np.unique(np.concatenate(cand)) if cand else np.zeros(0, dtype=np.int64)
        cand = cand[order_of[tnode[cand]] > order_of[t]]
        struct[t] = cand[np.argsort(node_pos[cand])]
    depth = np.zeros(nt, dtype=np.int64)
    for t in range(nt):  # ids are assigned parent-before-child
        for c in children[t]:
            depth[c] = depth[t] + 1
    T.own, T.children, T.struct, T.post, T.depth = own, children, struct, post, depth
    T.perm, T.iperm, T.nd_ptr, T.dof_sort, T.node_pos = perm, iperm, nd_ptr, dof_sort, node_pos
    T.n = n

    def dofs(nodes):
        return np.concatenate([dof_sort[nd_ptr[g]: nd_ptr[g + 1]] for g in nodes]) if len(nodes) else np.zeros(0, dtype=np.int64)

    T.own_dofs = [dofs(own[t]) for t in range(nt)]
    T.struct_dofs = [dofs(struct[t]) for t in range(nt)]
    return T


def lu_nopivot(F11, tiny):
    """In-place LU without pivoting; pivots below `tiny` in magnitude are replaced (static pivoting). Returns #replaced."""
    p = F11.shape[0]
    rep = 0
    for k in range(p):
        d = F11[k, k]
        if abs(d) < tiny:
            d = tiny if d >= 0 else -tiny
            F11[k, k] = d
            rep += 1
        F11[k + 1:, k] /= d
        F11[k + 1:, k + 1:] -= np.outer(F11[k + 1:, k], F11[k, k + 1:])
    return rep


def factor(T: Tree, A: sp.csr_matrix, piv_eps=1e-13):
    A = A.tocsr()
    nt = len(T.own)
    fronts = [None] * nt
    upd = [None] * nt
    pos = np.full(T.n, -1, dtype=np.int64)
    nrep = 0
    anorm = abs(A).max()
    stats = dict(flops=0.0, front_entries=0, max_front=0)
    for t in T.post:
        od, sd = T.own_dofs[t], T.struct_dofs[t]
        idx = np.concatenate([od, sd])
        p, m = len(od), len(idx)
        F = np.zeros((m, m))
        sub = A[idx][:, idx].toarray()
        F[:p, :] = sub[:p, :]
        F[p:, :p] = sub[p:, :p]
        pos[idx] = np.arange(m)
        for c in T.children[t]:
            ci = pos[T.struct_dofs[c]]
            assert (ci >= 0).all()
            F[np.ix_(ci, ci)] += upd[c]
            upd[c] = None
        pos[idx] = -1
        F11 = F[:p, :p]
        nrep += lu_nopivot(F11, piv_eps * anorm)
        L = np.tril(F11, -1) + np.eye(p)
        U = np.triu(F11)
        F[:p, p:] = np.linalg.solve(L, F[:p, p:])            # U12
        F[p:, :p] = np.linalg.solve(U.T, F[p:, :p].T).T      # L21
        upd[t] = F[p:, p:] - F[p:, :p] @ F[:p, p:]
        fronts[t] = (F[:p, :].copy(), F[p:, :p].copy())
        stats["flops"] += 2 / 3 * p**3 + 2 * p * p * (m - p) + 2 * p * (m - p) ** 2
        stats["front_entries"] += m * m
        stats["max_front"] = max(stats["max_front"], m)
    stats["replaced_pivots"] = nrep
    return fronts, stats


def solve(T: Tree, fronts, b):
    x = b.astype(float).copy()
    for t in T.post:
        od, sd = T.own_dofs[t], T.struct_dofs[t]
        top, L21 = fronts[t]
        p = len(od)
        L = np.tril(top[:, :p], -1) + np.eye(p)
        y = np.linalg.solve(L, x[od])
        x[od] = y
        x[sd] -= L21 @ y
    for t in reversed(T.post):
        od, sd = T.own_dofs[t], T.struct_dofs[t]
        top, _ = fronts[t]
        p = len(od)
        U = np.triu(top[:, :p])
        x[od] = np.linalg.solve(U, x[od] - top[:, p:] @ x[sd])
    return x
