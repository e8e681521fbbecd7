"""Prototype (numpy/scipy) of the two-level preconditioner of the P2 obstacle systems that pgx_patch.hip implements on the GPU.
TEST INFRASTRUCTURE / design study only - never imported by the product.

What it is for.  The reference solves the P2 Newton systems of `obstacle_pg.py -p 2` with a sparse direct solver
(/root/reference/examples/01_obstacle_problem/obstacle_pg.py:68-70,129-131,288).  The HIP path's alternative - FGMRES with a
two-level cycle, P2 level + P1 hierarchy - needs a smoother that survives the late proximal steps, where e^psi spans tens of orders
of magnitude inside single elements: the collective point-Jacobi smoother of rounds 1-2 needs 30-90 iterations at 64^2-128^2 and
hundreds at 256^2.  This file measures the remedy VERDICT r02 item 4 names: an additive VERTEX-STAR Schwarz smoother - for every
vertex the (u, psi) dofs on the vertex and on the edges that meet in it, 2 (1 + deg) <= 14 unknowns on the right-diagonal mesh,
solved exactly, overlapping corrections averaged - with the P1 space (exact Galerkin coarse solve here) as coarse level.

Measured with nu = 2 sweeps, omega = 1 (python oracle/p2_patch_proto.py N:nu:omega; settings B, every Newton system of the run;
FGMRES to 1e-10):  16^2: 7-11 iterations, 32^2: 7-15, 64^2: 8-12 (last 8 systems), 128^2: 8-18, 256^2: 8-15 on the regular systems and 34 / 56 on
the two OVERSHOT iterates that precede that run's Newton divergence (the divergence is the reference's algorithm, DESIGN.md section 3;
on such iterates the HIP path falls back to its sparse LU for the rest of the Newton solve: pgx_api.hip, pgx_newton_solve).
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pg_oracle as O, nd_lu as ND
from oracle.krylov_proto import fgmres

def p1_to_p2(prob):
    nv, ne = prob.nv, len(prob.edges)
    rows = np.concatenate([np.arange(nv), nv+np.arange(ne), nv+np.arange(ne)])
    cols = np.concatenate([np.arange(nv), prob.edges[:,0], prob.edges[:,1]])
    vals = np.concatenate([np.ones(nv), 0.5*np.ones(ne), 0.5*np.ones(ne)])
    return sp.csr_matrix((vals,(rows,cols)), shape=(nv+ne, nv))

class StarSmoother:
    def __init__(self, prob, J, mode="vertex"):
        n = prob.n; nv = prob.nv; ne = len(prob.edges)
        inc = [[] for _ in range(nv)]
        for e,(a,b) in enumerate(prob.edges):
            inc[a].append(e); inc[b].append(e)
        maxd = max(len(x) for x in inc)
        P = 2*(1+maxd)
        idx = np.full((nv, P), -1, dtype=np.int64)
        for v in range(nv):
            d = [v] + [nv+e for e in inc[v]]
            d = d + [n+q for q in d]
            idx[v,:len(d)] = d
        self.idx = idx
        J = J.tocsr()
        Ainv = np.zeros((nv,P,P))
        for v in range(nv):
            d = idx[v][idx[v]>=0]
            A = J[d][:,d].toarray()
            k=len(d)
            M = np.eye(P); M[:k,:k]=A
            Ainv[v] = np.linalg.inv(M)
        self.Ainv = Ainv
        mult = np.bincount(idx[idx>=0], minlength=2*n).astype(float)
        self.w = 1.0/np.maximum(mult,1)
        self.J = J
        self.n2 = 2*n
    def sweep(self, x, b, omega=1.0):
        r = b - self.J@x
        rp = np.where(self.idx>=0, r[np.maximum(self.idx,0)], 0.0)
        dp = np.einsum('pij,pj->pi', self.Ainv, rp)
        dx = np.zeros(self.n2)
        m = self.idx>=0
        np.add.at(dx, self.idx[m], dp[m])
        return x + omega*self.w*dx

LASTK = 6
def run(N, nu=2, omega=1.0, maxit=200):
    coords, cells = O.create_rectangle(N,N)
    prob = O.ObstacleLagrange(coords, cells, degree=2)
    n = prob.n
    systems=[]
    ls = ND.NDLinearSolve(*ND.nodes_of_problem(prob), leaf_nodes=16)
    def rec(J, b):
        systems.append((J.copy(), b.copy()))
        return ls(J,b)
    try:
        x,h = O.solve_problem(prob, 100, "double_exponential", 1e2, 1e-4, linear_solve=rec)
        print("N",N,"newton",h["Newton steps"], flush=True)
    except RuntimeError as e:
        print("N",N,"diverged:",e, "systems", len(systems), flush=True)
    T1 = p1_to_p2(prob)
    T = sp.block_diag([T1,T1]).tocsr()
    for k in range(max(0, len(systems) - LASTK), len(systems)):
        J,b = systems[k]
        t=time.perf_counter()
        sm = StarSmoother(prob, J)
        Jc = (T.T@J@T).tocsc()
        luc = spla.splu(Jc)
        def prec(r):
            x = np.zeros_like(r)
            for _ in range(nu): x = sm.sweep(x, r, omega)
            x = x + T@luc.solve(T.T@(r - J@x))
            for _ in range(nu): x = sm.sweep(x, r, omega)
            return x
        xs, its, hist = fgmres(J, b, prec, 1e-10, maxit)
        print(f"  N={N} system {k:2d}: its {its:3d} final {hist[-1]:.1e}  ({time.perf_counter()-t:.1f}s)", flush=True)

if __name__=="__main__":
    for a in sys.argv[1:]:
        N,nu,om = a.split(":")
        run(int(N), int(nu), float(om))
