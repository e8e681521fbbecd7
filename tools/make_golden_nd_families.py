#!/usr/bin/env python3
"""Goldens of examples 06 and 02 at the sizes of BASELINE configs 4 and 5, from the CPU oracles with the nested-dissection LU.

    python tools/make_golden_nd_families.py gc:1024        (build container, 8 cores: hours)
    python tools/make_golden_nd_families.py sg:70          (build container: about an hour)

`gc:N`  oracle/gc_oracle.py (the loop of /root/reference/examples/06_gradient_constraints/gradient_constraint_dolfinx.py:100-132,
        171-205: primal P2, latent vector-P1, doubling alpha, SNES rtol = atol = stol = 1e-9) on the N x N unit square;
`sg:n`  oracle/sg_oracle.py (/root/reference/examples/02_signorini/signorini_dolfinx.py:244-291,317-358: P1 elasticity + contact
        latent, doubling alpha, Newton tolerance 1e-6) on n^3 x 6 tetrahedra;
both with oracle/nd_lu.py as `pc_type lu` (iterative refinement on the exact matrix).  A fixture holds

* the per-proximal-step Newton counts (and `it`, the number of proximal steps, for example 02), the increment column of example 06;
* a FINGERPRINT of the final primal field in which every dof takes part: its values on a sub-lattice of the vertices, the sums over
  the stride^d vertex blocks, the sums over contiguous chunks of the dofs that are not vertices (edge midpoints of example 06), the
  2-norm, the maximum and the minimum (tools/make_golden_nd.py's format, extended to P2 dofs and three components);
* the largest relative residual the direct solver left (`lin_relres_max`).

tests/test_gpu_golden_families.py compares the HIP path with these.  Generated from the ORACLE (parity unpinned w.r.t. FEniCSx).
"""
import json
import pathlib
import platform
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import nd_lu as ND  # noqa: E402
from oracle import pg_oracle as O  # noqa: E402

GOLD = ROOT / "tests" / "golden"
LADDER = ROOT / "profiles" / "r05_cpu_golden_runs_families.json"


def chunk_sums(v, chunk):
    """sums over contiguous chunks of `chunk` entries (the tail chunk zero-padded)"""
    m = -(-v.size // chunk) * chunk
    w = np.zeros(m)
    w[: v.size] = v
    return w.reshape(-1, chunk).sum(axis=1)


def lattice_fingerprint(g, stride):
    """g: (M, ..., M[, c]) lattice values over d = 2 or 3 axes, M = k * stride + 1.  Returns (sample, blocksum)."""
    d = 3 if g.ndim >= 3 and g.shape[2] == g.shape[0] else 2
    M = g.shape[0]
    nb = (M - 1) // stride
    sl = (slice(None, None, stride),) * d
    sample = g[sl].copy()
    core = g[(slice(0, nb * stride),) * d]
    shp = []
    for _ in range(d):
        shp += [nb, stride]
    core = core.reshape(*shp, *g.shape[d:])
    return sample, core.sum(axis=tuple(range(1, 2 * d, 2)))


def _record(rec):
    doc = json.loads(LADDER.read_text()) if LADDER.exists() else {"what": "oracle runs behind tests/golden/*_nd.npz of examples 06 / 02 "
                                                                   "(tools/make_golden_nd_families.py; build container, 8 cores)", "runs": []}
    doc["runs"] = [r for r in doc["runs"] if (r["kind"], r["N"]) != (rec["kind"], rec["N"])] + [rec]
    LADDER.write_text(json.dumps(doc, indent=1))


def wrap(ls):
    relres = []

    def solve(J, rhs):
        x = ls(J, rhs)
        relres.append(ls.last_relres)
        return x

    return solve, relres


def gc(N):
    from oracle import gc_oracle as G

    c, e = O.create_rectangle(N, N, (0.0, 0.0), (1.0, 1.0))
    t0 = time.perf_counter()
    p = G.GradientConstraintP2(c, e)
    print(f"ex 06 {N}^2: problem built in {time.perf_counter() - t0:.1f} s, {p.ntot} unknowns", flush=True)
    ls = ND.NDLinearSolve(*ND.nodes_of_problem(p), verbose=True)
    solve, relres = wrap(ls)
    t0 = time.perf_counter()
    x, its, diffs = G.solve_problem(p, linear_solve=solve, verbose=True)
    wall = time.perf_counter() - t0
    M = N + 1
    stride = max(1, N // 256)
    u = x[: p.n2]
    sample, blocksum = lattice_fingerprint(u[: p.nv].reshape(M, M), stride)
    out = GOLD / f"gradient_constraint_p2_n{N}_defaults_nd.npz"
    np.savez_compressed(out, N=N, stride=stride, u_sample=sample, u_blocksum=blocksum, edge_chunk=64,
                        u_edge_chunksum=chunk_sums(u[p.nv:], 64), u_norm2=float(np.linalg.norm(u)),
                        u_vertex_norm2=float(np.linalg.norm(u[: p.nv])), u_max=float(u.max()), u_min=float(u.min()),
                        psi_absmax=float(np.abs(x[p.n2:]).max()), lin_relres_max=float(max(relres)), newton=np.asarray(its),
                        l2_increments=np.asarray(diffs))
    rec = {"kind": "gc", "N": N, "unknowns": int(p.ntot), "newton_steps": int(np.sum(its)), "wall_s": wall, "symbolic_s": ls.nd.symbolic_s,
           "s_per_newton_step": (wall - ls.nd.symbolic_s) / int(np.sum(its)), "t_factor": ls.t_factor, "t_solve_refine": ls.t_solve,
           "factor_flops": ls.nd.flops, "factor_entries": int(ls.nd.factor_entries), "lin_relres_max": float(max(relres)),
           "host": platform.processor() or platform.machine()}
    print(json.dumps(rec), flush=True)
    _record(rec)
    print(f"wrote {out} ({out.stat().st_size} bytes)", flush=True)


def sg(n):
    from oracle import sg_oracle as S

    coords, cells = S.create_unit_cube_tets(n, n, n)
    cf = S.boundary_facets_where(coords, cells, lambda c: np.isclose(c[:, 2], 0.0))
    t0 = time.perf_counter()
    p = S.SignoriniP1(coords, cells, cf, np.flatnonzero(np.isclose(coords[:, 2], 1.0)))
    print(f"ex 02 {n}^3: problem built in {time.perf_counter() - t0:.1f} s, {p.ntot} unknowns", flush=True)
    ls = ND.NDLinearSolve(*ND.nodes_of_problem(p), verbose=True)
    solve, relres = wrap(ls)
    t0 = time.perf_counter()
    x, it, its = S.solve_contact_problem(p, linear_solve=solve, verbose=True)
    wall = time.perf_counter() - t0
    M = n + 1
    stride = 2 if n % 2 == 0 else 1
    nv = p.nv
    u = x[: 3 * nv]
    # layout of the primal block: see how the test reads it back - kept FLAT here plus the lattice view of each component
    comp = np.stack([u[k * nv:(k + 1) * nv].reshape(M, M, M) for k in range(3)], axis=-1) if _is_blocked(p) else u.reshape(M, M, M, 3)
    sample, blocksum = lattice_fingerprint(comp, stride)
    out = GOLD / f"signorini_p1_n{n}_defaults_nd.npz"
    np.savez_compressed(out, n=n, degree=1, stride=stride, blocked=int(_is_blocked(p)), u_sample=sample, u_blocksum=blocksum, chunk=16,
                        u_chunksum=chunk_sums(u, 16), u_norm2=float(np.linalg.norm(u)), u_max=float(u.max()), u_min=float(u.min()),
                        psi_min=float(x[3 * nv:].min()), psi_max=float(x[3 * nv:].max()), lin_relres_max=float(max(relres)),
                        newton=np.asarray(its), it=it)
    rec = {"kind": "sg", "N": n, "unknowns": int(p.ntot), "newton_steps": int(np.sum(its)), "wall_s": wall, "symbolic_s": ls.nd.symbolic_s,
           "s_per_newton_step": (wall - ls.nd.symbolic_s) / int(np.sum(its)), "t_factor": ls.t_factor, "t_solve_refine": ls.t_solve,
           "factor_flops": ls.nd.flops, "factor_entries": int(ls.nd.factor_entries), "lin_relres_max": float(max(relres)),
           "host": platform.processor() or platform.machine()}
    print(json.dumps(rec), flush=True)
    _record(rec)
    print(f"wrote {out} ({out.stat().st_size} bytes)", flush=True)


def _is_blocked(p):
    """True if the primal block is u_x | u_y | u_z (component-major), False if interleaved per vertex."""
    nod, _ = ND.nodes_of_problem(p)
    return bool(nod[1] == 1)


if __name__ == "__main__":
    for a in sys.argv[1:]:
        kind, _, size = a.partition(":")
        {"gc": gc, "sg": sg}[kind](int(size))
