#!/usr/bin/env python3
"""Mid-size goldens of examples 06, 02 (degrees 1 and 2) and 01-P2 from the CPU oracles (SuperLU; minutes in the build container):
final primal field + per-step Newton counts, at sizes where the GPU path's sparse LU works on a deep dissection tree with several
size classes per depth (the oracle-compared full runs in tests/test_gpu_*.py stop at N = 20 / 8x6x5 / 32 because the oracle runs
inside the GPU test).  Generated from the ORACLE (parity unpinned, see the oracle headers).
    python tools/make_golden_families.py [gc N] [sg n] [sg2 n] [p2 N] [hexdefault D] [ic N] ...      default: gc 64  sg 14  sg2 8  p2 96
`ic N`: example 08 (oracle/ic_oracle.py) at the reference's size N = 1001 - the whole continuation in phic: final state, LVPP and
Newton counts per phic and the log of every attempt (phic, k, alpha, Newton steps, SNES reason)."""
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import gc_oracle as G6  # noqa: E402
from oracle import pg_oracle as O  # noqa: E402
from oracle import sg_oracle as S  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def gc(N):
    c, e = O.create_rectangle(N, N, (0.0, 0.0), (1.0, 1.0))
    p = G6.GradientConstraintP2(c, e)
    x, its = G6.solve_problem(p)[:2]
    np.savez_compressed(GOLD / f"gradient_constraint_p2_n{N}_defaults_mid.npz", N=N, u_final=x[:p.n2], newton=np.asarray(its))
    return its


def sg(n, degree):
    coords, cells = S.create_unit_cube_tets(n, n, n)
    cf = S.boundary_facets_where(coords, cells, lambda c: np.isclose(c[:, 2], 0.0))
    tf = S.boundary_facets_where(coords, cells, lambda c: np.isclose(c[:, 2], 1.0))
    if degree == 1:
        p = S.SignoriniP1(coords, cells, cf, np.flatnonzero(np.isclose(coords[:, 2], 1.0)))
    else:
        p = S.SignoriniP2(coords, cells, cf, tf)
    x, it, its = S.solve_contact_problem(p)
    np.savez_compressed(GOLD / f"signorini_p{degree}_n{n}_defaults_mid.npz", n=n, degree=degree, u_final=x[:3 * p.nv], newton=np.asarray(its), it=it)
    return its


def sghex(nx, ny, nz, degree):
    p = S.SignoriniHex(nx, ny, nz, degree=degree)
    x, it, its = S.solve_contact_problem(p)
    np.savez_compressed(GOLD / f"signorini_hex_q{degree}_{nx}x{ny}x{nz}_defaults_mid.npz", n=np.array([nx, ny, nz]), degree=degree,
                        u_final=x[:3 * p.nv], newton=np.asarray(its), it=it)
    return its


def ic(N):
    from oracle import ic_oracle as I8

    p = I8.Intersecting(N)
    z, n_lvpp, n_newton, log = I8.solve_problem(p)
    np.savez_compressed(GOLD / f"intersecting_n{N}.npz", N=N, z_final=z, lvpp=np.asarray(n_lvpp), newton=np.asarray(n_newton),
                        attempts=np.asarray([[r[0], r[1], r[2], r[3], r[4]] for r in log], dtype=np.float64))
    return n_newton


def p2(N):
    coords, cells = O.create_rectangle(N, N)
    p = O.ObstacleLagrange(coords, cells, degree=2)
    x, hist = O.solve_problem(p, 500, "double_exponential", 1e2, 1e-4)
    np.savez_compressed(GOLD / f"obstacle_p2_n{N}_settingsB_mid.npz", N=N, u_final=x[:p.n], newton=np.asarray(hist["Newton steps"]))
    return hist["Newton steps"]


if __name__ == "__main__":
    a = sys.argv[1:] or ["gc", "64", "sg", "14", "sg2", "8", "p2", "96"]
    for kind, size in zip(a[::2], a[1::2]):
        t = time.perf_counter()
        its = {"gc": gc, "sg": lambda n: sg(n, 1), "sg2": lambda n: sg(n, 2), "p2": p2,
               "hexdefault": lambda d: sghex(16, 7, 5, d), "ic": ic}[kind](int(size))  # hexdefault D: the reference's native mesh, degree D
        print(f"{kind} {size}: Newton {list(its)}  ({time.perf_counter() - t:.1f} s)", flush=True)
