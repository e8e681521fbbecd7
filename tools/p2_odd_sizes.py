#!/usr/bin/env python3
"""P2 example 01 on tiny, odd and strip-shaped meshes, HIP path (defaults: patch-smoother multigrid, (K,M) dictionary, balanced SpMV)
against the oracle: Newton counts per proximal step, primal field - or the same SNES failure (settings B diverges with DTOL on some
meshes in the oracle as well).   python tools/p2_odd_sizes.py"""
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from oracle import pg_oracle as O  # noqa: E402
from proximalgalerkin_amd import fem  # noqa: E402
from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem  # noqa: E402


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


CASES = [((2, 2), "B"), ((3, 5), "B"), ((27, 14), "B"), ((33, 31), "B"), ((65, 7), "B"), ((65, 7), "A"), ((34, 33), "A"), ((129, 3), "A")]
S = {"A": ("constant", 1e5, 1e-6), "B": ("double_exponential", 1e2, 1e-4)}
bad = 0
for (N, M), st in CASES:
    msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, M))
    pb, sol, solk, al = setup_problem(msh, 2)
    try:
        h = run_outer_loop(pb, sol, solk, al, 100, *S[st])["Newton steps"]
    except Exception as ex:  # noqa: BLE001
        h = "diverged: " + str(ex)[-40:]
    c, e = O.create_rectangle(N, M)
    pr = O.ObstacleLagrange(c, e, 2)
    try:
        xr, hr = O.solve_problem(pr, 100, *S[st])
        hr = hr["Newton steps"]
    except Exception as ex:  # noqa: BLE001
        xr, hr = None, "diverged: " + str(ex)[-40:]
    if isinstance(h, str) or isinstance(hr, str):
        ok = isinstance(h, str) and isinstance(hr, str)
        print(f"ex01 P2 {N}x{M} settings {st}: HIP {h} | oracle {hr} -> {'same outcome' if ok else 'DIFFERENT'}", flush=True)
    else:
        d = rel(sol.x.array[:pr.n], xr[:pr.n])
        ok = list(h) == list(hr) and d < 1e-9
        print(f"ex01 P2 {N}x{M} settings {st}: Newton {list(h)} | oracle {list(hr)}  rel L2(u) {d:.1e} -> {'ok' if ok else 'DIFFERENT'}", flush=True)
    bad += not ok
    pb.close()
sys.exit(1 if bad else 0)
