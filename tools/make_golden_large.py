#!/usr/bin/env python3
"""Large-mesh goldens of example 01 (P1, settings B) from the CPU oracle, plus the CPU ladder.

Two products per mesh size N (run in the build container; minutes to hours of SuperLU):

* tests/golden/obstacle_p1_n{N}_settingsB_large.npz - final u (fp64), per-step Newton counts, the
  six observable columns.  At N >= 256 the HIP path's multigrid hierarchy has its fused smoother /
  residual-restriction levels active, so the GPU parity test on these fixtures compares the kernels
  that dominate the 2048^2 benchmark with the exact-Newton SuperLU oracle.
* profiles/r02_cpu_ladder.json - seconds per Newton step of the oracle (assembly + SuperLU) at
  each N, and the exponent p of a least-squares fit  t = c N^p ; bench.py's cpu_baseline leg
  extrapolates its live measurement to the benchmarked mesh with that exponent.

These fixtures are generated from the ORACLE (parity unpinned, oracle/pg_oracle.py header).
Usage: python tools/make_golden_large.py 256 512 [1024:2]   (N:k = only k Newton steps, ladder only)
"""
import json
import pathlib
import platform
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import pg_oracle as O  # noqa: E402

GOLD = ROOT / "tests" / "golden"
LADDER = ROOT / "profiles" / "r02_cpu_ladder.json"


def one(N, max_newton=None):
    coords, cells = O.create_rectangle(N, N)
    prob = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    log = O.NewtonLog()
    t0 = time.perf_counter()
    if max_newton is None:
        x, hist = O.solve_problem(prob, 100, "double_exponential", 1e2, 1e-4, log=log, verbose=True)
        newton = int(sum(hist["Newton steps"]))
        wall = time.perf_counter() - t0
        n = prob.n
        np.savez_compressed(GOLD / f"obstacle_p1_n{N}_settingsB_large.npz", N=N, u_final=x[:n],
                            psi_min=float(x[n:].min()),
                            **{("hist_" + k.replace(" ", "_")): np.asarray(v) for k, v in hist.items()})
    else:  # ladder point only: the first max_newton Newton steps of proximal step 1
        snes = O.SnesOptions(rtol=1e-30, max_it=max_newton)
        z = np.zeros(2 * prob.n)
        O.newton_solve(prob, z, z, 1.0, snes, log=log)
        newton = max_newton
        wall = time.perf_counter() - t0
    rec = {"N": N, "newton_steps": newton, "wall_s": wall, "s_per_newton_step": wall / newton,
           "t_factor": log.t_factor, "t_solve": log.t_solve, "t_jacobian": log.t_jacobian, "t_residual": log.t_residual,
           "complete_run": max_newton is None}
    print(json.dumps(rec), flush=True)
    return rec


def main():
    recs = []
    if LADDER.exists():
        recs = json.loads(LADDER.read_text())["points"]
    for a in sys.argv[1:]:
        N, _, k = a.partition(":")
        r = one(int(N), int(k) if k else None)
        recs = [p for p in recs if p["N"] != r["N"]] + [r]
        recs.sort(key=lambda p: p["N"])
        out = {"what": "CPU oracle (numpy assembly + SuperLU/COLAMD exact Newton, 1 thread), P1 settings B on [-1,1]^2",
               "host": platform.processor() or platform.machine(), "points": recs}
        if len(recs) >= 2:
            lx = np.log([p["N"] for p in recs])
            ly = np.log([p["s_per_newton_step"] for p in recs])
            p, c = np.polyfit(lx, ly, 1)
            out["fit"] = {"model": "s_per_newton_step = c * N^p", "p": float(p), "c": float(np.exp(c))}
        LADDER.write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
