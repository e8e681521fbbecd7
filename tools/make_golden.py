#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (oracle/pg_oracle.py).

The reference holds no golden vectors for this path and cannot run here, so these
fixtures pin the ORACLE against regressions and give the HIP path fixed targets; they are not
reference outputs ("parity unpinned").  Regenerate with:  python tools/make_golden.py
"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import pg_oracle as O  # noqa: E402

OUT = ROOT / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)


def run(N, scheme, alpha_max, tol, tag):
    coords, cells = O.create_rectangle(N, N)
    prob = O.ObstacleP1(coords, cells, O.boundary_vertices_rectangle(N, N))
    its = []
    x, hist = O.solve_problem(prob, 100, scheme, alpha_max, tol, iterates=its)
    z = np.zeros(2 * prob.n)
    x3 = its[min(2, len(its) - 1)]
    xk3 = its[min(1, len(its) - 1)]
    np.savez_compressed(
        OUT / f"obstacle_p1_n{N}_{tag}.npz",
        N=N, scheme=scheme, alpha_max=alpha_max, tol=tol,
        x_final=x,
        **{("hist_" + k.replace(" ", "_")): np.asarray(v) for k, v in hist.items()},
        F_zero=prob.residual(z, z, 1.0),
        x_iter=x3, xk_iter=xk3, F_iter=prob.residual(x3, xk3, 2.5),
        D_iter=prob.jacobian_blocks(x3),
        obs_iter=prob.observables(x3, xk3, 2.5),
        b_phi=prob.b_phi,
        cells_hash=np.frombuffer(np.ascontiguousarray(cells).tobytes(), dtype=np.uint8).astype(np.uint64).sum(),
    )
    print(tag, N, hist["Newton steps"])


def run_p2(N, scheme, alpha_max, tol, tag):
    coords, cells = O.create_rectangle(N, N)
    prob = O.ObstacleLagrange(coords, cells, 2)
    its = []
    x, hist = O.solve_problem(prob, 100, scheme, alpha_max, tol, iterates=its)
    np.savez_compressed(OUT / f"obstacle_p2_n{N}_{tag}.npz", N=N, scheme=scheme, alpha_max=alpha_max, tol=tol, x_final=x,
                        **{("hist_" + k.replace(" ", "_")): np.asarray(v) for k, v in hist.items()},
                        x_iter=its[2], xk_iter=its[1], F_iter=prob.residual(its[2], its[1], 2.5),
                        D_iter=prob.jacobian_blocks(its[2]), obs_iter=prob.observables(its[2], its[1], 2.5))
    print("P2", tag, N, hist["Newton steps"])


def run_gc(N, tag):
    """example 06 (oracle/gc_oracle.py), reference defaults: doubling alpha, stopping_tol 1e-8, max 25"""
    from oracle import gc_oracle as G

    coords, cells = O.create_rectangle(N, N, (0.0, 0.0), (1.0, 1.0))
    prob = G.GradientConstraintP2(coords, cells)
    its = []
    x, newton, diffs = G.solve_problem(prob, iterates=its)
    x3, xk3 = its[3], its[2]
    J = prob.jacobian(x3, 5.0).tocsr()
    v = np.sin(np.arange(prob.ntot) * 0.37)
    np.savez_compressed(OUT / f"gradient_constraint_p2_n{N}_{tag}.npz", N=N, x_final=x, newton=newton, L2_diff=diffs,
                        x_iter=x3, xk_iter=xk3, F_iter=prob.residual(x3, xk3, 5.0), Jv_iter=J @ v, v=v,  # alpha != the step's own (residual would be ~0)
                        l2_iter=prob.l2_increment(x3, xk3))
    print("ex06", N, newton)


if __name__ == "__main__":
    run_gc(12, "defaults")
    run_p2(16, "double_exponential", 1e2, 1e-4, "settingsB")
    run(16, "double_exponential", 1e2, 1e-4, "settingsB")
    run(16, "constant", 1e5, 1e-6, "settingsA")
    run(32, "double_exponential", 1e2, 1e-4, "settingsB")
