#!/usr/bin/env python3
"""Where a wave of the level-0 smoother launch spends its time: a build of pgx_kernels.hip with cycle stamps in the interior-tile path of
k_st_smoothR<16,3,POST> (tools/make_stamp_build.py; library given as PGX_LIB) - per phase the mean over all waves of the
last POST launch of a V-cycle at 2048^2.   PGX_LIB=.../libpgx_stamp.so python tools/smoother_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proximalgalerkin_amd import _lib, fem  # noqa: E402
from proximalgalerkin_amd.obstacle import setup_problem  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
problem, sol, sol_k, alpha = setup_problem(msh)
x = np.random.default_rng(0).standard_normal(2 * msh.num_vertices) * 0.1
problem.assemble_jacobian(x)
lib = C.CDLL(os.environ["PGX_LIB"])
ms, _ = problem.vcycle_bench(0, reps=2)
buf = np.zeros((8192, 8, 12), dtype=np.uint64)
rc = lib.pgx_debug_read_stamps(buf.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
s = buf.astype(np.float64)
live = s[:, :, 10] > 0
nb = int(live.any(axis=1).sum())
t0 = s[live][:, 0].min()
t1 = s[live][:, 10].max()
print(f"V-cycle {ms * 1e3:.1f} us; {nb} interior tiles stamped; launch span {t1 - t0:.0f} ticks")
names = ["issue coefficient loads", "x / coarse loads -> LDS image", "barrier 0", "wait for coefficients", "sweep 1 compute", "barrier 1",
         "sweep 2 compute", "barrier 2", "sweep 3 compute + store"]
pairs = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9)]
tot = (s[:, :, 10] - s[:, :, 0])[live]
print(f"per wave, ticks (mean / median / p90), share of a tile's {tot.mean():.0f} ticks:")
for nm, (a, b) in zip(names, pairs):
    d = (s[:, :, b] - s[:, :, a])[live]
    print(f"  {nm:34s} {d.mean():8.0f} {np.median(d):8.0f} {np.percentile(d, 90):8.0f}   {100 * d.mean() / tot.mean():5.1f} %")
# per wave index: rows owned differ (waves 1..4 own 3 active rows in sweep 1, the others 2)
for w in range(8):
    lw = live[:, w]
    print(f"  wave {w}: sweep computes {[(s[:, w, b] - s[:, w, a])[lw].mean().round() for a, b in ((4, 5), (6, 7), (8, 9))]} barriers "
          f"{[(s[:, w, b] - s[:, w, a])[lw].mean().round() for a, b in ((2, 3), (5, 6), (7, 8))]}")
# slots: tiles per CU slot in sequence
start = s[:, 0, 0][live.any(axis=1)] - t0
end = s[:, 0, 10][live.any(axis=1)] - t0
print(f"tile start ticks: min {start.min():.0f} median {np.median(start):.0f} max {start.max():.0f}; tile duration mean {(end - start).mean():.0f}")
problem.close()
