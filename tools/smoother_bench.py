#!/usr/bin/env python3
"""Microbenchmark of the time-dominant kernel (pgx_smoother_bench): the finest level's smoother launch with an iterate and a coarse
correction - k_f_smooth<16,3,...> of the single-precision V-cycle (k_st_smoothR with PGX_MG_F32=0).  python tools/smoother_bench.py [cells]"""
import os
os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proximalgalerkin_amd import fem  # noqa: E402
from proximalgalerkin_amd.obstacle import setup_problem  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
problem, sol, sol_k, alpha = setup_problem(msh, 1)
problem.assemble_jacobian()
for rep in range(3):
    ms, by = problem.smoother_bench(reps=50)
    print(f"rep {rep}: {1e3 * ms:.1f} us  {by / ms / 1e6:.0f} GB/s ({by / ms / 1e6 / 80:.1f} % of 8 TB/s), {by:.0f} algorithmic bytes")
problem.close()
