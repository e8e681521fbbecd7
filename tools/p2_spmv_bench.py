#!/usr/bin/env python3
"""P2 operator apply (k_bspmv_bal) and patch sweep timings at N^2 (default 2048): python tools/p2_spmv_bench.py [N]"""
import os
import sys

os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")  # PGX_* switches reach the library through the loader's opt-in bridge
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proximalgalerkin_amd import fem  # noqa: E402
from proximalgalerkin_amd.obstacle import setup_problem  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
problem, sol, sol_k, alpha = setup_problem(msh, 2)
problem.assemble_jacobian()
ms, by = problem.spmv_bench(reps=20)
msc, _ = problem.spmv_bench_cold(reps=20)
print(f"P2 {N}^2 operator apply: {ms:.3f} ms ({by / ms / 1e6:.0f} GB/s = {by / ms / 1e6 / 8000:.3f} of peak); cold {msc:.3f} ms ({by / msc / 1e6 / 8000:.3f})")
problem.close()
