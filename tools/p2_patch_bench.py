"""P2 patch-sweep microbenchmark (GPU): python tools/p2_patch_bench.py N  -> ms and GB/s of k_patch_apply + k_patch_edges on the
N x N P2 Jacobian (pgx_smoother_bench; PGX_LIB selects another build of the library for A/B runs)."""
import os
os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proximalgalerkin_amd import fem
from proximalgalerkin_amd.obstacle import setup_problem
N = int(sys.argv[1])
msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
problem, sol, sol_k, alpha = setup_problem(msh, 2)
rng = np.random.default_rng(0)
x = rng.standard_normal(sol.function_space.num_dofs) * 0.1
problem.assemble_jacobian(x)
best = None
for rep in range(3):
    ms, by = problem.smoother_bench(reps=10)
    best = ms if best is None else min(best, ms)
print(f"N={N} P2 patch sweep lib={os.environ.get('PGX_LIB', 'default')}: {best * 1e3:.1f} us  {by / best / 1e6:.0f} GB/s ({by / best / 1e6 / 8000 * 100:.1f} % of 8 TB/s)")
