"""SpMV microbenchmark (GPU): python tools/spmv_bench.py N [kind]  -> ms, GB/s of the operator-apply kernel
(kind: 1 matrix-free stencil k_st_spmv_r (default), 0 block-CSR stream k_bspmv_stream, 2 generic stencil; PGX_SPMV_* env as set)."""
import os
os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")  # PGX_* switches reach the library through the loader's opt-in bridge
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proximalgalerkin_amd import fem
from proximalgalerkin_amd.obstacle import setup_problem
N = int(sys.argv[1])
msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
problem, sol, sol_k, alpha = setup_problem(msh)
rng = np.random.default_rng(0)
x = rng.standard_normal(2 * msh.num_vertices) * 0.1
problem.assemble_jacobian(x)
kind = problem.spmv_select(int(sys.argv[2]) if len(sys.argv) > 2 else -1)
best = None
for rep in range(3):
    ms, by = problem.spmv_bench(reps=20)
    best = ms if best is None else min(best, ms)
    print(f"  rep {rep}: {ms * 1e3:.1f} us")
print(f"N={N} kind={kind} bytes={by:.0f} stream={os.environ.get('PGX_SPMV_STREAM','1')} remap={os.environ.get('PGX_XCD_REMAP','1')}: {best*1e3:.1f} us  {by/best/1e6:.0f} GB/s  ({by/best/1e6/8000*100:.1f}% of 8 TB/s)")
