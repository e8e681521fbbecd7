#!/usr/bin/env python3
"""Device time of the V-cycle part that starts on each multigrid level (pgx_vcycle_bench; launches back to back), 2048^2 P1 by
default:  python tools/vcycle_bench.py [cells]   - level 0 is the whole preconditioner application, the last line the fused tail."""
import os
os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")  # PGX_* switches reach the library through the loader's opt-in bridge
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proximalgalerkin_amd import fem  # noqa: E402
from proximalgalerkin_amd.obstacle import setup_problem  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
problem, sol, sol_k, alpha = setup_problem(msh, 1)
problem.assemble_jacobian()
lv = 0
prev = None
while True:
    try:
        ms, n = problem.vcycle_bench(lv)
    except Exception:
        break
    print(f"from level {lv} ({n} vertices): {1e3 * ms:8.1f} us" + (f"   (this level alone: {1e3 * (prev - ms):7.1f} us)" if prev else ""))
    prev = ms
    lv += 1
ms, n = problem.vcycle_bench(-1)
print(f"fused tail launch ({n} vertices on its first level): {1e3 * ms:8.1f} us")
problem.close()
