"""PCIe-inclusive timing: the literal host-array statements of obstacle_pg.py:157-158,226 vs the device-resident loop."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proximalgalerkin_amd import fem
from proximalgalerkin_amd.obstacle import setup_problem, run_outer_loop
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
problem, sol, sol_k, alpha = setup_problem(msh, 1)
for resident in (True, False, True, False):
    t = time.perf_counter()
    h = run_outer_loop(problem, sol, sol_k, alpha, 500, "double_exponential", 1e2, 1e-4, device_resident=resident)
    print(f"device_resident={resident}: {1e3*(time.perf_counter()-t):.1f} ms, newton {sum(h['Newton steps'])}, outer {h['outer_iterations']}")
