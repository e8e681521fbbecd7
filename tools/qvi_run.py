"""Example 05 at the reference's size (M = 150) or any M: python tools/qvi_run.py [M]"""
import sys
import time

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from proximalgalerkin_amd.thermoforming import solve_problem  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 150
t = time.perf_counter()
its, diffs = solve_problem(M, verbose=False)
dt = time.perf_counter() - t
print(f"M={M}: {len(its)} LVPP iterations, {sum(its)} Newton steps {its}, final increment {diffs[-1]:.2e}, {dt:.2f} s incl. setup")
