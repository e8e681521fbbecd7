"""One full LVPP solve of example 06 (defaults of the reference: doubling alpha, tol 1e-8, max 25) on an N x N mesh:
python tools/gc_scaling.py N [max_iterations]"""
import sys
import time

import numpy as np

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from proximalgalerkin_amd import fem  # noqa: E402
from proximalgalerkin_amd.gradient_constraint import GradientConstraintProblem, f_default, phi_default  # noqa: E402

N = int(sys.argv[1])
maxit = int(sys.argv[2]) if len(sys.argv) > 2 else 25
t = time.perf_counter()
problem = GradientConstraintProblem(fem.create_unit_square(N, N), phi_default, f_default)
print(f"N={N} dofs u={problem.n2} total={problem.ndofs} setup {time.perf_counter() - t:.2f}s", flush=True)
problem.profile(True)
t = time.perf_counter()
its = []
for i in range(maxit):
    problem.set_alpha(2.0**i)
    t1 = time.perf_counter()
    reason, n = problem.solve()
    d = problem.l2_increment()
    its.append(n)
    print(f"  step {i + 1}: alpha={2.0**i:g} reason={reason} newton={n} |du|={d:.3e} ({(time.perf_counter() - t1) * 1e3:.0f} ms)", flush=True)
    if d < 1e-8:
        break
    problem.advance_prev()
dt = time.perf_counter() - t
print(f"  total {dt:.2f}s, Newton {its} sum {sum(its)} -> {sum(its) / dt:.2f} Newton it/s", flush=True)
print(f"  sparse LU storage {problem.lu_stats()['arena_doubles'] * 8 / 1e9:.1f} GB", flush=True)
print("  phases ms:", {k: round(v, 1) for k, v in problem.profile(False).items()}, flush=True)
