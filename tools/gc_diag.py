"""Example 06 LVPP run with the linear-solve monitor on (refinement / GMRES residuals per Newton step):
python tools/gc_diag.py N [first_monitored_step]"""
import sys

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from proximalgalerkin_amd import fem  # noqa: E402
from proximalgalerkin_amd.gradient_constraint import PETSC_OPTIONS, GradientConstraintProblem, f_default, phi_default  # noqa: E402

N = int(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
problem = GradientConstraintProblem(fem.create_unit_square(N, N), phi_default, f_default, petsc_options=dict(PETSC_OPTIONS))
for i in range(25):
    problem._opts.monitor = 2 if i + 1 >= first else 0
    problem.set_alpha(2.0**i)
    try:
        reason, n = problem.solve()
    except Exception as e:  # noqa: BLE001
        print("step", i + 1, "failed:", e, flush=True)
        break
    d = problem.l2_increment()
    print(f"step {i + 1}: alpha={2.0**i:g} reason={reason} newton={n} |du|={d:.3e}", flush=True)
    if d < 1e-8:
        break
    problem.advance_prev()
