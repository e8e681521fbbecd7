"""Diagnostic (GPU): example 06 with the sparse LU distributed over R ranks (in-process thread transport) vs one handle.
python tools/gc_dist_diag.py N [R]"""
import sys
import threading

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from proximalgalerkin_amd import comm as pcomm  # noqa: E402
from proximalgalerkin_amd.gradient_constraint import solve_problem  # noqa: E402

N = int(sys.argv[1])
R = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 2
opts = {"snes_atol": 1e-9, "snes_rtol": 1e-9, "snes_stol": 1e-9, "snes_max_it": 20, "snes_linesearch_type": "none"}
MON = "--monitor" in sys.argv
if MON:
    import proximalgalerkin_amd.gradient_constraint as G

    G.PETSC_OPTIONS = dict(G.PETSC_OPTIONS, ksp_monitor=True, snes_error_if_not_converged=False)
    print("=== single handle", flush=True)
its1, d1 = solve_problem(N, N, verbose=MON, max_iterations=6 if MON else 25)
print("single:", list(its1), flush=True)
if MON:
    print("=== distributed", flush=True)
comms = pcomm.local_group(R)
out = [None] * R


def work(r):
    try:
        out[r] = solve_problem(N, N, verbose=(r == 0), comm=comms[r], max_iterations=6 if MON else 25)
    except BaseException as e:  # noqa: BLE001
        out[r] = e


th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
[t.start() for t in th]
[t.join(600) for t in th]
for r in range(R):
    print("rank", r, out[r] if isinstance(out[r], BaseException) else list(out[r][0]), flush=True)
