"""A/B of the fused leaf kernel (PGX_ND_LEAF_FUSED=0/1) on the sparse-LU workloads: ms per factorisation and per solve.
python tools/leaf_ab.py [ENV_VAR]  (one GPU; default switch PGX_ND_LEAF_FUSED; each case runs in a child process so that the environment variable is read at create)"""
import os
os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")  # PGX_* switches reach the library through the loader's opt-in bridge
import subprocess
import sys

VAR = sys.argv[1] if len(sys.argv) > 1 else "PGX_ND_LEAF_FUSED"  # any 0/1 switch of the library
AB = ("0", "1")

CASES = {
    "ex06 1024^2": "import runpy, sys; sys.argv = ['tools/gc_scaling.py', '1024', '3']; runpy.run_path('tools/gc_scaling.py', run_name='__main__')",
    "ex01 P2 1024^2 (settings A, 5 proximal steps)": """
import sys, time
sys.path.insert(0, ".")
from proximalgalerkin_amd import fem
from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem
msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (1024, 1024))
problem, sol, sol_k, alpha = setup_problem(msh, 2, petsc_options={"snes_linesearch_type": "none", "snes_rtol": 1e-6, "snes_max_it": 100})
problem.profile(enable=True, reset=True)
t = time.perf_counter()
h = run_outer_loop(problem, sol, sol_k, alpha, 5, "constant", 1e5, 1e-6)
print("  P2 1024: Newton", sum(h["Newton steps"]), f"in {time.perf_counter() - t:.2f} s; phases ms:", {k: round(v, 1) for k, v in problem.profile().items()})
""",
}
for name, code in CASES.items():
    for fused in AB:
        env = dict(os.environ, **{VAR: fused})
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        lines = [l for l in out.stdout.splitlines() if "phases" in l or "total" in l]
        print(f"{name}  {VAR}={fused}")
        for l in lines:
            print("   ", l.strip())
        if out.returncode:
            print(out.stderr[-2000:])
            sys.exit(1)
