"""Sweep of PGX_K6_MAX (largest level that runs six smoother sweeps per launch) on the headline workload:
python tools/k6_sweep.py [cells]"""
import json
import os
os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")  # PGX_* switches reach the library through the loader's opt-in bridge
import subprocess
import sys

n = sys.argv[1] if len(sys.argv) > 1 else "2048"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for k6 in (0, 5000, 20000, 70000, 300000, 1100000, 5000000):
    e = dict(os.environ, PGX_K6_MAX=str(k6))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cells", n, "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       env=e, capture_output=True, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not lines or r.returncode != 0 or "Memory access fault" in r.stderr:
        print(k6, "FAILED", r.stderr[-300:])
        sys.exit(1)  # never run on after a failed GPU step
    d = json.loads(lines[-1])
    print(f"cells {n} PGX_K6_MAX={k6:8d}: {d['ms_per_step']:8.2f} ms/solve  {d['value']:.2f} Newton it/s  newton {d['config']['newton_iterations_per_step']}  "
          f"last lin its {d['last_newton_linear_iterations']}", flush=True)
