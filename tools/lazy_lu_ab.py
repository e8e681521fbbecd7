"""A/B of lazy refactorisation (PGX_LAZY_LU) on examples 06 and 02: python tools/lazy_lu_ab.py ex06 1024 | ex02 70"""
import os
os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")  # PGX_* switches reach the library through the loader's opt-in bridge
import subprocess
import sys

kind, n = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for label, env in (("fresh LU every step", {"PGX_LAZY_LU": "0"}), ("lazy, budget 10", {"PGX_LAZY_LU": "1"}), ("lazy, budget 6", {"PGX_LAZY_LU": "1", "PGX_LAZY_BUDGET": "6"})):
    e = dict(os.environ, PGX_LAZY_REPORT="1", **env)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", kind, "--cells", n, "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline"], env=e, capture_output=True, text=True)
    import json
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not lines:
        print(label, "FAILED", r.stderr[-400:])
        continue
    d = json.loads(lines[-1])
    rep = [ln for ln in r.stderr.splitlines() if ln.startswith("pgx: lazy")]
    print(f"{kind} {n} {label:22s} {d['ms_per_step']:9.1f} ms/solve  {d['value']:.3f} Newton it/s  newton {d['config']['newton_per_lvpp_step']}  {rep[-1] if rep else ''}", flush=True)
