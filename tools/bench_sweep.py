#!/usr/bin/env python3
"""Run bench.py over a list of variants and print one summary line each (same box, back to back: timings differ 5-10 % between
boxes of the pool, so every A/B runs in ONE gpurun call).
    python tools/bench_sweep.py "label:ENV=VAL,ENV2=VAL2:--cells 1024 --opts mg_nu=9" ...
Empty env / args parts are allowed ("base::")."""
import json
import os
os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")  # PGX_* switches reach the library through the loader's opt-in bridge
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in sys.argv[1:]:
    label, env_s, args_s = (spec.split(":", 2) + ["", ""])[:3]
    env = dict(os.environ)
    for kv in filter(None, env_s.split(",")):
        k, v = kv.split("=", 1)
        env[k] = v
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "5", "--warmup", "2"] + args_s.split()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode or not lines:
        print(f"{label:28s} FAILED rc={r.returncode} {r.stderr[-300:]!r}", flush=True)
        continue
    d = json.loads(lines[-1])
    c = d["config"]
    print(f"{label:28s} {d['ms_per_step']:9.1f} ms/step  {d['value']:8.2f} Newton it/s  newton/step {c.get('newton_iterations_per_step')}  "
          f"lin(last) {d.get('last_newton_linear_iterations')}  spmv {d['roofline']['achieved']:.0f} {d['roofline']['unit']}", flush=True)
