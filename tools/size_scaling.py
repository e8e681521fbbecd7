"""Single-GPU size scaling table of DESIGN.md section 3: python tools/size_scaling.py [sizes...]"""
import json
import os
os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")  # PGX_* switches reach the library through the loader's opt-in bridge
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = [int(a) for a in sys.argv[1:]] or [512, 1024, 2048, 3072, 4096]
for n in sizes:
    row = [f"{n}^2"]
    for st in ("A", "B"):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cells", str(n), "--settings", st, "--steps", "2", "--warmup", "1",
                            "--no-cpu-baseline"], capture_output=True, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if "Memory access fault" in r.stderr:
            print(n, st, "GPU FAULT", r.stderr[-300:])
            sys.exit(1)  # never run on after a faulting GPU step
        if r.returncode or not lines:
            row.append(f"settings {st}: no converged run ({r.stderr.strip().splitlines()[-1][-120:] if r.stderr.strip() else 'rc ' + str(r.returncode)})")
            continue
        d = json.loads(lines[-1])
        c = d["config"]
        row.append(f"settings {st}: {d['ms_per_step']:.0f} ms / {c['newton_iterations_per_step']:.0f} Newton / {c['proximal_iterations_per_step']:.0f} proximal "
                   f"({d['ms_per_step'] / max(c['newton_iterations_per_step'], 1):.1f} ms per Newton step), apply {d['roofline']['achieved']:.0f} GB/s, setup {d['setup_s']:.1f} s")
    print(" | ".join(row), flush=True)
