"""How accurate is the FIRST sparse-LU solve of every Newton step, and how many refinement solves follow?  Runs example 06 with
`ksp_monitor` (the library prints the true relative residual after each solve) and summarises its own output.
    python tools/lu_refinement_stats.py ex06 1024
"""
import re
import subprocess
import sys

if len(sys.argv) > 3 and sys.argv[3] == "child":
    sys.path.insert(0, ".")
    what, n = sys.argv[1], int(sys.argv[2])
    if what != "ex06":
        raise SystemExit("ex06 only")
    from proximalgalerkin_amd import gradient_constraint as G

    G.PETSC_OPTIONS["ksp_monitor"] = None
    G.solve_problem(n, n)
    sys.exit(0)

out = subprocess.run([sys.executable, __file__, sys.argv[1], sys.argv[2], "child"], capture_output=True, text=True).stdout
first, nref = [], []
cur = None
for line in out.splitlines():
    m = re.match(r"\s+refinement (\d+)\s+true rel residual ([0-9.eE+-]+)", line)
    if m:
        k, v = int(m.group(1)), float(m.group(2))
        if k == 0:
            first.append(v)
            nref.append(0)
        else:
            nref[-1] = k
first.sort()
print(f"{len(first)} Newton linear systems; first-solve true relative residual: min {first[0]:.1e} median {first[len(first) // 2]:.1e} "
      f"max {first[-1]:.1e}")
for t in (1e-9, 1e-10, 1e-11, 1e-12):
    print(f"  first solve already <= {t:.0e}: {sum(v <= t for v in first)} of {len(first)}")
print("refinement solves per system:", {k: nref.count(k) for k in sorted(set(nref))})
