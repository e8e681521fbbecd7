#!/bin/bash
# A/B of solver options on the headline workload (run through gpurun from the repo root): bash tools/opts_sweep.sh "mg_nu=6" "mg_nu=9,mg_omega=0.8" ...
mkdir -p gpurun_out
for o in "$@"; do
  python bench.py --no-cpu-baseline --solves-only --steps 2 --warmup 1 --opts "$o" > gpurun_out/sw.json 2>> gpurun_out/sw.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1]); print('$o', round(d['value'],2), round(d['ms_per_step'],1), d['config'].get('newton_iterations_per_step'), d.get('last_newton_linear_iterations'))
"
done
