#!/bin/bash
# A/B of tuning keys on the headline workload (through gpurun, from the repo root): bash tools/env_sweep.sh "PGX_A=1 PGX_B=2" "PGX_A=0" ...
mkdir -p gpurun_out
export PGX_TUNING_FROM_ENV=1
for e in "$@"; do
  env $e python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/sw.json 2>> gpurun_out/sw.err
  python -c "
import json
d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1]); v=d['vcycle_parts']['from_level_us']; print('$e', round(d['value'],2), round(d['ms_per_step'],1), d.get('last_newton_linear_iterations'), {k.split()[0]: round(x,1) for k,x in v.items()})
"
done
