#!/bin/bash
# A/B of tuning keys on BASELINE config 3 on one GPU (2048^2 P2, settings A): bash tools/p2_env_sweep.sh "PGX_P2_PATCH_NU=1" ...
mkdir -p gpurun_out
export PGX_TUNING_FROM_ENV=1
for e in "$@"; do
  env $e python bench.py --degree 2 --settings A --no-cpu-baseline --solves-only --steps 1 --warmup 1 > gpurun_out/swp2.json 2>> gpurun_out/swp2.err
  python -c "
import json
d=json.loads(open('gpurun_out/swp2.json').read().strip().splitlines()[-1]); print('$e', round(d['value'],2), round(d['ms_per_step'],1), d['config']['newton_iterations_per_step'], d.get('last_newton_linear_iterations'))
"
done
