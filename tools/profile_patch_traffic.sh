#!/bin/bash
# HBM traffic of the P2 patch sweep (k_patch_apply + k_patch_edges at 2048^2 P2): rocprofv3 --pmc FETCH_SIZE and
# WRITE_SIZE in SEPARATE passes over tools/p2_patch_bench.py, summarised per kernel by tools/pmc_summary.py.
#   bash tools/profile_patch_traffic.sh <outdir under gpurun_out>        (run from the repo root through gpurun)
# Algorithmic bytes (pgx_smoother_bench's count): per patch the float inverse in symmetric packing (116 float4
# = 464 B... see pgxk_patch_inverse_bytes), its dof table (4 NN B), the residual at its dofs once (16 B per dof), the vertex iterate
# read + written (32 B), the parked edge contributions (2 x 2 x 4 B per edge); per edge in k_patch_edges: the two parked pairs (16 B)
# and the edge iterate read + written (32 B).
set -o pipefail
OUT=gpurun_out/$1
mkdir -p $OUT
export TMPDIR=/tmp
export PGX_TUNING_FROM_ENV=1
N=${2:-2048}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/patch_$c -o p -- python3 tools/p2_patch_bench.py $N > $OUT/patch_$c.log 2>&1 || exit 1
done
NV=$(( (N + 1) * (N + 1) )); NE=$(( 3 * N * N + 2 * N ))
EDGES=$(( 48 * NE ))
python3 - <<PY > $OUT/patch_alg.txt
import re
t = open("$OUT/patch_FETCH_SIZE.log").read()
m = re.search(r"([0-9.]+) us\s+([0-9.]+) GB/s", t)
us, gbs = float(m.group(1)), float(m.group(2))
print(int(us * 1e-6 * gbs * 1e9))
PY
TOTAL=$(cat $OUT/patch_alg.txt)
APPLY=$(( TOTAL - EDGES ))
python3 tools/pmc_summary.py --kernel k_patch_apply --traffic --cells $N --algorithmic-bytes $APPLY --out $OUT/patch_apply_pmc_traffic.json $OUT/patch_FETCH_SIZE $OUT/patch_WRITE_SIZE > /dev/null || exit 1
python3 tools/pmc_summary.py --kernel k_patch_edges --traffic --cells $N --algorithmic-bytes $EDGES --out $OUT/patch_edges_pmc_traffic.json $OUT/patch_FETCH_SIZE $OUT/patch_WRITE_SIZE > /dev/null || exit 1
find $OUT -name '*.csv' -size +2M -delete
echo "profile_patch_traffic: done (sweep $TOTAL B = apply $APPLY + edges $EDGES)"
