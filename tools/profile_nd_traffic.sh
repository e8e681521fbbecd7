#!/bin/bash
# HBM traffic of the sparse LU on BASELINE configs 4 / 5: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes
# over one short run, summed per kernel and per factorisation / solve by tools/nd_traffic.py, + the kernel statistics of a third,
# counter-free pass.     bash tools/profile_nd_traffic.sh <outdir under gpurun_out> ex06|ex02
set -o pipefail
OUT=gpurun_out/$1
mkdir -p $OUT
export TMPDIR=/tmp PGX_TUNING_FROM_ENV=1 PGX_ND_DEPTHPROF=1
if [ "$2" = "ex02" ]; then PROG="tools/sg_scaling.py 70"; TAG=ex02_70; else PROG="tools/gc_scaling.py 1024 1"; TAG=ex06_1024; fi
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${TAG}_$c -o p -- python3 $PROG > $OUT/${TAG}_$c.log 2>&1 || exit 1
done
ND_TRAFFIC_PROGRAM="python3 $PROG" python3 tools/nd_traffic.py $OUT/${TAG}_FETCH_SIZE $OUT/${TAG}_WRITE_SIZE > $OUT/nd_traffic_$TAG.json || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o s -- python3 $PROG > $OUT/${TAG}_stats.log 2>&1 || exit 1
cp $(find $OUT/${TAG}_stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
find $OUT -name '*.csv' -size +2M -delete
echo "profile_nd_traffic $TAG: done"
