#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes, tools/pmc_summary.py) of the three operator-apply kernels
# on the library in the tree; run from the repo root through gpurun:   bash tools/profile_traffic.sh <outdir under gpurun_out>
set -o pipefail
OUT=gpurun_out/$1
mkdir -p $OUT
export TMPDIR=/tmp
export PGX_TUNING_FROM_ENV=1
run() {  # name, counter, program args...
  local d=$OUT/$1_$2; shift; local c=$1; shift
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- "$@" > $d.log 2>&1 || exit 1
}
run st FETCH_SIZE python3 tools/spmv_bench.py 2048 1
run st WRITE_SIZE python3 tools/spmv_bench.py 2048 1
python3 tools/pmc_summary.py --kernel "k_st_spmv_r<true>" --traffic --cells 2048 --algorithmic-bytes 239308857 --out $OUT/stspmv_pmc_traffic.json $OUT/st_FETCH_SIZE $OUT/st_WRITE_SIZE > /dev/null || exit 1
run csr FETCH_SIZE python3 tools/spmv_bench.py 2048 0
run csr WRITE_SIZE python3 tools/spmv_bench.py 2048 0
python3 tools/pmc_summary.py --kernel k_bspmv_stream --traffic --cells 2048 --algorithmic-bytes 973570116 --out $OUT/spmv_pmc_traffic.json $OUT/csr_FETCH_SIZE $OUT/csr_WRITE_SIZE > /dev/null || exit 1
run p2 FETCH_SIZE python3 tools/p2_spmv_bench.py 2048
run p2 WRITE_SIZE python3 tools/p2_spmv_bench.py 2048
python3 tools/pmc_summary.py --kernel k_bspmv_bal --traffic --cells 2048 --algorithmic-bytes 3112894517 --out $OUT/p2spmv_pmc_traffic.json $OUT/p2_FETCH_SIZE $OUT/p2_WRITE_SIZE > /dev/null || exit 1
find $OUT -name '*.csv' -size +2M -delete
echo "profile_traffic: done"
