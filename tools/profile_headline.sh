#!/bin/bash
# Profiles of the headline workload (2048^2 P1, settings B) on the GPU box; run from the repo root through gpurun.
#   bash tools/profile_headline.sh <outdir under gpurun_out> [stats|spmv|stspmv|p2stspmv|fsmooth|smoother ...]
# stats    : rocprofv3 --kernel-trace --stats of `bench.py --steps 3`
# spmv     : FETCH_SIZE and WRITE_SIZE of k_bspmv_stream in SEPARATE passes (TCC slots) -> profiles/ via tools/pmc_summary.py
# smoother : two SQ passes over one solve (wait/issue counters; LDS conflict counters)
set -o pipefail
OUT=gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
export PGX_TUNING_FROM_ENV=1
BENCH="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --solves-only"
for what in "$@"; do
  case $what in
    stats)
      rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1 || exit 1
      cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
      rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1 || exit 1
      python3 tools/trace_by_grid.py $OUT/trace > $OUT/trace_by_level.txt; rm -rf $OUT/trace ;;
    spmv)
      rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python3 tools/spmv_bench.py 2048 0 > $OUT/pmc_fetch.log 2>&1 || exit 1
      rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python3 tools/spmv_bench.py 2048 0 > $OUT/pmc_write.log 2>&1 || exit 1
      python3 tools/pmc_summary.py --kernel k_bspmv_stream --traffic --cells 2048 --algorithmic-bytes 973570116 \
        --out $OUT/spmv_pmc_traffic.json $OUT/pmc_fetch $OUT/pmc_write > /dev/null || exit 1 ;;
    stspmv)  # the matrix-free operator apply in the solver's sequence (x = the float2 z_j of FGMRES): 57 B x 2049^2 vertices
      rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_sfetch -o f -- python3 tools/spmv_bench.py 2048 1 > $OUT/pmc_sfetch.log 2>&1 || exit 1
      rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_swrite -o w -- python3 tools/spmv_bench.py 2048 1 > $OUT/pmc_swrite.log 2>&1 || exit 1
      python3 tools/pmc_summary.py --kernel "k_st_spmv_r<true>" --traffic --cells 2048 --algorithmic-bytes 239308857 \
        --out $OUT/stspmv_pmc_traffic.json $OUT/pmc_sfetch $OUT/pmc_swrite > /dev/null || exit 1 ;;
    p2stspmv)  # the structured P2 operator apply (pgx_p2st.hip): 496 B x 2045^2 interior groups at 2048^2
      rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_pfetch -o f -- python3 tools/p2_spmv_bench.py 2048 > $OUT/pmc_pfetch.log 2>&1 || exit 1
      rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_pwrite -o w -- python3 tools/p2_spmv_bench.py 2048 > $OUT/pmc_pwrite.log 2>&1 || exit 1
      python3 tools/pmc_summary.py --kernel "k_p2st_apply_lds<double" --traffic --cells 2048 --algorithmic-bytes 2074284400 \
        --out $OUT/p2stspmv_pmc_traffic.json $OUT/pmc_pfetch $OUT/pmc_pwrite > /dev/null || exit 1 ;;
    fsmooth)  # the time-dominant kernel: the finest level's single-precision smoother launch, 40 B x 2049^2 vertices (+ coarse correction)
      rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_ffetch -o f -- python3 tools/smoother_bench.py 2048 > $OUT/pmc_ffetch.log 2>&1 || exit 1
      rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fwrite -o w -- python3 tools/smoother_bench.py 2048 > $OUT/pmc_fwrite.log 2>&1 || exit 1
      python3 tools/pmc_summary.py --kernel "k_f_smooth<16, 3, false, 0, 1, 0>" --traffic --cells 2048 --algorithmic-bytes 176341040 \
        --out $OUT/fsmooth_pmc_traffic.json $OUT/pmc_ffetch $OUT/pmc_fwrite > /dev/null || exit 1 ;;
    smoother)
      rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE \
        --kernel-trace --output-format csv -d $OUT/pmc_sqA -o a -- $BENCH > $OUT/pmc_sqA.log 2>&1 || exit 1
      rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM \
        --kernel-trace --output-format csv -d $OUT/pmc_sqB -o b -- $BENCH > $OUT/pmc_sqB.log 2>&1 || exit 1
      for k in k_f_smooth k_f_resid_restrict k_st_spmv_r k_mg_tail2 k_multiaxpy_norm; do
        python3 tools/pmc_summary.py --kernel $k --out $OUT/pmc_sq_$k.json $OUT/pmc_sqA $OUT/pmc_sqB > /dev/null || true
      done ;;
  esac
done
# keep the merged output small: raw counter CSVs stay on the box unless asked for
find $OUT -name '*.csv' -size +8M -delete
echo "profile_headline: done $*"
