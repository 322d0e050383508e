#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the rocprofv3 passes behind profiles/ -- kernel trace + stats, then the PMC passes
# one counter group at a time (never combined with other trace domains).  Usage: tools/profile_round.sh <precision> <tag>
set -e
PREC=${1:-f16x2}; TAG=${2:-r1}
OUT=gpurun_out/prof_${TAG}_${PREC}
mkdir -p $OUT
export TMPDIR=/tmp
run() { # name, rocprof args..., then bench args
  local name=$1; shift
  timeout -k 10 400 rocprofv3 "$@" --output-format csv -d $OUT/$name -o run -- python3 bench.py --precision $PREC --no-cpu-baseline --no-dedup-compare $BENCH_ARGS > $OUT/$name.log 2>&1
  echo "$name done"
}
BENCH_ARGS="" run trace --kernel-trace --stats          # the default command: 64 move rounds = whole games
BENCH_ARGS="--steps 1 --warmup 0" run fetch --kernel-trace --pmc FETCH_SIZE
BENCH_ARGS="--steps 1 --warmup 0" run write --kernel-trace --pmc WRITE_SIZE
BENCH_ARGS="--steps 1 --warmup 0" run sq --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY
# keep only the CSVs (the merge back is capped at 64 MiB)
find $OUT -type f ! -name '*.csv' ! -name '*.log' -delete
ls -la $OUT/*
