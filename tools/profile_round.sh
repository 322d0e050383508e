#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the rocprofv3 passes behind profiles/ -- kernel trace + stats of the default bench
# command (secondary legs off), then the PMC passes, one counter group at a time (never combined with other trace domains).
# Usage: tools/profile_round.sh [precision] [tag] [extra bench args]
set -e
PREC=${1:-bf16x3}; TAG=${2:-r6}
[ $# -ge 1 ] && shift
[ $# -ge 1 ] && shift
EXTRA="$*"
OUT=gpurun_out/prof_${TAG}_${PREC}
mkdir -p $OUT
export TMPDIR=/tmp
run() { # name, rocprof args..., then bench args
  local name=$1; shift
  timeout -k 10 300 rocprofv3 "$@" --output-format csv -d $OUT/$name -o run -- python3 bench.py --precision $PREC --no-cpu-baseline --no-compare $EXTRA $BENCH_ARGS > $OUT/$name.log 2>&1 || { echo "pass $name FAILED:"; tail -5 $OUT/$name.log; exit 1; }
  grep -q '^{' $OUT/$name.log || { echo "pass $name printed no bench line:"; tail -5 $OUT/$name.log; exit 1; }
  echo "$name done"
}
BENCH_ARGS="" run trace --kernel-trace --stats          # the default command: staggered slots, 2 warm-up + 20 timed steps (free-running driver, capped batches; EXTRA="--driver lockstep" for the other one)
# counter passes: one timed step (100 batches) after a cheap stagger (every launch is serialised under --pmc)
BENCH_ARGS="--steps 1 --warmup 0 --stagger-sims 8" run fetch --kernel-trace --pmc FETCH_SIZE
BENCH_ARGS="--steps 1 --warmup 0 --stagger-sims 8" run write --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
BENCH_ARGS="--steps 1 --warmup 0 --stagger-sims 8" run sq --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY
# keep only the CSVs (the merge back is capped at 64 MiB); the kernel trace of the long run is summarised on the box
LINE() { python3 -c "import json,sys; d=json.loads([l for l in open('$OUT/$1.log') if l.startswith('{')][0]); print($2)"; }
LEAVES=$(LINE trace "d['leaves_evaluated_rank0']/d['roofline']['launches']")
TIMED=$(LINE trace "d['roofline']['launches']")
python3 tools/summarize_prof.py trace $OUT/trace $OUT/kernel_trace_by_shape.csv "round ${TAG}, precision ${PREC}: rocprofv3 --kernel-trace --stats -- python3 bench.py --precision ${PREC} --no-cpu-baseline --no-compare ${EXTRA}" $LEAVES $TIMED
grep -h '^{' $OUT/trace.log > $OUT/bench_line_of_the_traced_run.json
python3 tools/summarize_prof.py pmc $OUT/pmc_by_shape.csv "round ${TAG}, precision ${PREC}: rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum | SQ group), bench.py --steps 1 --warmup 0 --stagger-sims 8; per-launch averages over the 100 launches of the ONE timed step (4096 games, batches of the cap); GEMM rows carry their layer (launch order inside the forward)" last=100 $OUT/fetch $OUT/write $OUT/sq
# per-launch HBM-side traffic of the dominant launch (roofline.traffic of bench.py) and of the table gather, and the tree side per simulation
PLEAVES=$(LINE fetch "d['leaves_evaluated_rank0']/d['roofline']['launches']")
PSIMS=$(LINE fetch "d['simulations']/d['roofline']['launches']")
python3 tools/summarize_prof.py traffic $OUT/pmc_by_shape.csv $OUT/conv3_traffic_${PREC}.json $PREC "[conv3]" 0 conv3 $PLEAVES || true
python3 tools/summarize_prof.py tree $OUT/pmc_by_shape.csv $OUT/tree_traffic.json $PSIMS || true
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
find $OUT -type f -name '*kernel_trace.csv' -size +20M -delete
find $OUT -type f ! -name '*.csv' ! -name '*.log' ! -name '*.json' -delete
ls -la $OUT
