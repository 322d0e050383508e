#!/bin/bash
# Run ON THE GPU BOX through gpurun: the bench contract test, the race screen of the ping-pong loops, then the driver's own bench command
# (stdout line + bench_detail.json kept under gpurun_out/r6a).
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6a
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bench_single_rank_contract" > $O/pytest_contract.txt 2>&1 || { tail -40 $O/pytest_contract.txt; exit 1; }
tail -3 $O/pytest_contract.txt
timeout -k 10 300 python tools/pp_race_check.py > $O/race.txt 2>&1 || { tail -20 $O/race.txt; exit 1; }
tail -3 $O/race.txt
SECONDS=0
timeout -k 10 560 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
echo "bench wall ${SECONDS}s; line bytes: $(wc -c < $O/bench.out)"
cp bench_detail.json $O/bench_detail.json
cat $O/bench.out
