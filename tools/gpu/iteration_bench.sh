#!/bin/bash
# Run ON THE GPU BOX through gpurun (GRAFT_REPO_ROOT is set there): a missing variable or a failed step ends the script.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/iter
R="$GRAFT_REPO_ROOT"
timeout -k 10 500 python tools/iteration_bench.py 2>/dev/null | grep "^{" > gpurun_out/iter/iteration_f16x2.json; cat gpurun_out/iter/iteration_f16x2.json
timeout -k 10 500 python tools/iteration_bench.py --batched-eval 2>/dev/null | grep "^{" > gpurun_out/iter/iteration_f16x2_batched_eval.json; cat gpurun_out/iter/iteration_f16x2_batched_eval.json
