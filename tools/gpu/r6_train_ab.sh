#!/bin/bash
# Run ON THE GPU BOX through gpurun: the training step at the reference's batch (32) and at 256, f16x2 and f32, three repetitions each
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6train
mkdir -p $O
for rep in 1 2 3; do for b in 32 256; do for p in f16x2 f32; do
  timeout -k 10 120 python tools/train_bench.py --batch $b --precision $p --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b $p', round(d['ms_per_step'],4))"
done; done; done | tee $O/train_steps.txt
