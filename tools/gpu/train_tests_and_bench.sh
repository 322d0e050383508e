#!/bin/bash
# Run ON THE GPU BOX through gpurun (GRAFT_REPO_ROOT is set there): a missing variable or a failed step ends the script.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r22
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_train.py -x -q > $O/pytest_train.txt 2>&1 || { tail -40 $O/pytest_train.txt; exit 1; }
tail -2 $O/pytest_train.txt
for b in 32 256 1024; do for p in f16x2 f32; do python tools/train_bench.py --batch $b --precision $p --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b $p', round(d['ms_per_step'],3), round(d['tflops_fp32'],1))"; done; done
