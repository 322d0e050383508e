#!/bin/bash
# Run ON THE GPU BOX through gpurun (GRAFT_REPO_ROOT is set there): a missing variable or a failed step ends the script.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r6tests
timeout -k 10 1150 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r6tests/pytest_gpu.txt 2>&1 && rc=0 || rc=$?
echo "rc=$rc" >> gpurun_out/r6tests/pytest_gpu.txt
tail -30 gpurun_out/r6tests/pytest_gpu.txt
