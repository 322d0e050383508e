#!/bin/bash
# Run ON THE GPU BOX through gpurun (GRAFT_REPO_ROOT is set there): a missing variable or a failed step ends the script.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r12
timeout -k 10 500 python tools/conv3_tile_probe.py 200 > gpurun_out/r12/tile_probe.txt 2>&1 && rc=0 || rc=$?
echo "rc=$rc" >> gpurun_out/r12/tile_probe.txt
cat gpurun_out/r12/tile_probe.txt
