#!/bin/bash
# Run ON THE GPU BOX through gpurun (GRAFT_REPO_ROOT is set there): a missing variable or a failed step ends the script.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r30
for i in 1 2 3 4 5 6; do timeout -k 10 300 python tools/pp_race_check.py 2>/dev/null | grep -v amdgpu | head -1 >> gpurun_out/r30/race.txt || exit 1; done
cat gpurun_out/r30/race.txt
timeout -k 10 400 python tools/conv3_tile_probe.py 300 2>/dev/null | grep -c '"bit_identical": false' > gpurun_out/r30/tile_mismatch_lines.txt; echo "tile probe lines with a mismatch: $(cat gpurun_out/r30/tile_mismatch_lines.txt)"
