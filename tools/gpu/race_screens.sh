#!/bin/bash
# Run ON THE GPU BOX through gpurun (GRAFT_REPO_ROOT is set there): a missing variable or a failed step ends the script.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r30
rm -f gpurun_out/r30/race.txt
for i in 1 2 3; do timeout -k 10 400 python tools/pp_race_check.py 2>/dev/null | grep -v amdgpu | tail -3 >> gpurun_out/r30/race.txt || { echo "pp_race_check run $i FAILED"; cat gpurun_out/r30/race.txt; exit 1; }; done
cat gpurun_out/r30/race.txt
n=$(timeout -k 10 400 python tools/conv3_tile_probe.py 300 2>/dev/null | grep -c '"bit_identical": false' || true)
echo "conv3 tile probe lines with a mismatch: $n"
m=$(timeout -k 10 300 python tools/low_loop_probe.py 100 2>/dev/null | grep -c '"same_bits": false' || true)
echo "128-row tile loop probe lines with a mismatch: $m"
[ "$n" = 0 ] && [ "$m" = 0 ]
