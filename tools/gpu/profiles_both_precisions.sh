#!/bin/bash
# Run ON THE GPU BOX through gpurun (GRAFT_REPO_ROOT is set there): a missing variable or a failed step ends the script.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r25
mkdir -p $O
bash tools/profile_round.sh f16x2 r5 > $O/profile_f16x2.log 2>&1 || { tail -20 $O/profile_f16x2.log; exit 1; }
tail -3 $O/profile_f16x2.log
bash tools/profile_round.sh f32 r5 > $O/profile_f32.log 2>&1 || { tail -20 $O/profile_f32.log; exit 1; }
tail -3 $O/profile_f32.log
