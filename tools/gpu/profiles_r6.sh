#!/bin/bash
# Run ON THE GPU BOX through gpurun: the rocprofv3 passes behind profiles/r6_* -- kernel trace + stats of the default bench command and the PMC
# passes (one counter group at a time), for the precision given (default bf16x3, the bench default), then the rest of the test suite from
# where a failure stopped it.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6prof
mkdir -p $O
for P in ${PRECS:-bf16x3}; do
  bash tools/profile_round.sh $P r6 > $O/profile_$P.log 2>&1 || { tail -20 $O/profile_$P.log; exit 1; }
  tail -3 $O/profile_$P.log
done
