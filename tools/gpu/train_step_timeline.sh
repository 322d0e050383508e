#!/bin/bash
# Run ON THE GPU BOX through gpurun (GRAFT_REPO_ROOT is set there): a missing variable or a failed step ends the script.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r23
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tp32 -o run -- python3 $R/tools/train_bench.py --batch 32 --precision f16x2 --steps 100 > $R/$O/train_b32.log 2>&1 || { tail -5 $R/$O/train_b32.log; exit 1; }
cd $R
python tools/trace_timeline.py $O/tp32 k_t_conv1_fwd -3 > $O/train_b32_timeline.txt
cat $O/train_b32_timeline.txt
rm -rf $O/tp32
