#!/bin/bash
# Run ON THE GPU BOX through gpurun (GRAFT_REPO_ROOT is set there): rocprofv3 kernel trace of two arena plies (512 games x 800 sims, real networks) -> the
# timeline of one simulation step and the per-kernel stats.  A missing variable or a failed step ends the script.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/arena_bench
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/arena_trace -o arena -- python3 $R/tools/arena_real_bench.py --plies 2 > $R/$O/arena_trace.log 2>&1 || { tail -20 $R/$O/arena_trace.log; exit 1; }
cd $R
python tools/trace_timeline.py $O/arena_trace k_compact -5 > $O/arena_timeline.txt 2>&1
cp $(find $O/arena_trace -name "*kernel_stats.csv" | head -1) $O/arena_kernel_stats.csv
rm -rf $O/arena_trace
cat $O/arena_timeline.txt
# (the bench itself: tools/gpu/run_bench.sh)
