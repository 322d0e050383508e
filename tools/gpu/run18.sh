cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r18
mkdir -p $O
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/arena_trace -o arena -- python3 $GRAFT_REPO_ROOT/tools/arena_real_bench.py --plies 1 > $GRAFT_REPO_ROOT/$O/arena_trace.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/$O/arena_trace.log; exit 1; }
cd $GRAFT_REPO_ROOT
tail -2 $O/arena_trace.log
python tools/trace_timeline.py $O/arena_trace k_compact -5 > $O/arena_timeline.txt 2>&1
cat $O/arena_timeline.txt
python tools/summarize_prof.py $O/arena_trace > $O/arena_stats.txt 2>&1; head -30 $O/arena_stats.txt
rm -rf $O/arena_trace
