cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r19
mkdir -p $O
timeout -k 10 300 python tools/net_by_batch.py > $O/by_batch.txt 2>&1 || { tail -20 $O/by_batch.txt; exit 1; }
tail -5 $O/by_batch.txt
timeout -k 10 500 python tools/pp_race_check.py > $O/race.txt 2>&1 || { tail -20 $O/race.txt; exit 1; }
tail -3 $O/race.txt
timeout -k 10 300 python tools/arena_real_bench.py --plies 6 --kernels > $O/arena6.txt 2>&1 || { tail -20 $O/arena6.txt; exit 1; }
tail -1 $O/arena6.txt
