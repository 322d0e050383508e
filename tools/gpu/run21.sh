cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r21
mkdir -p $O
timeout -k 10 300 python tools/arena_real_bench.py --plies 6 --kernels > $O/arena6.txt 2>&1 || { tail -20 $O/arena6.txt; exit 1; }
tail -1 $O/arena6.txt
