cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r17
mkdir -p $O
timeout -k 10 300 python tools/net_by_batch.py > $O/by_batch.txt 2>&1 || { tail -20 $O/by_batch.txt; exit 1; }
tail -5 $O/by_batch.txt
