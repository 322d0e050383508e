cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r16
mkdir -p $O
timeout -k 10 600 python tools/cap_probe.py 10 3640 0 3968 > $O/cap.txt 2>&1 || { tail -20 $O/cap.txt; exit 1; }
cat $O/cap.txt
