cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r28
timeout -k 10 500 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r28/smoke.txt 2>&1; tail -3 gpurun_out/r28/smoke.txt
