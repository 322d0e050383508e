cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r12
timeout -k 10 500 python tools/conv3_tile_probe.py 200 > gpurun_out/r12/tile_probe.txt 2>&1
echo "rc=$?" >> gpurun_out/r12/tile_probe.txt
cat gpurun_out/r12/tile_probe.txt
