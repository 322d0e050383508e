cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r27
IDS_AB=1 timeout -k 10 300 python tools/conv2_lut_probe.py 50 f16x2 > gpurun_out/r27/ids_ab.txt 2>&1; cat gpurun_out/r27/ids_ab.txt
