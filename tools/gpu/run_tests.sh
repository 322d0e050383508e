cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5tests
timeout -k 10 1150 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r5tests/pytest_gpu.txt 2>&1
echo "rc=$?" >> gpurun_out/r5tests/pytest_gpu.txt
tail -30 gpurun_out/r5tests/pytest_gpu.txt
