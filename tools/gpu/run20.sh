cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r20
mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_config.py -x -q -k "not bench_ and not rccl and not ranks" > $O/pytest_sel.txt 2>&1 || { tail -40 $O/pytest_sel.txt; exit 1; }
tail -2 $O/pytest_sel.txt
timeout -k 10 300 python tools/predict_latency.py > $O/predict.txt 2>&1 || { tail -20 $O/predict.txt; exit 1; }
tail -12 $O/predict.txt
timeout -k 10 300 python tools/arena_real_bench.py --plies 6 --kernels > $O/arena6.txt 2>&1 || { tail -20 $O/arena6.txt; exit 1; }
tail -1 $O/arena6.txt
