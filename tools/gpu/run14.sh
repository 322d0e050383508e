cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r14
mkdir -p $O
timeout -k 10 300 python tools/conv3_tile_probe.py 100 > $O/tile_probe.txt 2>&1 || { tail -20 $O/tile_probe.txt; exit 1; }
cat $O/tile_probe.txt
timeout -k 10 500 python tools/pp_race_check.py > $O/race.txt 2>&1 || { tail -20 $O/race.txt; exit 1; }
tail -4 $O/race.txt
timeout -k 10 300 python tools/net_by_batch.py > $O/by_batch.txt 2>&1 || { tail -20 $O/by_batch.txt; exit 1; }
tail -30 $O/by_batch.txt
timeout -k 10 900 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_train.py -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
