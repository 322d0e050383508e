cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5h
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_bench_config.py -x -q -k "network_at or big_tile or gemm_form" > $O/pytest_sel.txt 2>&1 || { tail -40 $O/pytest_sel.txt; exit 1; }
tail -2 $O/pytest_sel.txt
bash tools/profile_round.sh f16x2 r5 > $O/profile_f16x2.log 2>&1 || { tail -20 $O/profile_f16x2.log; exit 1; }
tail -5 $O/profile_f16x2.log
bash tools/profile_round.sh f32 r5 > $O/profile_f32.log 2>&1 || { tail -20 $O/profile_f32.log; exit 1; }
tail -5 $O/profile_f32.log
