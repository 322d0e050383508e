set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5a
mkdir -p $O
python -c "
from othellozero_amd import _lib
for k in ('f32','f16','f32','f16'):
    print(k, _lib.mfma_rate(k, 50.0))
" > $O/calib.txt 2>&1
cat $O/calib.txt
timeout -k 10 900 python -m pytest tests/test_gpu_bench_config.py -x -q -k "big_tile or timed_shape" tests/test_gpu_parity.py -k "big_tile or timed_shape or range_guards or exchange_step or template_hooks" > $O/pytest_sel.txt 2>&1 || { tail -30 $O/pytest_sel.txt; exit 1; }
tail -3 $O/pytest_sel.txt
python tools/arena_real_bench.py --plies 4 --kernels > $O/arena_kernels.json 2>&1
cat $O/arena_kernels.json
python tools/arena_real_bench.py --plies 4 > $O/arena_plain.json 2>&1
cat $O/arena_plain.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/arena_prof -o run -- python3 tools/arena_real_bench.py --plies 2 > $O/arena_prof.log 2>&1
cp $(find $O/arena_prof -name '*kernel_stats.csv' | head -1) $O/arena_kernel_stats.csv
find $O/arena_prof -type f -delete
head -30 $O/arena_kernel_stats.csv
python tools/latency_profile.py > $O/latency.txt 2>&1; cat $O/latency.txt
python tools/predict_latency.py > $O/predict.txt 2>&1; cat $O/predict.txt
for p in f32 f16x2; do python tools/train_bench.py --batch 32 --precision $p --steps 200 > $O/train_b32_$p.json 2>&1; cat $O/train_b32_$p.json; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_prof -o run -- python3 tools/train_bench.py --batch 32 --precision f16x2 --steps 200 > $O/train_prof.log 2>&1
cp $(find $O/train_prof -name '*kernel_stats.csv' | head -1) $O/train_b32_f16x2_kernel_stats.csv
find $O/train_prof -type f -delete
head -50 $O/train_b32_f16x2_kernel_stats.csv
