cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r29
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_loop.py -x -q -m gpu > gpurun_out/r29/pytest_train.txt 2>&1 || { tail -40 gpurun_out/r29/pytest_train.txt; exit 1; }
tail -2 gpurun_out/r29/pytest_train.txt
for p in f16x2 f32; do timeout -k 10 300 python tools/train_bench.py --batch 32 --precision $p --steps 200 --fit-examples 6400 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$p', round(d['ms_per_step'],3), {k: round(v['ms_per_step'],3) for k,v in d['fit'].items()})"; done | tee gpurun_out/r29/fit.txt
