cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5g
mkdir -p $O
bench() { for p in f16x2; do python tools/train_bench.py --batch 32 --precision $p --steps 300 --fit-examples 6400 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $p', round(d['ms_per_step'],4), {k: round(v['ms_per_step'],4) for k,v in d['fit'].items()})"; done; }
bench s4
bench s4
for b in 256 1024; do python tools/train_bench.py --batch $b --precision f16x2 --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b', round(d['ms_per_step'],3))"; done
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -x -q > $O/pytest_train.txt 2>&1 || { tail -30 $O/pytest_train.txt; exit 1; }
tail -2 $O/pytest_train.txt
