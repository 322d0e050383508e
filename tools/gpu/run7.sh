cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5g
mkdir -p $O
bench() { for p in f16x2 f32; do python tools/train_bench.py --batch 32 --precision $p --steps 300 --fit-examples 6400 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $p', round(d['ms_per_step'],4), {k: round(v['ms_per_step'],4) for k,v in d['fit'].items()})"; done; }
bench now
bench now
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -x -q > $O/pytest_train.txt 2>&1 || { tail -30 $O/pytest_train.txt; exit 1; }
tail -2 $O/pytest_train.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o run -- python3 tools/train_bench.py --batch 32 --precision f16x2 --steps 60 > $O/train_prof.log 2>&1
python tools/trace_timeline.py $O/tr k_t_conv1_fwd -3 > $O/train_timeline4.txt
find $O/tr -type f -delete
grep "period\|k_t_heads \|conv1_wgrad\|sum_partials\|k_t_colreduce\|bnb_apply\|k_wgrad_f32" $O/train_timeline4.txt
