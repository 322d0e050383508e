cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
for rep in 1 2 3; do for root in ab_base .; do ( cd $root && python tools/train_bench.py --batch 32 --precision f16x2 --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$root', round(d['ms_per_step'],4))" ); done; done | tee gpurun_out/ab/train_ab.txt
