cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5k
mkdir -p $O
timeout -k 10 400 python tools/trained_net_self_check.py --iterations 3 --episodes 64 --sims 25 > $O/trained_self_check.txt 2>&1; tail -5 $O/trained_self_check.txt
bash tools/gpu/run5.sh 2>&1 | grep "forwards of\|^  *[0-9]* [a-zA-Z_]" | tee $O/predict_gaps_f32.txt
for b in 32 1024; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tp$b -o run -- python3 tools/train_bench.py --batch $b --precision f16x2 --steps 100 > $O/train_b$b.log 2>&1
  cp $(find $O/tp$b -name '*kernel_stats.csv' | head -1) $O/train_b${b}_f16x2_kernel_stats.csv
  find $O/tp$b -type f -delete
  grep '^{' $O/train_b$b.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b (under rocprofv3)', d['ms_per_step'])"
done
for b in 32 256 1024; do for p in f16x2 f32; do python tools/train_bench.py --batch $b --precision $p --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b $p', round(d['ms_per_step'],3), round(d['tflops_fp32'],1))"; done; done
