cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/iter
R=$GRAFT_REPO_ROOT
timeout -k 10 500 python tools/iteration_bench.py 2>/dev/null | grep "^{" > gpurun_out/iter/iteration_f16x2.json; cat gpurun_out/iter/iteration_f16x2.json
timeout -k 10 500 python tools/iteration_bench.py --batched-eval 2>/dev/null | grep "^{" > gpurun_out/iter/iteration_f16x2_batched_eval.json; cat gpurun_out/iter/iteration_f16x2_batched_eval.json
