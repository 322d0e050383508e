#!/bin/bash
# Run ON THE GPU BOX through gpurun (GRAFT_REPO_ROOT is set there): a missing variable or a failed step ends the script.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/arena_bench
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/arena_trace -o arena -- python3 $R/tools/arena_real_bench.py --plies 2 > $R/$O/arena_trace.log 2>&1 || { tail -20 $R/$O/arena_trace.log; exit 1; }
cd $R
python tools/trace_timeline.py $O/arena_trace k_compact -5 > $O/arena_timeline.txt 2>&1
cp $(find $O/arena_trace -name "*kernel_stats.csv" | head -1) $O/arena_kernel_stats.csv
rm -rf $O/arena_trace
cat $O/arena_timeline.txt
SECONDS=0; timeout -k 10 1000 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
echo "bench wall: $SECONDS s"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/arena_bench/bench.out") if l.startswith("{")][0])
print(json.dumps({k:d[k] for k in ("value","games_per_s","sims_per_s","ms_per_step","value_exact_fp32","games_per_s_exact_fp32","roofline_exact_fp32_frac","parity_max_err_f16x2","parity_max_err_f32","wall_breakdown")}))
print("roofline", {k:d["roofline"][k] for k in ("achieved","frac","avg_launch_ms","traffic")}, d["device_calibration"]["f16"]["sustained_tflops"], d["device_calibration"]["f32"]["sustained_tflops"], d["device_calibration"]["dominant_kernel_share_of_sustained"])
print("kernels", [(k["name"], round(k["ms_per_step"],2), round(k["frac"],3)) for k in d["kernels"]])
for k in ("cross_game_dedup","eval_cache","other_driver","all_layers_as_gemm"): print(k, round(d[k]["value"]), round(d[k].get("games_per_s",0),1))
c5=d["config5"]; print("config5", {k:c5[k] for k in ("games_per_s","us_per_sim_step","sample_mismatches","seconds")}, c5["roofline"]["avg_launch_ms"], c5["roofline"]["frac"], {k:v for k,v in c5["with_dedup_and_eval_cache"].items() if k!="note"})
c4=d["config4"]; print("config4", c4["value"], c4["games_per_s"], c4["roofline"]["frac"], c4["exact_fp32"]["value"], c4["exact_fp32"]["roofline"]["frac"], c4["parity_sample"]["max_abs_err_pi"], c4["parity_sample"]["max_abs_err_v"])
print("dropin", d["dropin_config0"]["gpu_dropin"]["seconds"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["one_game_per_thread"]["value"], d["cpu_baseline"]["one_thread"]["value"])
PY
