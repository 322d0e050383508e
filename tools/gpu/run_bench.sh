#!/bin/bash
# Run ON THE GPU BOX through gpurun (GRAFT_REPO_ROOT is set there): the driver's own bench command; the stdout line and bench_detail.json are kept
# under gpurun_out/r6bench, a digest of the detail is printed.  A missing variable or a failed step ends the script.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6bench
mkdir -p $O
SECONDS=0; timeout -k 10 1000 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
echo "bench wall: $SECONDS s; line bytes $(wc -c < $O/bench.out)"
cp bench_detail.json $O/bench_detail.json
cat $O/bench.out
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6bench/bench_detail.json"))
print("wall", d["wall_breakdown"])
print("kernels", [(k["name"], round(k["ms_per_step"], 2), round(k["frac"], 3)) for k in d["kernels"]])
for p, leg in d["precisions"].items():
    print(p, {k: leg[k] for k in ("value", "games_per_s", "ms_per_step")}, {k: leg["roofline"][k] for k in ("frac", "avg_launch_ms", "traffic")}, leg.get("parity_sample", {}).get("max_abs_err_pi"))
c5 = d["config5"]; print("config5", {k: c5[k] for k in ("games_per_s", "us_per_sim_step", "sample_mismatches", "seconds", "precision")}, c5["roofline"]["avg_launch_ms"], c5["roofline"]["frac"],
                         {k: v for k, v in c5["with_dedup_and_eval_cache"].items() if k != "note"})
c4 = d["config4"]; print("config4", c4["value"], c4["games_per_s"], c4["roofline"]["frac"], {p: (round(x["value"]), round(x["roofline"]["frac"], 3)) for p, x in c4["precisions"].items()}, c4.get("parity_sample", {}).get("max_abs_err_pi"))
print("dropin", d["dropin_config0"]["gpu_dropin"]["seconds"], "parity", d.get("parity_sample", {}).get("max_abs_err_pi"), d.get("parity_sample", {}).get("max_abs_err_v"))
PY
