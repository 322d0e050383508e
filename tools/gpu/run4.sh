set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5d
mkdir -p $O
timeout -k 10 1000 python bench.py > $O/bench.out 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5d/bench.out") if l.startswith("{")][0])
print(json.dumps({k:d[k] for k in ("value","games_per_s","ms_per_step","value_exact_fp32","roofline_exact_fp32_frac","device_calibration","per_rank","wall_breakdown")}, indent=0)[:3000])
print("roofline", {k:d["roofline"][k] for k in ("achieved","frac","avg_launch_ms","traffic")})
c5=d["config5"]; print("config5", {k:c5[k] for k in ("games_per_s","us_per_sim_step","sample_mismatches","seconds")}, c5["roofline"]["avg_launch_ms"], c5["roofline"]["frac"], c5["roofline"]["grid_note"])
print([(k["name"], round(k["us_per_sim_step"],1)) for k in c5["kernels"]])
c4=d["config4"]; print("config4", c4["value"], c4["games_per_s"], c4["roofline"]["frac"], c4["roofline"]["avg_launch_ms"], c4.get("parity_sample",{}).get("max_abs_err_pi"), c4["exact_fp32"]["value"], c4["exact_fp32"]["roofline"]["frac"], c4["exact_fp32"].get("parity_sample"))
print([(k["name"], round(k["ms_per_step"],2), round(k["frac"],3)) for k in c4["kernels"]])
print("dropin", d["dropin_config0"]["gpu_dropin"])
print("cpu", {k:d["cpu_baseline"][k] for k in ("value","cores","host_share")})
PY
