cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r13
mkdir -p $O
timeout -k 10 600 python - > $O/config5.txt 2>&1 <<'PY'
import sys, json
sys.path.insert(0, ".")
import bench_legs
o = bench_legs.config5_arena(512, "f16x2")
print(json.dumps({k: o[k] for k in ("games_per_s", "us_per_sim_step", "sample_mismatches")}), json.dumps({k: v for k, v in o["with_dedup_and_eval_cache"].items() if k != "note"}))
PY
tail -3 $O/config5.txt
