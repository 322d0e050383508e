cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5i
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_bench_config.py -x -q -k "config5" > $O/pytest_sel.txt 2>&1 || { tail -40 $O/pytest_sel.txt; exit 1; }
tail -2 $O/pytest_sel.txt
python - <<'PY'
import sys, json
sys.path.insert(0, ".")
import bench_legs
o = bench_legs.config5_arena(512, "f16x2")
print(json.dumps({k: o[k] for k in ("games_per_s", "us_per_sim_step", "sample_mismatches")}), json.dumps({k: v for k, v in o["with_dedup_and_eval_cache"].items() if k != "note"}))
PY
