cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5f
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "latency or vendor or network_ or conv_pattern or f16x2 or precision or thread_workers or dropin or episode" > $O/pytest_sel.txt 2>&1 || { tail -40 $O/pytest_sel.txt; exit 1; }
tail -3 $O/pytest_sel.txt
bash tools/gpu/run5.sh 2>&1 | grep "forwards of\|^  *[0-9]* [a-zA-Z_]"
python tools/predict_latency.py
python - <<'PY'
import sys, json
sys.path.insert(0, ".")
import bench_legs
print(json.dumps(bench_legs.dropin_config0(512, "f32")["gpu_dropin"]))
PY
