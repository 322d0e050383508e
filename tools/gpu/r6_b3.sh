#!/bin/bash
# Run ON THE GPU BOX through gpurun: precision bf16x3 -- the probe (per-kernel times, error vs float64, bit-identity), its parity tests, and the
# self-play step with and without the batch cap.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r6b3
mkdir -p $O
[ "${SKIP_PROBE:-0}" = 1 ] || { timeout -k 10 300 python tools/b3_probe.py > $O/b3_probe.txt 2>&1 || { tail -20 $O/b3_probe.txt; exit 1; }; }
[ "${SKIP_PROBE:-0}" = 1 ] || cat $O/b3_probe.txt
timeout -k 10 900 python -m pytest tests/test_gpu_bench_config.py -x -q -m gpu -k "bf16x3" > $O/pytest_b3.txt 2>&1 || { tail -40 $O/pytest_b3.txt; exit 1; }
tail -3 $O/pytest_b3.txt
for cap in -1 0; do
  timeout -k 10 300 python bench.py --precision bf16x3 --steps 10 --warmup 2 --stagger-sims 8 --no-compare --no-cpu-baseline --batch-cap $cap > $O/bench_cap$cap.out 2> $O/bench_cap$cap.err || { tail -20 $O/bench_cap$cap.err; exit 1; }
  python - "$O/bench_cap$cap.out" <<'P'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print({k: d[k] for k in ("value", "ms_per_step", "games_per_s", "dtype")}, d["config"]["batch_cap"], {k: d["roofline"][k] for k in ("frac", "avg_launch_ms", "leaves_per_launch", "achieved")})
P
done
