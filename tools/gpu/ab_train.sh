#!/bin/bash
# Run ON THE GPU BOX through gpurun (GRAFT_REPO_ROOT is set there): a missing variable or a failed step ends the script.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
# ab_base = an untracked checkout of the control commit with its library built, placed in the repo root before the gpurun call:
#   git worktree add ab_base <control-commit> && (cd ab_base && python -m othellozero_amd.build)
[ -d ab_base ] || { echo "ab_train.sh: no ab_base/ checkout of the control commit (see the comment above)"; exit 1; }
for rep in 1 2 3; do for root in ab_base .; do ( cd $root && python tools/train_bench.py --batch 32 --precision f16x2 --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$root', round(d['ms_per_step'],4))" ); done; done | tee gpurun_out/ab/train_ab.txt
