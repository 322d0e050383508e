#!/bin/bash
# Run ON THE GPU BOX through gpurun (GRAFT_REPO_ROOT is set there): a missing variable or a failed step ends the script.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r26
mkdir -p $O
timeout -k 10 550 python tools/soak_real_net.py --driver lockstep --sample 6 > $O/soak_lockstep.txt 2>&1 || { tail -20 $O/soak_lockstep.txt; exit 1; }
tail -3 $O/soak_lockstep.txt
timeout -k 10 550 python tools/soak_real_net.py --driver free --batch-cap 0 --sample 6 > $O/soak_free_nocap.txt 2>&1 || { tail -20 $O/soak_free_nocap.txt; exit 1; }
tail -3 $O/soak_free_nocap.txt

# precision bf16x3 (bench.py's default): the lock-step generation and the free-running driver as the bench runs it (no batch cap in this precision)
timeout -k 10 550 python tools/soak_real_net.py --precision bf16x3 --driver lockstep --sample 6 > $O/soak_bf16x3_lockstep.txt 2>&1 || { tail -20 $O/soak_bf16x3_lockstep.txt; exit 1; }
tail -3 $O/soak_bf16x3_lockstep.txt
timeout -k 10 550 python tools/soak_real_net.py --precision bf16x3 --driver free --sample 6 > $O/soak_bf16x3_free.txt 2>&1 || { tail -20 $O/soak_bf16x3_free.txt; exit 1; }
tail -3 $O/soak_bf16x3_free.txt
