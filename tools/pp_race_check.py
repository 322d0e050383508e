"""Race / equivalence screen of the ping-pong conv loop: the one-barrier loop (OZ_H2_PP=0) and the ping-pong loop accumulate
every output in the same order, so their (pi, v) must be BIT-identical; a DMA-visibility race would show up as a rare
mismatch.  The same holds for conv1 as a pattern-table lookup inside conv2's gather (default) against the conv1 kernel
(OZ_H2_LUT=0): the table rows are the rows the kernel would write.  Runs each loop in its own process over several batch sizes / boards, many repetitions, and compares.
    python tools/pp_race_check.py            (on the GPU box)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from othellozero_amd.NNet import NNetWrapper, NeuralNets
from othellozero_amd.weights import init_weights
out = {}
for n, C, B, reps, cin in ((8, 512, 4096, 12, 2), (8, 512, 1000, 8, 2), (8, 256, 257, 8, 2), (6, 512, 4096, 8, 2), (6, 256, 33, 8, 2),
                           (8, 512, 1, 4, 2), (8, 512, 777, 4, 1), (6, 512, 100, 4, 1)):
    net = NNetWrapper((n, n), num_channels_1=C, max_batch=B, precision="f16x2", network=NeuralNets.ONN if cin == 2 else NeuralNets.BNN,
                      weights=init_weights(n, seed=3, channels=C, randomize_all=True, in_channels=cin))
    rs = np.random.RandomState(n * 1000 + B)
    valid = np.uint64(sum(1 << (r * 8 + c) for r in range(n) for c in range(n)))
    for rep in range(reps):
        own = rs.randint(0, 2**63, size=B, dtype=np.uint64) & valid
        opp = rs.randint(0, 2**63, size=B, dtype=np.uint64) & valid & ~own
        pi, v = net.predict_batch(own, opp)
        out[f"{n}_{C}_{B}_{cin}_{rep}_pi"] = pi
        out[f"{n}_{C}_{B}_{cin}_{rep}_v"] = v
np.savez(sys.argv[2], **out)
'''
def run(tag, extra):
    path = f"/tmp/pp_race_{tag}.npz"
    subprocess.run([sys.executable, "-c", WORKER, ROOT, path], env=dict(os.environ, **extra), check=True, timeout=600)
    return np.load(path)


# group A: conv2 as a GEMM (OZ_H2_T2=0) -- every loop / staging variant must agree to the bit
A = [run(tag, dict(extra, OZ_H2_T2="0")) for tag, extra in (
    ("simple", {"OZ_H2_PP": "0"}), ("pingpong", {"OZ_H2_PP": "1"}), ("pingpong_fc1small", {"OZ_H2_PP": "1", "OZ_H2_FC1PP": "0"}),
    ("pingpong_conv1kernel", {"OZ_H2_PP": "1", "OZ_H2_LUT": "0"}), ("pingpong_conv3_3phase", {"OZ_H2_PP": "1", "OZ_H2_PP3": "1"}),
    ("small_tiles_2stage", {"OZ_H2_PP": "1", "OZ_H2_STAGES": "2"}), ("small_tiles_3stage", {"OZ_H2_PP": "1", "OZ_H2_STAGES": "3"}))]
# group B: the default (conv1 + conv2 as the table gather-sum): the two loops still agree to the bit (conv3, conv4, fc1),
# and the gather-sum agrees with the GEMM to rounding (another summation order of the same products)
B = [run(tag, extra) for tag, extra in (("t2_simple", {"OZ_H2_PP": "0"}), ("t2_pingpong", {"OZ_H2_PP": "1"}))]
bad = [k for k in A[0].files if not all(np.array_equal(A[0][k], r[k]) for r in A[1:])]
bad += [k for k in B[0].files if not np.array_equal(B[0][k], B[1][k])]
gap = max(float(np.abs(A[0][k] - B[0][k]).max()) for k in A[0].files)
print(f"{len(A[0].files)} arrays x {len(A)} + {len(B)} loop configurations compared, {len(bad)} differ", bad[:5])
print(f"gather-sum vs GEMM conv2: max |difference| of (pi, v) = {gap:.3g}")
sys.exit(1 if bad or gap > 2e-6 else 0)
