"""Race / equivalence screen of the ping-pong conv loop: the one-barrier loop (oz_net_set_option OZ_NET_OPT_SIMPLE_LOOP) and the
4-phase ping-pong loop accumulate every output in the same order, so their (pi, v) must be BIT-identical; an LDS-DMA visibility
race in the ping-pong schedule would show up as a rare mismatch.  The same holds for conv1 as a pattern-table lookup inside
conv2's operand gather (oz_net_set_tables 1) against the conv1 kernel (0): the table rows are the rows the kernel would write.
The default form (conv1 + conv2 as the table gather-sum, mode 2) adds the same products in another order: equal to rounding.
Several batch sizes / boards / networks, many repetitions.
    python tools/pp_race_check.py            (on the GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from othellozero_amd import _lib                                    # noqa: E402
from othellozero_amd.NNet import NNetWrapper, NeuralNets            # noqa: E402
from othellozero_amd.weights import init_weights                    # noqa: E402

CASES = ((8, 512, 4096, 12, 2), (8, 512, 3640, 8, 2), (8, 512, 1000, 8, 2), (8, 256, 257, 8, 2), (6, 512, 4096, 8, 2), (6, 256, 33, 8, 2),
         (8, 512, 1, 4, 2), (8, 512, 777, 4, 1), (6, 512, 100, 4, 1))
# (tables mode, simple loop): group A = conv2 as a GEMM, every variant bit-identical; group B = the default gather-sum form
GROUP_A = ((0, 1), (0, 0), (1, 0))
GROUP_B = ((2, 1), (2, 0))
compared, bad, gap = 0, [], 0.0
for n, C, B, reps, cin in CASES:
    net = NNetWrapper((n, n), num_channels_1=C, max_batch=B, precision="f16x2", network=NeuralNets.ONN if cin == 2 else NeuralNets.BNN,
                      weights=init_weights(n, seed=3, channels=C, randomize_all=True, in_channels=cin))
    rs = np.random.RandomState(n * 1000 + B)
    valid = np.uint64(sum(1 << (r * 8 + c) for r in range(n) for c in range(n)))
    for rep in range(reps):
        own = rs.randint(0, 2**63, size=B, dtype=np.uint64) & valid
        opp = rs.randint(0, 2**63, size=B, dtype=np.uint64) & valid & ~own
        out = {}
        for tables, simple in GROUP_A + GROUP_B:
            net.set_tables(tables)
            net.set_option(_lib.NET_OPT_SIMPLE_LOOP, simple)
            out[(tables, simple)] = net.predict_batch(own, opp)
        ref_a, ref_b = out[GROUP_A[0]], out[GROUP_B[0]]
        for key in GROUP_A[1:]:
            compared += 1
            if not (np.array_equal(out[key][0], ref_a[0]) and np.array_equal(out[key][1], ref_a[1])):
                bad.append((n, C, B, cin, rep, key))
        compared += 1
        if not (np.array_equal(out[GROUP_B[1]][0], ref_b[0]) and np.array_equal(out[GROUP_B[1]][1], ref_b[1])):
            bad.append((n, C, B, cin, rep, GROUP_B[1]))
        gap = max(gap, float(np.abs(ref_a[0] - ref_b[0]).max()), float(np.abs(ref_a[1] - ref_b[1]).max()))
    del net
# precision bf16x3 (k_gemm_b3: one tile, one loop -- no alternative schedule to compare with): the SAME call repeated must reproduce its own bits, and a
# position's (pi, v) must not depend on where it sits in the batch; an LDS-DMA visibility race in the 2-phase schedule would show up as a rare mismatch
b3_compared = 0
for n, C, B, reps in ((8, 512, 4096, 10), (8, 512, 3640, 10), (8, 256, 777, 10), (6, 512, 4096, 10), (8, 512, 130, 10)):
    net = NNetWrapper((n, n), num_channels_1=C, max_batch=max(B, 128), precision="bf16x3", weights=init_weights(n, seed=4, channels=C, randomize_all=True))
    rs = np.random.RandomState(n * 77 + B)
    valid = np.uint64(sum(1 << (r * 8 + c) for r in range(n) for c in range(n)))
    own = rs.randint(0, 2**63, size=B, dtype=np.uint64) & valid
    opp = rs.randint(0, 2**63, size=B, dtype=np.uint64) & valid & ~own
    ref = net.predict_batch(own, opp)
    for rep in range(reps):
        perm = rs.permutation(B)
        a = net.predict_batch(own, opp)
        b = net.predict_batch(own[perm], opp[perm])
        b3_compared += 2
        if not (np.array_equal(a[0], ref[0]) and np.array_equal(a[1], ref[1])):
            bad.append((n, C, B, "bf16x3", rep, "repeat"))
        if not (np.array_equal(b[0], ref[0][perm]) and np.array_equal(b[1], ref[1][perm])):
            bad.append((n, C, B, "bf16x3", rep, "permuted"))
    del net
print(f"bf16x3: {b3_compared} repeated / permuted calls over 5 networks compared with their first call")
print(f"{compared} comparisons over {len(CASES)} networks x {len(GROUP_A) + len(GROUP_B)} loop / table configurations, {len(bad)} differ", bad[:5])
print(f"gather-sum vs GEMM conv2: max |difference| of (pi, v) = {gap:.3g}")
sys.exit(1 if bad or gap > 2e-6 else 0)
