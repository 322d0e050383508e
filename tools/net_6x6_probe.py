"""per-kernel time of the 6x6 forward (BASELINE configs[3]) at 4096 positions, both precisions"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from othellozero_amd.NNet import NNetWrapper
n, G = 6, 4096
rs = np.random.RandomState(0)
valid = np.uint64(sum(1 << (r * 8 + c) for r in range(n) for c in range(n)))
own = rs.randint(0, 2**63, size=G, dtype=np.uint64) & rs.randint(0, 2**63, size=G, dtype=np.uint64) & valid
opp = rs.randint(0, 2**63, size=G, dtype=np.uint64) & rs.randint(0, 2**63, size=G, dtype=np.uint64) & valid & ~own
for prec in ("f16x2", "f32"):
    net = NNetWrapper((n, n), num_channels_1=512, max_batch=G, seed=0, precision=prec)
    for _ in range(3): net.predict_batch(own, opp)
    net.profile(2); net.profile_kernels(reset=True)
    for _ in range(30): net.predict_batch(own, opp)
    k = net.profile_kernels(); net.profile(0)
    print(prec, {a: round(ms / c * 1e3, 1) for a, (ms, c) in k.items() if c}, "sum", round(sum(ms / c for ms, c in k.values() if c) * 1e3, 1))
