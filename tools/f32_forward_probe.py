"""per-kernel time of the exact-fp32 forward at the bench's launch size (3640 positions on the max_batch = 4096 network) + max error vs float64"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nn_numpy
from othellozero_amd.NNet import NNetWrapper
from othellozero_amd.weights import init_weights
n, G, cap = 8, 4096, 3640
net = NNetWrapper((n, n), num_channels_1=512, max_batch=G, seed=0, precision="f32")
rs = np.random.RandomState(0)
own = rs.randint(0, 2**63, size=cap, dtype=np.uint64) & rs.randint(0, 2**63, size=cap, dtype=np.uint64)
opp = rs.randint(0, 2**63, size=cap, dtype=np.uint64) & rs.randint(0, 2**63, size=cap, dtype=np.uint64) & ~own
for _ in range(3): pi, v = net.predict_batch(own, opp)
net.profile(2); net.profile_kernels(reset=True)
for _ in range(30): net.predict_batch(own, opp)
k = net.profile_kernels(); net.profile(0)
rows = np.linspace(0, cap - 1, 128).astype(np.int64)
pi64, v64 = nn_numpy.forward_chunked(init_weights(n, seed=0, channels=512), own[rows], opp[rows], n, chunk=128)
err = max(float(np.abs(pi.reshape(cap, -1)[rows] - pi64).max()), float(np.abs(v[rows] - v64).max()))
print({a: round(ms / c * 1e3, 1) for a, (ms, c) in k.items() if c}, "sum", round(sum(ms / c for ms, c in k.values() if c) * 1e3, 1), "err", f"{err:.2e}")
