"""Per-kernel time of the OthelloNN forward at medium batch sizes (the arena's 512 games, 1024, 2048): where the launches of a small batch go.
    python tools/net_by_batch.py      (on the GPU box)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from othellozero_amd.NNet import NNetWrapper
rs = np.random.RandomState(0)
for B in (512, 1024, 2048, 4096):
    own = rs.randint(0, 2**63, size=B, dtype=np.uint64); opp = rs.randint(0, 2**63, size=B, dtype=np.uint64) & ~own
    net = NNetWrapper((8, 8), num_channels_1=512, max_batch=B, seed=0, precision="f16x2")
    if B == 4096: own, opp = own[:3640], opp[:3640]          # bench.py's batch cap
    for _ in range(3): net.predict_batch(own, opp)
    net.profile(2); net.profile_kernels(reset=True)
    for _ in range(50): net.predict_batch(own, opp)
    k = net.profile_kernels(); net.profile(0)
    print(B, {n: round(ms / c * 1e3, 1) for n, (ms, c) in k.items()}, "sum us", round(sum(ms / c for ms, c in k.values()) * 1e3, 1), "linear share of 3640-batch:", round(2130 * B / 3640, 1))
