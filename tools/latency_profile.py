import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from othellozero_amd.NNet import NNetWrapper
for prec in ("f16x2", "f32"):
    net = NNetWrapper((8, 8), num_channels_1=512, max_batch=1, seed=0, precision=prec)
    own = np.array([0x0000001008000000], np.uint64); opp = np.array([0x0000000810000000], np.uint64)
    for _ in range(20): net.predict_batch(own, opp)
    net.profile(2); net.profile_kernels(reset=True)
    for _ in range(200): net.predict_batch(own, opp)
    k = net.profile_kernels()
    print(prec, {n: round(ms / max(c, 1) * 1e3, 1) for n, (ms, c) in k.items()}, "sum us", round(sum(ms / max(c, 1) for ms, c in k.values()) * 1e3, 1))
