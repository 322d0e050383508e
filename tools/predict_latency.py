"""End-to-end latency of NNetWrapper.predict_batch for one position (host call + uploads + forward + downloads)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from othellozero_amd.NNet import NNetWrapper
for prec in ("f16x2", "f32"):
    net = NNetWrapper((8, 8), num_channels_1=512, max_batch=1, seed=0, precision=prec)
    own = np.array([0x0000000810000000], dtype=np.uint64); opp = np.array([0x0000001008000000], dtype=np.uint64)
    for _ in range(20): net.predict_batch(own, opp)
    t0 = time.perf_counter()
    N = 500
    for _ in range(N): net.predict_batch(own, opp)
    dt = (time.perf_counter() - t0) / N
    print(f"{prec}: {dt * 1e6:.1f} us per predict_batch(1)   (GPU forward alone: {net.time_forward(1, 50) * 1e3:.1f} us)", flush=True)
