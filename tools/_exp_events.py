import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from othellozero_amd import _lib
from othellozero_amd.NNet import NNetWrapper
from othellozero_amd.training import SelfPlayEngine, preferred_batch_cap
n, G, sims = 8, 4096, 100
net = NNetWrapper((n, n), num_channels_1=512, max_batch=G, seed=0, precision="f16x2")
eng = SelfPlayEngine(net, n, G, sims, 1.0, 1.0, 0.9, seed=1234, q_mode=1, refill=True, record_cap=G * 140, dedup=False)
eng.set_batch_cap(preferred_batch_cap(n, G, 512))
eng.stagger(8)
eng.run_steps(200)
for rep in range(3):
    for mode in (0, 1, 0, 1):
        net.profile(mode)
        eng.run_steps(100, sync=True)
        t0 = time.perf_counter()
        eng.run_steps(1000, sync=True)
        dt = time.perf_counter() - t0
        print(f"net.profile({mode}): {dt / 10 * 1e3:.2f} ms per 100 batches", flush=True)
