import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from othellozero_amd.NNet import NNetWrapper
from othellozero_amd.weights import init_weights
w = init_weights(8, seed=3, channels=512, randomize_all=True)
t0 = time.perf_counter(); net = NNetWrapper((8, 8), num_channels_1=512, max_batch=64, precision="f16x2", weights=w); t1 = time.perf_counter()
net.set_weights(w); t2 = time.perf_counter()
net.set_weights(w); t3 = time.perf_counter()
f32 = NNetWrapper((8, 8), num_channels_1=512, max_batch=64, precision="f32", weights=w); t4 = time.perf_counter()
f32.set_weights(w); t5 = time.perf_counter()
print(f"f16x2 create+commit {t1-t0:.3f} s, re-commit {t2-t1:.3f} s, {t3-t2:.3f} s;  f32 create {t4-t3:.3f} s, re-commit {t5-t4:.3f} s")
