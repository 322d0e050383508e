"""Probe: time the conv1 + conv2 table gather (k_lut_ids + k_conv2_lut_xcd) on realistic positions (every ply of stub-network games), 3640 per call:
    python tools/conv2_lut_probe.py [iterations] [precisions]      e.g.  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/conv2_lut_probe.py 4 f16x2"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from othellozero_amd.NNet import NNetWrapper, StubNetWrapper
from othellozero_amd.training import SelfPlayEngine
n, G, cap = 8, 4096, 3640
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
PRECS = sys.argv[2].split(",") if len(sys.argv) > 2 else ("f16x2", "f32")
eng = SelfPlayEngine(StubNetWrapper((n, n), 17, 0, max_batch=G), n, G, 8, 1.0, 1.0, 0.9, seed=3, refill=True, record_cap=G * 80)
eng.stagger(4)
st = eng.state()
own = np.where(st["player"] == 1, st["black"], st["white"])[:cap]
opp = np.where(st["player"] == 1, st["white"], st["black"])[:cap]
print("plies", st["ply"].min(), st["ply"].max(), flush=True)
for prec in PRECS:
    net = NNetWrapper((n, n), num_channels_1=512, max_batch=G, seed=0, precision=prec)
    for _ in range(3):
        net.predict_batch(own, opp)
    net.profile(2); net.profile_kernels(reset=True)
    for _ in range(ITERS):
        net.predict_batch(own, opp)
    k = net.profile_kernels(); net.profile(0)
    print(prec, "gather %.4f ms  ids %.4f ms  conv3 %.4f ms" % tuple(k[x][0] / k[x][1] for x in ("conv2", "input", "conv3")), flush=True)
