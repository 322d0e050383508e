"""Does the sustained clock of the f16x2 GEMMs follow the residual planes' bits?  The same network, the same positions, the activation window
moved by OZ_NET_OPT_ACT_TARGET_LOG2 / OZ_NET_OPT_W_TARGET_LOG2 (arguments: act,weights pairs; 9,10 = everything normal; -2,-2 = the default: maxima in
[2^-3, 2^-2), the residuals of smaller elements in fp16 subnormals = fewer significant bits): per-kernel time and error against float64."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from oracle import nn_numpy
    from othellozero_amd import _lib
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    from othellozero_amd.weights import init_weights
    n, G, cap = 8, 4096, 3640
    net = NNetWrapper((n, n), num_channels_1=512, max_batch=G, seed=0, precision="f16x2")
    eng = SelfPlayEngine(net, n, G, 8, 1.0, 1.0, 0.9, seed=1234, refill=True, record_cap=G * 80)
    eng.stagger(8)
    st = eng.state()
    own = np.where(st["player"] == 1, st["black"], st["white"])[:cap].copy()
    opp = np.where(st["player"] == 1, st["white"], st["black"])[:cap].copy()
    del eng
    rows = np.linspace(0, cap - 1, 256).astype(np.int64)
    pi64, v64 = nn_numpy.forward_chunked(init_weights(n, seed=0, channels=512), own[rows], opp[rows], n, chunk=128)
    targets = [tuple(int(y) for y in x.split(",")) for x in sys.argv[1:]] or [(9, 10), (-2, 10), (-2, -2)]
    for tgt, wt in targets:
        net.set_option(_lib.NET_OPT_ACT_TARGET_LOG2, tgt)
        net.set_option(_lib.NET_OPT_W_TARGET_LOG2, wt)
        net.set_option(_lib.NET_OPT_LOW_GUARD_LOG2, -100)
        net.commit()
        for _ in range(5):
            pi, v = net.predict_batch(own, opp)
        net.profile(2); net.profile_kernels(reset=True)
        for _ in range(150):
            net.predict_batch(own, opp)
        k = net.profile_kernels(); net.profile(0)
        err = max(float(np.abs(pi.reshape(cap, -1)[rows] - pi64).max()), float(np.abs(v[rows] - v64).max()))
        print("target", tgt, "weights", wt, {a: round(ms / c * 1e3, 1) for a, (ms, c) in k.items() if c}, "sum", round(sum(ms / c for ms, c in k.values() if c) * 1e3, 1),
              "max err vs float64", f"{err:.2e}", flush=True)


if __name__ == "__main__":
    main()
