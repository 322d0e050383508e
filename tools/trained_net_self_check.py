#!/usr/bin/env python3
"""A TRAINED-network data point for the f16x2 commit-time self-check limit: a few iterations of the reference's loop (self-play
episodes -> fit on the replay buffer) on the GPU, then the trained weights committed in precision f16x2 with the self-check in measure-only
mode: D = max |f16x2 - exact fp32| on the calibration positions (the limit is 8e-6), next to the error of both precisions against the float64
oracle on self-play positions.

    python tools/trained_net_self_check.py [--board 8] [--channels 512] [--iterations 3] [--episodes 64] [--sims 25]"""
import argparse
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", type=int, default=8)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--iterations", type=int, default=3)
    ap.add_argument("--episodes", type=int, default=64)
    ap.add_argument("--sims", type=int, default=25)
    args = ap.parse_args()
    from oracle import nn_numpy
    from othellozero_amd import _lib
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.training import selfplay_batch
    from othellozero_amd.loop import examples_from_records
    n, C_ = args.board, args.channels
    random.seed(0); np.random.seed(0)
    net = NNetWrapper((n, n), num_channels_1=C_, max_batch=args.episodes, seed=0, precision="f32", epochs=3)
    out = []
    for it in range(args.iterations):
        rec = selfplay_batch(net, n, args.episodes, args.sims, 1.0, 1.0, 0.9, seed=100 + it)
        ex = examples_from_records(rec, n, alias_final=False)
        hist = net.train(ex, seed=it)
        w = net.get_weights()
        own = np.where(rec["player"] == 1, rec["black"], rec["white"])[:256]
        opp = np.where(rec["player"] == 1, rec["white"], rec["black"])[:256]
        pi64, v64 = nn_numpy.forward_chunked(w, own, opp, n, chunk=64)
        row = {"iteration": it + 1, "examples": len(ex), "loss": hist.history["loss"][-1]}
        p32, v32 = net.predict_batch(own, opp)
        row["E32"] = float(max(np.abs(p32.reshape(own.size, -1) - pi64).max(), np.abs(v32 - v64).max()))
        h = NNetWrapper.__new__(NNetWrapper)                    # an f16x2 twin of the trained weights, self-check in measure-only mode
        import ctypes as C
        from othellozero_amd.NNet import _NetHandle, NeuralNets
        _NetHandle.__init__(h)
        h.board_size_x = h.board_size_y = n; h.num_channels = C_; h.max_batch = 256; h.in_channels = 2
        h.network_type = NeuralNets.ONN; h.precision = "f16x2"
        lib = _lib.load()
        _lib.check(lib.oz_net_create(C.byref(h._h), n, C_, 256))
        _lib.check(lib.oz_net_set_precision(h._h, 1))
        _lib.check(lib.oz_net_set_option(h._h, _lib.NET_OPT_SELF_CHECK, 2))
        h.set_weights(w)
        p16, v16 = h.predict_batch(own, opp)
        row["E16"] = float(max(np.abs(p16.reshape(own.size, -1) - pi64).max(), np.abs(v16 - v64).max()))
        dpi, dv, npos = h.self_check()
        row.update(D_pi=dpi, D_v=dv, D=max(dpi, dv), positions=npos, guard=h.self_check_guard(), limit=8e-6)
        out.append(row)
        print(json.dumps(row), flush=True)
    print(json.dumps({"board": n, "channels": C_, "trained_network_self_check": out}))


if __name__ == "__main__":
    main()
