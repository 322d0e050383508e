"""Diagnostics of the f16x2 commit-time scaling (oz_net_get_scaling): exponent ranges per tensor / layer, commit time, and whether
a self-play run or a batch of positions raises a guard.  python tools/scaling_probe.py [--board 8] [--channels 512] [--games 4096]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", type=int, default=8)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=2)
    a = ap.parse_args()
    from othellozero_amd import _lib
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    n = a.board
    t0 = time.time()
    net = NNetWrapper((n, n), num_channels_1=a.channels, max_batch=a.games, seed=0, precision="f16x2")
    t1 = time.time()
    net.commit()
    t2 = time.time()
    print(f"create+commit {t1 - t0:.3f} s, re-commit {t2 - t1:.3f} s")
    names = ["act1", "act2", "act3", "act4", "f1", "w conv2", "w conv3", "w conv4", "w fc1", "w fc2"]
    for k in range(10):
        e = net.scaling(k)
        print(f"{names[k]:8s} exponents min {e.min():4d} median {int(np.median(e)):4d} max {e.max():4d}")
    lib = _lib.load()

    def check(tag):
        rc = lib.oz_net_check(net._h)
        print(tag, "check ->", rc, lib.oz_last_error().decode() if rc else "ok")
    check("after commit")
    rs = np.random.RandomState(0)
    own = rs.randint(0, 2 ** 62, size=64, dtype=np.int64).astype(np.uint64)
    opp = rs.randint(0, 2 ** 62, size=64, dtype=np.int64).astype(np.uint64) & ~own
    if n == 6:
        valid = np.uint64(sum(1 << (r * 8 + c) for r in range(6) for c in range(6)))
        own &= valid; opp &= valid
    for cnt in (1, 7, 64):
        try:
            net.predict_batch(own[:cnt], opp[:cnt])
            print("predict", cnt, "ok")
        except _lib.OzError as e:
            print("predict", cnt, "->", e)
    eng = SelfPlayEngine(net, n, a.games, a.sims, 1.0, 1.0, 0.9, seed=1234, first_game_id=0, q_mode=1, dedup=False)
    for r in range(a.rounds):
        try:
            eng.run(1)
            eng.stats()
            print("round", r, "ok")
        except _lib.OzError as e:
            print("round", r, "->", e)
        check(f"after round {r}")


if __name__ == "__main__":
    main()
