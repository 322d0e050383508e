#!/usr/bin/env python3
"""Whole games at the bench configuration (4096 x 8x8 x 100 sims, 512-filter OthelloNN, max_batch 4096) replayed by the oracle's search fed
with the GPU network's own (pi, v): `--sample` games of the first generation, move for move.   python tools/soak_real_net.py [--precision f16x2]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="f16x2")
ap.add_argument("--sample", type=int, default=6)
ap.add_argument("--games", type=int, default=4096)
args = ap.parse_args()
import oracle
from othellozero_amd.NNet import NNetWrapper
from othellozero_amd.training import SelfPlayEngine
n, G, sims = 8, args.games, 100
net = NNetWrapper((n, n), num_channels_1=512, max_batch=G, seed=0, precision=args.precision)
eng = SelfPlayEngine(net, n, G, sims, 1.0, 1.0, 0.9, seed=1234, first_game_id=0, q_mode=1)
t0 = time.perf_counter()
rec = eng.play_to_end()
dt = time.perf_counter() - t0
st = eng.stats()
cache = {}
def ev(own, opp, nn):
    if (own, opp) not in cache:
        p, v = net.predict_batch([own], [opp])
        cache[(own, opp)] = (p[0].ravel(), float(v[0]))
    return cache[(own, opp)]
bad = 0
for gi in np.linspace(0, G - 1, args.sample).astype(int):
    ep = oracle.Mcts(n, 1.0, 1, evaluator=ev).episode(sims, 1.0, 0.9, 1234, int(gi))
    r = rec[rec["game_id"] == gi]
    ok = np.array_equal(r["action"], ep["action"]) and np.array_equal(r["z"], ep["z"]) and np.array_equal(r["black"], ep["black"])
    bad += 0 if ok else 1
    print("game", gi, "ok" if ok else "MISMATCH", len(r), flush=True)
print(json.dumps({"precision": args.precision, "games": G, "seconds": dt, "games_completed": int(st["games_completed"]), "expansions_per_s": st["expansions"] / dt,
                  "sampled": args.sample, "mismatching_games": bad}))
sys.exit(1 if bad else 0)
