#!/usr/bin/env python3
"""Whole games at the bench configuration (4096 x 8x8 x 100 sims, 512-filter OthelloNN, max_batch 4096) replayed by the oracle's search fed
with the GPU network's own (pi, v): `--sample` games of the first generation, move for move.   python tools/soak_real_net.py [--precision f16x2 | f32 | bf16x3]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="f16x2")
ap.add_argument("--sample", type=int, default=6)
ap.add_argument("--games", type=int, default=4096)
ap.add_argument("--driver", default="lockstep", choices=["lockstep", "free"],
                help="free: bench.py's driver -- staggered slots (8 sims/move on the stagger plies), oz_selfplay_run_steps with the batch cap, refilled games")
ap.add_argument("--batch-cap", type=int, default=-1, help="free driver: leaves per batch (-1 = training.preferred_batch_cap, 0 = none)")
args = ap.parse_args()
import oracle
from othellozero_amd.NNet import NNetWrapper
from othellozero_amd.training import SelfPlayEngine
n, G, sims = 8, args.games, 100
net = NNetWrapper((n, n), num_channels_1=512, max_batch=G, seed=0, precision=args.precision)
pre_plies = np.zeros(G, int)
if args.driver == "free":
    from othellozero_amd.training import preferred_batch_cap
    eng = SelfPlayEngine(net, n, G, sims, 1.0, 1.0, 0.9, seed=1234, first_game_id=0, game_id_stride=G, q_mode=1, refill=True, record_cap=G * 140,
                         dedup=False)                                # the bench headline: one evaluation per expansion
    cap = preferred_batch_cap(n, G, 512, args.precision) if args.batch_cap < 0 else args.batch_cap
    eng.set_batch_cap(cap)
    eng.stagger(8)
    pre_plies = (np.arange(G) * (n * n - 4)) // G            # slot g played its first pre_plies[g] plies at 8 simulations each
    t0 = time.perf_counter()
    for _ in range(74):                                      # 74 x 100 batches: every first-generation game ends (60 plies, capped batches)
        eng.run_steps(100)
    dt = time.perf_counter() - t0
    rec = eng.records()
else:
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
    ep = oracle.Mcts(n, 1.0, 1, evaluator=ev).episode(sims, 1.0, 0.9, 1234, int(gi), sims_pre=8, pre_plies=int(pre_plies[gi]))
    r = rec[rec["game_id"] == gi]
    r = r[np.argsort(r["ply"])]
    ok = np.array_equal(r["action"], ep["action"]) and np.array_equal(r["z"], ep["z"]) and np.array_equal(r["black"], ep["black"])
    bad += 0 if ok else 1
    print("game", gi, "ok" if ok else "MISMATCH", len(r), flush=True)
print(json.dumps({"precision": args.precision, "driver": args.driver, "games": G, "seconds": dt, "games_completed": int(st["games_completed"]), "expansions_per_s": st["expansions"] / dt,
                  "sampled": args.sample, "mismatching_games": bad}))
sys.exit(1 if bad else 0)
