#!/usr/bin/env python3
"""Soak: many generations of continuous self-play (device stub network: the tree / rules / driver kernels are the whole cost),
then a random sample of the completed games replayed by the CPU oracle, move for move.

    python tools/soak_check.py [--board 8] [--games 4096] [--sims 100] [--rounds 600] [--sample 200]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", type=int, default=8)
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=600)
    ap.add_argument("--sample", type=int, default=200)
    ap.add_argument("--q-mode", type=int, default=1)
    ap.add_argument("--driver", default="lockstep", choices=["lockstep", "free"], help="free: oz_selfplay_run_steps (rounds x sims batches)")
    ap.add_argument("--batch-cap", type=int, default=0, help="free-running driver: leaves per batch (0 = none)")
    args = ap.parse_args()
    import oracle
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    n, G = args.board, args.games
    gens = args.rounds // (n * n - 4) + 2
    eng = SelfPlayEngine(StubNetWrapper((n, n), 41, 0, max_batch=G), n, G, args.sims, 1.0, 1.0, 0.9, seed=77, first_game_id=0,
                         game_id_stride=G, q_mode=args.q_mode, refill=True, record_cap=G * gens * (n * n - 3))
    if args.batch_cap:
        eng.set_batch_cap(args.batch_cap)
    eng.stagger(8)
    t0 = time.perf_counter()
    done = 0
    while done < args.rounds:
        k = min(50, args.rounds - done)
        if args.driver == "free":
            eng.run_steps(k * args.sims)       # as many network batches as k move rounds (capped batches serve fewer games each)
        else:
            eng.run(k)
        done += k
        print(f"round {done}: {eng.stats()['games_completed']} games", flush=True)
    dt = time.perf_counter() - t0
    st = eng.stats()
    assert st["overflow"] == 0, st
    rec = eng.records()
    ids = np.unique(rec["game_id"])
    rs = np.random.RandomState(0)
    pick = rs.choice(ids, size=min(args.sample, ids.size), replace=False)
    offs = (np.arange(G) * (n * n - 4)) // G
    bad = 0
    for gid in pick:
        gid = int(gid)
        r = rec[rec["game_id"] == gid]
        pre = int(offs[gid]) if gid < G else 0                  # first-generation games played their first plies at 8 sims (stagger)
        ep = oracle.Mcts(n, 1.0, args.q_mode, salt=41).episode(args.sims, 1.0, 0.9, 77, gid, sims_pre=8, pre_plies=pre)
        ok = (np.array_equal(r["action"], ep["action"]) and np.array_equal(r["z"], ep["z"]) and np.array_equal(r["black"], ep["black"])
              and np.array_equal(r["white"], ep["white"]))
        bad += 0 if ok else 1
    print(json.dumps({"rounds": args.rounds, "seconds": dt, "games_completed": int(st["games_completed"]), "simulations": int(st["simulations"]),
                      "sims_per_s": st["simulations"] / dt, "records": int(rec.size), "distinct_games": int(ids.size),
                      "sampled": int(pick.size), "mismatching_games": bad}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
