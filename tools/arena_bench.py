#!/usr/bin/env python3
"""BASELINE configs[4]: 800 sims/move arena evaluation, 512 parallel games, deterministic play (temperature 0, tie stream).
With the device stub network the leaf evaluation costs nothing, so this times the TREE side alone -- select / compaction /
expand / backup at 800 simulations per move, two search trees per game (one per agent, agents.py:44-68).

    python tools/arena_bench.py [--games 512] [--sims 800] [--board 8] [--check 2]

Prints one JSON line: simulations/s, games/s, and (--check k) the first k games re-played by the CPU oracle bit for bit."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=512)
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--board", type=int, default=8)
    ap.add_argument("--check", type=int, default=2)
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    import numpy as np
    from othellozero_amd import _lib
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.agents import arena_batch
    _lib.require_gpu()
    n, G = args.board, args.games
    a, b = StubNetWrapper((n, n), 301, 0, max_batch=G), StubNetWrapper((n, n), 302, 0, max_batch=G)
    best, r = None, None
    for rep in range(args.reps):                               # the first repetition also pays allocation / code load
        t0 = time.perf_counter()
        r = arena_batch(a, b, n, G, args.sims, 1.0, seed=11, first_game_id=0, q_mode=1)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    moves = int(r["n_moves"].sum())
    out = {"workload": f"{G} parallel {n}x{n} arena games, {args.sims} sims/move per agent, stub network (tree side only), deterministic",
           "seconds": best, "moves": moves, "simulations": moves * args.sims, "sims_per_s": moves * args.sims / best,
           "games_per_s": G / best, "moves_per_s": moves / best,
           "tree_side_hbm_GBps_at_1300B_per_sim": moves * args.sims * 1300 / best / 1e9}
    if args.check:
        import oracle
        ok = 0
        for gi in range(args.check):
            o = oracle.arena(oracle.Mcts(n, 1.0, 1, salt=301), oracle.Mcts(n, 1.0, 1, salt=302), args.sims, 11, gi)
            k = o["n_moves"]
            assert int(r["n_moves"][gi]) == k and np.array_equal(r["actions"][gi][:k], o["action"]), gi
            assert (int(r["winner"][gi]), int(r["points"][gi])) == (o["winner"], o["points"]), gi
            ok += 1
        out["games_checked_bit_exact_vs_oracle"] = ok
    print(json.dumps(out))


if __name__ == "__main__":
    main()
