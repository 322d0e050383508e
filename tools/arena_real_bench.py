#!/usr/bin/env python3
"""BASELINE configs[4] with two REAL 512-filter networks for a bounded number of plies: the regime of 512-leaf network batches
(one agent searches per round).  Prints one JSON line: seconds per simulation step, per-kernel HIP-event times of both networks
(oz_net_profile 2) when --kernels is given.  Run under `rocprofv3 --kernel-trace --stats` for the tree kernels as well.

    python tools/arena_real_bench.py [--plies 4] [--games 512] [--sims 800] [--precision f16x2] [--kernels] [--dedup]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plies", type=int, default=4)
    ap.add_argument("--games", type=int, default=512)
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--precision", default="f16x2")
    ap.add_argument("--kernels", action="store_true")
    ap.add_argument("--dedup", action="store_true")
    args = ap.parse_args()
    from othellozero_amd import _lib
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.agents import arena_batch
    _lib.require_gpu()
    n, G = 8, args.games
    nets = [NNetWrapper((n, n), num_channels_1=512, max_batch=G, seed=sd, precision=args.precision) for sd in (0, 1)]
    arena_batch(nets[0], nets[1], n, G, 8, 1.0, seed=11, first_game_id=0, q_mode=1, max_rounds=2)
    if args.kernels:
        for nt in nets:
            nt.profile(2); nt.profile_kernels(reset=True)
    t0 = time.perf_counter()
    r = arena_batch(nets[0], nets[1], n, G, args.sims, 1.0, seed=11, first_game_id=0, q_mode=1, max_rounds=args.plies, dedup=args.dedup)
    dt = time.perf_counter() - t0
    st = r["stats_black"] + r["stats_white"]
    steps = args.plies * args.sims
    plies_played = float(r["n_moves"].sum()) / G
    out = {"rounds": args.plies, "plies_per_game": plies_played,  "games": G, "sims": args.sims, "precision": args.precision, "seconds": dt, "sims_per_s": float(st[0]) / dt,
           "expansions_per_s": float(st[2]) / dt, "us_per_step": dt / max(steps, 1) * 1e6, "steps": steps,
           "games_per_s_if_60_plies": G / (dt / plies_played * 60), "leaves_evaluated": int(r["leaves_evaluated"])}
    if args.kernels:
        k = {}
        for nt in nets:
            for name, (ms, cnt) in nt.profile_kernels().items():
                a = k.setdefault(name, [0.0, 0])
                a[0] += ms; a[1] += cnt
        out["kernels_us_per_launch"] = {name: round(ms / max(cnt, 1) * 1e3, 2) for name, (ms, cnt) in k.items()}
        out["kernels_sum_us"] = round(sum(ms / max(cnt, 1) for ms, cnt in k.values()) * 1e3, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
