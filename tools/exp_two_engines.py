"""Experiment: the 4096 games of a GPU as K engines of 4096/K games, each with its own network instance and stream, driven
from K host threads: while one engine is in its small latency-bound kernels (select, compact, expand, heads, ids) or in the
bandwidth-bound table gather, the others' GEMMs fill the chip.   python tools/exp_two_engines.py [K] [rounds]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from othellozero_amd import _lib
from othellozero_amd.NNet import NNetWrapper
from othellozero_amd.training import SelfPlayEngine
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 32
G, n = 4096, 8
g = G // K
nets = [NNetWrapper((n, n), num_channels_1=512, max_batch=g, seed=0, precision="f16x2") for _ in range(K)]
engs = [SelfPlayEngine(nets[k], n, g, 100, 1.0, 1.0, 0.9, seed=1234, first_game_id=k * g, game_id_stride=G, q_mode=_lib.QMODE_F64,
                       refill=True, record_cap=int(g * (rounds + 4) * 1.25), dedup=False) for k in range(K)]
def run(e, r):
    e.run(r, sync=False); e.sync()
def all_run(r):
    th = [threading.Thread(target=run, args=(e, r)) for e in engs]
    [t.start() for t in th]; [t.join() for t in th]
all_run(1)
s0 = [e.stats() for e in engs]
t0 = time.perf_counter()
all_run(rounds)
dt = time.perf_counter() - t0
s1 = [e.stats() for e in engs]
exp = sum(b["expansions"] - a["expansions"] for a, b in zip(s0, s1))
print(f"K={K}: {exp / dt:,.0f} expansions/s  ({dt / rounds * 1e3:.1f} ms per move round, {rounds} rounds)", flush=True)
