#!/usr/bin/env python3
"""Times the REFERENCE's own Python self-play -- training.execute_episode (training.py:26-72) over its OthelloMCTS / MCTS / OthelloGame --
in the build container, where /root/reference lives (it never travels: bench.py only echoes the JSON this script writes).

TensorFlow / Keras are not installed, so the leaf evaluator behind the reference's `neural_network.predict(board)` call
(othelo_mcts.py:82-88) is the build's float32 C restatement of OthelloNN (oracle.CNet: the same 512-filter network, random Keras-default
weights, seed 0) -- one position per call, as the reference evaluates leaves; `Net.NNet` is stubbed for its `NeuralNets` enum only, the
way tests/golden/gen_golden.py does.  Everything else that runs is the reference's own code.

    python tools/time_reference_python.py            -> profiles/reference_python_cpu.json
"""
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    import gen_golden as G                     # imports the reference (Othello, MCTS, othelo_mcts, training, agents) with Net.NNet stubbed
    import oracle
    from othellozero_amd.weights import init_weights
    threads = min(oracle.lib().orc_nn_max_threads(), os.cpu_count() or 1)

    class CNetAsNNetWrapper:
        """duck-typed NNetWrapper: .network_type and .predict(board (n, n, 2)) -> (pi (n, n) float32, v float32)"""
        network_type = G.NeuralNets.ONN

        def __init__(self, n, channels=512):
            self.n = n
            self.net = oracle.CNet(init_weights(n, seed=0, channels=channels), n, channels=channels, nthreads=threads)
            self.calls = 0
            self.seconds = 0.0

        def predict(self, board):
            own, opp = G.pack(board)
            t0 = time.perf_counter()
            pi, v = self.net.forward(np.array([own], np.uint64), np.array([opp], np.uint64))
            self.seconds += time.perf_counter() - t0
            self.calls += 1
            return pi[0].reshape(self.n, self.n), np.float32(v[0])

    model = ""
    with open("/proc/cpuinfo") as f:
        model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    out = {"what": "the reference's own Python self-play (training.execute_episode -> OthelloMCTS.simulate -> MCTS.simulate -> OthelloGame), "
                   "timed in the build container; leaf evaluator = the build's float32 C restatement of the 512-filter OthelloNN behind "
                   "neural_network.predict (one position per call, OpenMP over the host cores), because TensorFlow is not installed",
           "measured_in": "build container (no GPU)", "cpu_model": model, "host_cpus": os.cpu_count(), "nn_threads": threads,
           "python": sys.version.split()[0], "numpy": np.__version__, "runs": []}
    for name, n, sims in (("BASELINE configs[0]: one 8x8 game, 25 sims/move", 8, 25), ("one 8x8 game, 100 sims/move (the metric's setting)", 8, 100),
                          ("one 6x6 game, 100 sims/move (configs[3])", 6, 100)):
        net = CNetAsNNetWrapper(n)
        random.seed(0); np.random.seed(0)
        net.predict(G.OthelloGame.initial_board(n))              # untimed: thread pool, first touch
        net.calls, net.seconds = 0, 0.0
        t0 = time.perf_counter()
        ex = G.training.execute_episode(n, net, 1, sims, 1, 0.9)
        dt = time.perf_counter() - t0
        moves = len(ex) // 8
        run = {"workload": name, "board": n, "sims_per_move": sims, "seconds": round(dt, 2), "moves": moves, "simulations": moves * sims,
               "node_expansions": net.calls, "node_expansions_per_s": net.calls / dt, "sims_per_s": moves * sims / dt, "games_per_s": 1.0 / dt,
               "seconds_in_leaf_evaluation": round(net.seconds, 2), "seconds_in_python_tree_and_rules": round(dt - net.seconds, 2)}
        out["runs"].append(run)
        print(json.dumps(run), flush=True)
    path = os.path.join(ROOT, "profiles", "reference_python_cpu.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
