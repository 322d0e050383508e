#!/usr/bin/env python3
"""Observed behaviour of the reference's training.duel_between_neural_networks / training.evaluate_neural_network
(training.py:75-118), which workers.py:14-15 imports next to execute_episode.  Run in the build container only
(imports /root/reference through gen_golden.py's stub machinery); writes DATA only: tests/golden/drivers_misc.json.

Both functions pass the (agent, points) tuple that agents.duel_between_agents returns on as if it were the agent:
the first raises KeyError at its `agents[agent_winner]` lookup, the second never counts a win."""
import json
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G                     # imports the reference with the Net.NNet stub

out = {}
n, sims = 4, 6
net_a, net_b = G.StubNet(n, 5, 0, True), G.StubNet(n, 6, 0, True)
random.seed(1); np.random.seed(1)
try:
    r = G.training.duel_between_neural_networks(n, net_a, net_b, 1, sims)
    out["duel_between_neural_networks"] = {"returns": r}
except Exception as e:                      # noqa: BLE001
    out["duel_between_neural_networks"] = {"raises": type(e).__name__}
random.seed(2); np.random.seed(2)
wins = G.training.evaluate_neural_network(n, 5, net_a, sims, 1, G.agents.RandomOthelloAgent, ())
out["evaluate_neural_network"] = {"returns": wins, "iterations": 5}
json.dump(out, open(os.path.join(G.OUT, "drivers_misc.json"), "w"), indent=1)
print(out)
