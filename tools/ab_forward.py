"""A/B of two builds of the library on ONE device in ONE gpurun call: per-kernel time of the f16x2 forward at the bench's launch size (3640
positions on the max_batch = 4096 network), the same positions for both.  Each build runs in its own child process with its own package.
    python tools/ab_forward.py <repo root A> <repo root B> [more roots ...] [rounds]"""
import json
import os
import subprocess
import sys

CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
from othellozero_amd.NNet import NNetWrapper
from othellozero_amd.training import SelfPlayEngine
n, G, cap = 8, 4096, 3640
net = NNetWrapper((n, n), num_channels_1=512, max_batch=G, seed=0, precision="f16x2")
# positions of staggered self-play (every ply of the game), produced by the engine itself at 8 sims/move
eng = SelfPlayEngine(net, n, G, 8, 1.0, 1.0, 0.9, seed=1234, refill=True, record_cap=G * 80)
eng.stagger(8)
st = eng.state()
own = np.where(st["player"] == 1, st["black"], st["white"])[:cap].copy()
opp = np.where(st["player"] == 1, st["white"], st["black"])[:cap].copy()
del eng
for _ in range(5): net.predict_batch(own, opp)
net.profile(2); net.profile_kernels(reset=True)
for _ in range(int(sys.argv[2])): net.predict_batch(own, opp)
k = net.profile_kernels(); net.profile(0)
print("RESULT " + json.dumps({name: ms / c * 1e3 for name, (ms, c) in k.items() if c}))
'''


def main():
    roots = [a for a in sys.argv[1:] if not a.isdigit()]
    rounds = next((int(a) for a in sys.argv[1:] if a.isdigit()), 200)
    res = {}
    for rep in range(2):                      # A B A B: drift of the box shows up as a difference between the repeats
        for r in roots:
            out = subprocess.run([sys.executable, "-c", CHILD, r, str(rounds)], capture_output=True, text=True, timeout=600)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")]
            if not line:
                print("child failed for", r, out.stderr[-2000:])
                sys.exit(1)
            k = json.loads(line[0][7:])
            res.setdefault(r, []).append(k)
            print(os.path.basename(os.path.abspath(r)) or r, {a: round(b, 1) for a, b in k.items()}, "sum", round(sum(k.values()), 1), flush=True)


if __name__ == "__main__":
    main()
