"""conv3's three row tiles (OZ_NET_OPT_CONV3_TILE) on a max_batch = 512 network -- the arena's -- at the batch sizes an arena step holds:
per-kernel time of the f16x2 forward with each tile forced and with the forward's own choice, and a bitwise comparison of (pi, v).
    python tools/conv3_tile_probe.py [rounds]"""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from othellozero_amd import _lib
from othellozero_amd.NNet import NNetWrapper

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n, G = 8, 512
net = NNetWrapper((n, n), num_channels_1=512, max_batch=G, seed=0, precision="f16x2")
rng = np.random.default_rng(5)
cells = rng.integers(0, 3, size=(G, 64))
own = np.zeros(G, dtype=np.uint64)
opp = np.zeros(G, dtype=np.uint64)
for b in range(G):
    for c in range(64):
        if cells[b, c] == 1: own[b] |= np.uint64(1) << np.uint64(c)
        elif cells[b, c] == 2: opp[b] |= np.uint64(1) << np.uint64(c)
out = {}
for count in (192, 256, 341, 384, 430, 470, 512):
    ref = None
    row = {}
    for tile in (0, 128, 192, 256):
        net.set_option(_lib.NET_OPT_CONV3_TILE, tile)
        for _ in range(5): pi, v = net.predict_batch(own[:count], opp[:count])
        picked = net.conv3_tile_rows()
        net.profile(2); net.profile_kernels(reset=True)
        for _ in range(rounds): pi, v = net.predict_batch(own[:count], opp[:count])
        k = net.profile_kernels(); net.profile(0)
        per = {name: ms / c * 1e3 for name, (ms, c) in k.items() if c}
        if ref is None: ref = (pi.copy(), v.copy())
        same = bool(np.array_equal(ref[0].view(np.uint32), pi.view(np.uint32)) and np.array_equal(ref[1].view(np.uint32), v.view(np.uint32)))
        row[str(tile)] = {"picked": picked, "conv3_us": round(per.get("conv3", 0.0), 1), "sum_us": round(sum(per.values()), 1), "bit_identical": same}
    out[count] = row
    print(count, json.dumps(row), flush=True)
