"""the 128 x 256 tile's two main loops (OZ_NET_OPT_LOW_LOOP_PHASES: 1 = one phase per k-tile on three LDS stages, 2 = round 5's 2-phase loop) on a
max_batch = 512 network -- the arena's: per-kernel time of the f16x2 forward at the batch sizes an arena step holds, and a bitwise comparison of (pi, v).
    python tools/low_loop_probe.py [rounds]"""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from othellozero_amd import _lib
from othellozero_amd.NNet import NNetWrapper

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n, G = 8, 512
net = NNetWrapper((n, n), num_channels_1=512, max_batch=G, seed=0, precision="f16x2")
rs = np.random.RandomState(5)
valid = np.uint64(0xFFFFFFFFFFFFFFFF)
own = rs.randint(0, 2**63, size=G, dtype=np.uint64) & rs.randint(0, 2**63, size=G, dtype=np.uint64)
opp = rs.randint(0, 2**63, size=G, dtype=np.uint64) & rs.randint(0, 2**63, size=G, dtype=np.uint64) & ~own
for count in (256, 384, 430, 455, 512):
    row, ref = {}, None
    for phases in (2, 1, 2, 1):
        net.set_option(_lib.NET_OPT_LOW_LOOP_PHASES, phases)
        for _ in range(5): pi, v = net.predict_batch(own[:count], opp[:count])
        net.profile(2); net.profile_kernels(reset=True)
        for _ in range(rounds): pi, v = net.predict_batch(own[:count], opp[:count])
        k = net.profile_kernels(); net.profile(0)
        per = {name: ms / c * 1e3 for name, (ms, c) in k.items() if c}
        if ref is None: ref = (pi.copy(), v.copy())
        same = bool(np.array_equal(ref[0].view(np.uint32), pi.view(np.uint32)) and np.array_equal(ref[1].view(np.uint32), v.view(np.uint32)))
        row.setdefault(str(phases), []).append({"tile": net.conv3_tile_rows(), "conv3": round(per.get("conv3", 0.0), 1), "conv4": round(per.get("conv4", 0.0), 1),
                                                "fc1": round(per.get("fc1", 0.0), 1), "sum": round(sum(per.values()), 1), "same_bits": same})
    print(count, json.dumps(row), flush=True)
