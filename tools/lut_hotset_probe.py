"""How much of the table gather's record traffic a small hot set would take: the frequency of the (tap, pattern id) records conv2's gather reads
on bench-like positions (every ply of self-play), and the share of the reads the K most frequent records cover.
    python tools/lut_hotset_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from othellozero_amd.NNet import StubNetWrapper
from othellozero_amd.training import SelfPlayEngine
n, G, cap = 8, 4096, 3640
eng = SelfPlayEngine(StubNetWrapper((n, n), 17, 0, max_batch=G), n, G, 8, 1.0, 1.0, 0.9, seed=3, refill=True, record_cap=G * 80)
eng.stagger(4)
st = eng.state()
own = np.where(st["player"] == 1, st["black"], st["white"])[:cap].astype(np.uint64)
opp = np.where(st["player"] == 1, st["white"], st["black"])[:cap].astype(np.uint64)
bits = np.arange(64, dtype=np.uint64)
cells = (((own[:, None] >> bits) & np.uint64(1)) + 2 * ((opp[:, None] >> bits) & np.uint64(1))).astype(np.int64).reshape(cap, 8, 8)
pad = np.zeros((cap, 10, 10), np.int64); pad[:, 1:9, 1:9] = cells
ids = np.zeros((cap, 8, 8), np.int64); pw = 1
for ky in range(3):
    for kx in range(3):
        ids += pw * pad[:, ky:ky + 8, kx:kx + 8]; pw *= 3
# conv2 pixel p, tap t reads record (t, ids[p + t]) or the zero record off the board
idp = np.full((cap, 10, 10), -1, np.int64); idp[:, 1:9, 1:9] = ids
total = cap * 64 * 9
counts = {}
off = 0
for t in range(9):
    v = idp[:, t // 3:t // 3 + 8, t % 3:t % 3 + 8].ravel()
    off += int((v < 0).sum())
    u, c = np.unique(v[v >= 0], return_counts=True)
    for a, b in zip(u, c): counts[(t, int(a))] = int(b)
freq = np.array(sorted(counts.values(), reverse=True), dtype=np.float64)
print("positions", cap, "plies", int(st["ply"].min()), int(st["ply"].max()), "record reads", total, "off-board (zero record)", round(off / total, 4), "distinct (tap, id) records", len(counts))
for K in (9, 18, 28, 64, 128, 256, 1024):
    print("top", K, "records cover", round(freq[:K].sum() / total, 4), "of the reads;  with the zero record", round((freq[:K].sum() + off) / total, 4))
top = sorted(counts.items(), key=lambda kv: -kv[1])[:12]
print("most frequent:", [(t, i, round(c / total, 4)) for (t, i), c in top])
