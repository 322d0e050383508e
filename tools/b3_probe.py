"""precision bf16x3 at the bench's launch size (3640 positions on the max_batch = 4096 network): per-kernel time, max error vs float64 on 256 rows
(beside the exact-fp32 and f16x2 networks on the same rows), bit-identity of a position across call sizes and batch positions.
    python tools/b3_probe.py [board] [positions]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nn_numpy
from othellozero_amd.NNet import NNetWrapper
from othellozero_amd.weights import init_weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 3640
G = 4096
rs = np.random.RandomState(0)
valid = np.uint64(sum(1 << (r * 8 + c) for r in range(n) for c in range(n)))
own = rs.randint(0, 2**63, size=cap, dtype=np.uint64) & rs.randint(0, 2**63, size=cap, dtype=np.uint64) & valid
opp = rs.randint(0, 2**63, size=cap, dtype=np.uint64) & rs.randint(0, 2**63, size=cap, dtype=np.uint64) & ~own & valid
rows = np.linspace(0, cap - 1, 256).astype(np.int64)
for seed, rand_all in ((0, False), (5, True)):
    w = init_weights(n, seed=seed, channels=512, randomize_all=rand_all)
    pi64, v64 = nn_numpy.forward_chunked(w, own[rows], opp[rows], n, chunk=128)
    res = {}
    for prec in ("bf16x3", "f32", "f16x2"):
        net = NNetWrapper((n, n), num_channels_1=512, max_batch=G, weights=w, precision=prec)
        for _ in range(2): pi, v = net.predict_batch(own, opp)
        net.profile(2); net.profile_kernels(reset=True)
        for _ in range(20): net.predict_batch(own, opp)
        k = net.profile_kernels(); net.profile(0)
        err = max(float(np.abs(pi.reshape(cap, -1)[rows] - pi64).max()), float(np.abs(v[rows] - v64).max()))
        res[prec] = (pi, v)
        print(f"seed {seed} {prec:7s}", {a: round(ms / c * 1e3, 1) for a, (ms, c) in k.items() if c}, "sum us", round(sum(ms / c for ms, c in k.values() if c) * 1e3, 1),
              "err vs f64", f"{err:.2e}", "tile", net.conv3_tile_rows(), flush=True)
        if prec == "bf16x3":
            # the two tiles of k_gemm_b3 (OZ_NET_OPT_B3_TILE): same products in the same order -> the same bits; per-kernel times of each
            from othellozero_amd import _lib
            for tile in (128, 256):
                net.set_option(_lib.NET_OPT_B3_TILE, tile)
                for _ in range(2): pt, vt = net.predict_batch(own, opp)
                net.profile(2); net.profile_kernels(reset=True)
                for _ in range(20): net.predict_batch(own, opp)
                kk = net.profile_kernels(); net.profile(0)
                print(f"   tile {tile}:", {a: round(ms / c * 1e3, 1) for a, (ms, c) in kk.items() if c and a in ("conv3", "conv4", "fc1")}, "conv3 rows", net.conv3_tile_rows(),
                      "| same bits as the default:", bool(np.array_equal(pt, pi) and np.array_equal(vt, v)), flush=True)
            net.set_option(_lib.NET_OPT_B3_TILE, 0)
            # a position's (pi, v) must not depend on the call it sits in: a shorter call, and the same positions in another order
            p2, v2 = net.predict_batch(own[:300], opp[:300])
            perm = rs.permutation(cap)
            p3, v3 = net.predict_batch(own[perm], opp[perm])
            print("   bit-identical: shorter call", bool(np.array_equal(p2, pi[:300]) and np.array_equal(v2, v[:300])),
                  "| permuted batch", bool(np.array_equal(p3, pi[perm]) and np.array_equal(v3, v[perm])), flush=True)
        del net
    d = max(float(np.abs(res["bf16x3"][0] - res["f32"][0]).max()), float(np.abs(res["bf16x3"][1] - res["f32"][1]).max()))
    print(f"seed {seed}: max |bf16x3 - f32| over all {cap} positions = {d:.2e}", flush=True)
