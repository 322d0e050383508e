"""Times the training step (oz_trainer_forward_backward + oz_trainer_apply) on one GPU.

    python tools/train_bench.py [--board 8] [--channels 512] [--batch 32] [--steps 50]

Prints one JSON line: steps/s, examples/s and the fp32 TFLOP/s of the step (3 x the forward contraction FLOPs:
forward + data gradient + weight gradient; the first layer has no data gradient).  Not part of bench.py -- the
BASELINE metric is self-play node expansions/s; this is the side measurement for SURVEY 8(f) item 2.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", type=int, default=8)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    from othellozero_amd.trainer import Trainer
    from othellozero_amd.weights import init_weights
    n, C, B = args.board, args.channels, args.batch
    tr = Trainer(n, C, 2, max_batch=B, seed=1)
    tr.set_weights(init_weights(n, seed=0, channels=C))
    rs = np.random.RandomState(0)
    valid = np.uint64(sum(1 << (r * 8 + c) for r in range(n) for c in range(n)))
    own = rs.randint(0, 2**63, size=B, dtype=np.uint64) & valid
    opp = rs.randint(0, 2**63, size=B, dtype=np.uint64) & valid & ~own
    pi = np.zeros((B, n * n), np.float32)
    pi[np.arange(B), rs.randint(0, n * n, B)] = 1
    z = rs.choice([-1.0, 1.0], B).astype(np.float32)
    for _ in range(args.warmup):
        tr.forward_backward(own, opp, pi, z)
        tr.apply()
    tr.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.forward_backward(own, opp, pi, z)
        tr.apply()
    tr.sync()
    dt = time.perf_counter() - t0
    F = (n - 4) ** 2 * C
    fwd = 2 * (n * n * 18 * C + n * n * 9 * C * C + (n - 2) ** 2 * 9 * C * C + (n - 4) ** 2 * 9 * C * C + F * 1024 + 1024 * 512 + 512 * (n * n + 1))
    flop = (3 * fwd - 2 * n * n * 18 * C) * B
    params = 18 * C + 3 * 9 * C * C + F * 1024 + 1024 * 512
    print(json.dumps({"metric": "training_steps_per_sec", "value": args.steps / dt, "examples_per_s": args.steps * B / dt,
                      "ms_per_step": 1e3 * dt / args.steps, "batch": B, "board": n, "channels": C,
                      "tflops_fp32": flop * args.steps / dt / 1e12, "flop_per_step": flop,
                      "adam_bytes_per_step": params * 4 * 7, "last_loss": loss[0], "dtype": "f32", "data": "synthetic"}))


if __name__ == "__main__":
    main()
