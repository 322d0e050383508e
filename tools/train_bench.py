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
    ap.add_argument("--precision", default="f32", choices=("f32", "f16x2"), help="arithmetic of the 3x3 layers' forward / data-gradient GEMMs")
    ap.add_argument("--fit-examples", type=int, default=0,
                    help="also time trainer.fit over this many examples (2 epochs) both ways: step-wise host loop vs HBM-resident data set")
    args = ap.parse_args()
    from othellozero_amd.trainer import Trainer
    from othellozero_amd.weights import init_weights
    n, C, B = args.board, args.channels, args.batch
    tr = Trainer(n, C, 2, max_batch=B, seed=1, precision=args.precision)
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
    fit_cmp = None
    if args.fit_examples:
        from othellozero_amd.trainer import fit
        N = args.fit_examples
        o2 = rs.randint(0, 2**63, size=N, dtype=np.uint64) & valid
        p2 = rs.randint(0, 2**63, size=N, dtype=np.uint64) & valid & ~o2
        pi2 = np.zeros((N, n * n), np.float32)
        pi2[np.arange(N), rs.randint(0, n * n, N)] = 1
        z2 = rs.choice([-1.0, 1.0], N).astype(np.float32)
        fit_cmp = {}
        for name, resident in (("stepwise_host_loop", False), ("resident_dataset", True)):
            fit(tr, o2[:B * 4], p2[:B * 4], pi2[:B * 4], z2[:B * 4], batch_size=B, epochs=1, resident=resident)      # warm-up
            t1 = time.perf_counter()
            fit(tr, o2, p2, pi2, z2, batch_size=B, epochs=2, shuffle_seed=3, resident=resident)
            d = time.perf_counter() - t1
            steps = 2 * ((N + B - 1) // B)
            fit_cmp[name] = {"seconds": d, "steps": steps, "ms_per_step": 1e3 * d / steps}
    print(json.dumps({"metric": "training_steps_per_sec", "value": args.steps / dt, "examples_per_s": args.steps * B / dt, "fit": fit_cmp,
                      "ms_per_step": 1e3 * dt / args.steps, "batch": B, "board": n, "channels": C,
                      "tflops_fp32": flop * args.steps / dt / 1e12, "flop_per_step": flop,
                      # the step as a whole against the fp32 matrix-core roof (v_mfma_f32_32x32x2_f32, MI355X_MICROARCH.md): algorithmic
                      # contraction FLOP of forward + data gradient + weight gradient / wall time of the whole step (BN, losses, Adam included)
                      "roofline": {"bound": "mfma", "achieved": flop * args.steps / dt / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                                   "frac": flop * args.steps / dt / 1e12 / 157.3, "traffic": None,
                                   "note": ("fp32 matrix roof; precision f16x2 runs the 3x3 layers' forward and data gradient as 3 fp16 MFMA products per "
                                            "fp32 product (roof 2500 / 3 = 833 TFLOP/s for that share of the FLOP), the weight gradients and dense layers "
                                            "on the fp32 matrix cores -- frac may exceed 1") if args.precision == "f16x2" else "fp32 matrix roof"},
                      "adam_bytes_per_step": params * 4 * 7, "last_loss": loss[0], "dtype": args.precision, "data": "synthetic"}))


if __name__ == "__main__":
    main()
