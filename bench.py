#!/usr/bin/env python3
"""bench.py -- self-play hot path on MI355X: BASELINE.json configs[1]
(4096 concurrent 8x8 self-play games per GPU, 100 MCTS sims/move, random-init OthelloNN, batched leaf eval).

One *step* = one move round: every one of the 4096 resident games runs 100 lock-step simulations (select ->
leaf compaction -> one batched OthelloNN evaluation -> expand/backup, x100), then chooses, records and plays
one move; a finished game is replaced by a fresh one in the same slot (continuous self-play), so all slots
stay busy for the whole timed region and completed games are counted exactly.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

Prints ONE JSON line on rank 0: value = node expansions per second over all GPUs (the BASELINE metric; the
same line carries games/s and sims/s), `roofline` for the dominant kernel (precision f16x2: the conv3 implicit-GEMM
launch -- conv1 + conv2 are table lookups there; precision f32: the conv2 launch; HIP events on the launch stream)
and `cpu_baseline` (the CPU oracle -- the reference algorithm with batch-1 leaf evaluation -- timed on this host's
cores on a bounded sample).  In the timed region the network evaluates EVERY expansion (cross-game de-duplication
off); the rate with the library default (on) is measured afterwards and reported as `cross_game_dedup` (N = 1).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this host driver (set before torch loads)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_EXPANSION = {8: 566428672, 6: 270185472}          # SURVEY.md 8(d), whole OthelloNN forward
PEAK_F32_MATRIX_TFLOPS = 157.3                             # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32


def conv_flop_per_leaf(layer, n, C):
    """implicit-GEMM FLOP of conv2 (layer 2: n x n outputs) / conv3 (layer 3: (n-2) x (n-2) outputs) per leaf"""
    px = n * n if layer == 2 else (n - 2) * (n - 2)
    return 2 * px * (9 * C) * C                            # 8x8: conv2 301 989 888, conv3 169 869 312 FLOP


PEAK_F16_MATRIX_TFLOPS = 2500.0                            # MI355X_MICROARCH.md: ~2.5 PF dense fp16/bf16 MFMA


def roofline(precision, layer, achieved, layer_ms, launches, expansions, n, channels):
    """Dominant kernel.  precision f32: the conv2 implicit GEMM (53 % of the network's FLOPs).  precision f16x2 (default):
    conv1 + conv2 run as a table gather-sum (k_conv2_lut, ~0.15 ms), so the dominant launch is the conv3 implicit GEMM
    (64 % of the FLOPs that are left).  `achieved` is ALGORITHMIC fp32 TFLOP/s (2*M*K*N per launch / HIP-event time on
    the launch stream).  f32: v_mfma_f32_32x32x2_f32, peak 157.3.  f16x2: every fp32 product costs 3 fp16 MFMA
    products, so the matrix pipe executes 3x `achieved`; both fractions are reported against the 2.5 PFLOP/s dense fp16 peak."""
    hbm = None
    try:      # HBM-side bytes per leaf from the committed PMC profile of this kernel (profiles/), scaled per launch
        tj = json.load(open(os.path.join(ROOT, "profiles", f"conv{layer}_traffic_{precision}.json")))
        hbm = tj["hbm_bytes_per_leaf"] * expansions / max(launches, 1)
    except Exception:
        pass
    r = {"bound": "mfma", "achieved": achieved, "unit": "TFLOP/s", "traffic": hbm, "launches": int(launches),
         "avg_launch_ms": layer_ms / max(launches, 1), "flop_per_leaf": conv_flop_per_leaf(layer, n, channels)}
    if precision == "f32":
        r.update(kernel=("k_gemm_f32 (conv3: 3x3 valid, 512->512, 8x8 -> 6x6, implicit GEMM, v_mfma_f32_32x32x2_f32); conv1 + conv2 = k_conv2_lut_f32 table gather-sum"
                         if layer == 3 else "k_gemm_f32 (conv2: 3x3 same, 512->512, implicit GEMM, v_mfma_f32_32x32x2_f32)"),
                 peak=PEAK_F32_MATRIX_TFLOPS, frac=achieved / PEAK_F32_MATRIX_TFLOPS)
    else:
        pp = os.environ.get("OZ_H2_PP", "1") != "0"
        lut = os.environ.get("OZ_H2_LUT", "1") != "0" and pp
        tail = "implicit GEMM, f32 as 2xfp16 split, v_mfma_f32_16x16x32_f16, " + ("4-phase ping-pong loop" if pp else "one barrier per k-tile")
        if layer == 3:
            kernel = f"k_gemm_h2<{'H2MidPP' if pp else 'H2Mid'}> (conv3: 3x3 valid, 512->512, 8x8 -> 6x6, {tail}); conv1 + conv2 = k_conv2_lut table gather-sum"
        elif lut:
            kernel = f"k_gemm_h2<H2BigPPLut> (conv2: 3x3 same, 512->512, {tail}; A rows gathered from the conv1 pattern table)"
        else:
            kernel = f"k_gemm_h2<{'H2BigPP' if pp else 'H2Big'}> (conv2: 3x3 same, 512->512, {tail})"
        r.update(kernel=kernel, peak=PEAK_F16_MATRIX_TFLOPS, frac=achieved / PEAK_F16_MATRIX_TFLOPS,
                 mfma_products_per_fp32_product=3, matrix_pipe_tflops=3 * achieved,
                 matrix_pipe_frac=3 * achieved / PEAK_F16_MATRIX_TFLOPS,
                 vs_fp32_matrix_peak=achieved / PEAK_F32_MATRIX_TFLOPS)
    return r


def cpu_baseline(n, channels, sims, budget_s=12.0):
    """The oracle port of the reference path (sequential simulations, one game, batch-1 leaf evaluation by the
    float32 C restatement of OthelloNN on all host cores), timed on a bounded sample of about budget_s seconds."""
    import oracle
    from othellozero_amd.weights import init_weights
    w = init_weights(n, seed=0, channels=channels)
    threads = min(oracle.lib().orc_nn_max_threads(), 16)        # the box's CPU share for one GPU
    net = oracle.CNet(w, n, channels=channels, nthreads=threads)

    def run(game_id, max_moves):
        m = oracle.Mcts(n, 1.0, oracle.QMODE_F64, evaluator=net.evaluator())        # a fresh tree per episode (training.py:29)
        t0 = time.perf_counter()
        ep = m.episode(sims, 1.0, 0.9, 1234, game_id, max_moves=max_moves)
        return time.perf_counter() - t0, ep
    run(0, 1)                                                                        # untimed: thread pool / cache warm-up
    t, exp, plies, games = 0.0, 0, 0, 0
    while t < budget_s and games < 64:                                               # whole games until the budget is used
        dt, ep = run(games, n * n)
        t, exp, plies, games = t + dt, exp + ep["stats"]["expansions"], plies + ep["n_moves"], games + 1
    # the same port on ONE thread ("at 1 thread and at all host cores"): a few plies are enough
    net1 = oracle.CNet(w, n, channels=channels, nthreads=1)
    m1 = oracle.Mcts(n, 1.0, oracle.QMODE_F64, evaluator=net1.evaluator())
    t0 = time.perf_counter()
    ep1 = m1.episode(sims, 1.0, 0.9, 1234, 0, max_moves=2)
    t1 = time.perf_counter() - t0
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {
        "value": exp / t, "unit": "node-expansions/s", "cores": threads, "kind": "port",
        "sample": f"{plies} plies of {games} sequential {n}x{n} game(s) at {sims} sims/move "
                  f"({exp} expansions, {plies * sims} sims, {t:.1f} s), batch-1 leaf eval, OpenMP x{threads}",
        "sims_per_s": plies * sims / t,
        "value_1_thread": ep1["stats"]["expansions"] / t1, "sample_1_thread": f"first 2 plies of game 0 ({t1:.1f} s)",
        "cpu_model": model, "host_cpus": os.cpu_count(),
    }


def main():
    t_proc = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--board", type=int, default=8)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--precision", default="f16x2", choices=["f32", "f16x2"],
                    help="conv arithmetic: exact fp32 matrix cores, or f32 via 2 x fp16 split (same 1e-5 parity tolerance)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dedup-compare", "--no-compare", dest="no_dedup_compare", action="store_true",
                    help="skip the secondary measurements (de-duplication on; conv2 as a GEMM): profiling runs")
    ap.add_argument("--dedup", default="off", choices=["off", "on"],
                    help="cross-game leaf de-duplication in the timed region.  off (default for the headline): the network evaluates "
                         "every expansion -- no output is shared or cached; on: the library default (a board reached by several "
                         "games in the same step is evaluated once).  With off, the on-rate is measured afterwards and reported "
                         "next to it (N = 1 only)")
    ap.add_argument("--driver", default="lockstep", choices=["free", "lockstep"],
                    help="free: oz_selfplay_run_steps (every game runs on by itself, full leaf batches; identical records); "
                         "lockstep: oz_selfplay_run (one simulation per game per step, moves aligned)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the control flow)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses GPU 0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_device:
        local_rank = 0
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import torch
    import torch.distributed as dist
    from othellozero_amd import _lib
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.distributed import engine_records_tensor, gather_records
    from othellozero_amd.training import SelfPlayEngine

    _lib.require_gpu()
    _lib.check(_lib.load().oz_set_device(local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    n, G = args.board, args.games
    net = NNetWrapper((n, n), num_channels_1=args.channels, max_batch=G, seed=0, precision=args.precision)   # same weights on every rank
    def make_engine(dedup):
        os.environ["OZ_DEDUP"] = "1" if dedup else "0"          # read when the engine's search object is created
        return SelfPlayEngine(net, n, G, args.sims, 1.0, 1.0, 0.9, seed=1234, first_game_id=rank * G,
                              game_id_stride=world * G, q_mode=_lib.QMODE_F64, refill=True,
                              record_cap=int(G * (args.steps + args.warmup + 2) * 1.25))
    eng = make_engine(args.dedup == "on")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # one bench step = `sims` network batches of up to G leaves: a move round in lock step; in free-running mode the same
    # number of batches, each full (network-free simulations and moves ride along).  Measured: no throughput difference --
    # a batch's cost is proportional to its leaves, so filling the ~8 % empty slots buys nothing
    def advance(k, sync, eng=eng):
        if args.driver == "free":
            eng.run_steps(k * args.sims, sync=sync)
        else:
            eng.run(k, sync=sync)
    t_setup = time.perf_counter() - t_proc                      # imports, network + table build, engine allocation
    advance(args.warmup, True)
    eng.sync()
    if world > 1:           # warm-up of the exchange step too (communicator channels for both collectives), like the W untimed steps
        gather_records(torch.zeros((8, 48), dtype=torch.uint8, device=dev))
    net.profile(True)
    s0 = eng.stats()
    ev0 = eng.eval_time()
    barrier()
    t0 = time.perf_counter()
    advance(args.steps, False)
    eng.sync()
    pooled = gather_records(engine_records_tensor(eng, dev))       # the path's only exchange step
    barrier()
    dt = time.perf_counter() - t0
    s1 = eng.stats()
    ev1 = eng.eval_time()
    conv2_ms, conv2_launches = net.profile_read()               # the dominant launch: conv2, or conv3 when conv2 is a gather-sum
    layer = net.profiled_layer()

    d = {k: s1[k] - s0[k] for k in ("simulations", "expansions", "terminal_hits", "node_visits", "moves", "games_completed", "leaves_evaluated")}
    vec = torch.tensor([d["expansions"], d["simulations"], d["games_completed"], d["moves"], d["node_visits"]],
                       dtype=torch.float64, device=dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    exp_all, sims_all, games_all, moves_all, visits_all = (float(x) for x in vec.tolist())

    if rank == 0:
        # network work is counted per position actually evaluated (a board reached by several games in one step is evaluated once)
        flop_conv2 = d["leaves_evaluated"] * conv_flop_per_leaf(layer, n, args.channels)
        achieved = flop_conv2 / (conv2_ms * 1e-3) / 1e12 if conv2_ms > 0 else 0.0
        # FLOP the GPU executes per expansion: the reference network's, minus conv1 + conv2 when they run as table lookups
        flop_ref = FLOP_PER_EXPANSION.get(n, 0)
        flop_exec = flop_ref - (conv_flop_per_leaf(2, n, args.channels) + 2 * n * n * 18 * args.channels if layer == 3 else 0)
        nn_ms = ev1["ms"] - ev0["ms"]
        out = {
            "metric": "mcts_node_expansions_per_sec", "value": exp_all / dt, "unit": "node-expansions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{G} concurrent {n}x{n} self-play games per GPU, {args.sims} sims/move, batched leaf eval "
                            "(BASELINE configs[1]); step = " + (f"{args.sims} network batches of up to {G} leaves, games free-running "
                                                                  "(a game plays its move as soon as its simulations are complete; records identical to lock step), "
                                                                  if args.driver == "free" else
                                                                  "one move round (100 lock-step simulations + one move per game), ") +
                            f"finished games refilled; leaf evaluator = the reference's OthelloNN ({args.channels} filters), random init seed 0",
                "games_per_gpu": G, "sims_per_move": args.sims, "board": n,
                "q_mode": "float64 (NumPy 1.18.5 promotion)", "driver": args.driver, "parallelism": f"games sharded x{world}, all-gather of move records",
                "leaf_dedup": ("on: a board reached by several games in the same step is evaluated once (library default)" if args.dedup == "on" else
                               "off: the network evaluates every expansion, nothing is shared or cached between games "
                               "(the library default is on -- see cross_game_dedup for that rate)"),
            },
            "games_per_s": games_all / dt, "sims_per_s": sims_all / dt, "moves_per_s": moves_all / dt,
            "games_completed": int(games_all), "expansions": int(exp_all), "simulations": int(sims_all),
            "expansions_per_sim": exp_all / max(sims_all, 1), "node_visits_per_sim": visits_all / max(sims_all, 1),
            "pooled_records": int(pooled.shape[0]),
            "nn_forward_ms_total_rank0": nn_ms, "nn_fraction_of_wall_rank0": nn_ms * 1e-3 / dt,
            "leaves_evaluated_rank0": int(d["leaves_evaluated"]),
            "whole_net_tflops_rank0": d["leaves_evaluated"] * flop_exec / max(nn_ms * 1e-3, 1e-9) / 1e12,
            "flop_per_expansion": {"reference_network": flop_ref, "executed": flop_exec,
                                   "note": "executed < reference when conv1 + conv2 are evaluated as pattern-table lookups (exact refactoring, no GEMM)"},
            "roofline": roofline(args.precision, layer, achieved, conv2_ms, conv2_launches, d["leaves_evaluated"], n, args.channels),
            # the tree / rules side is latency-bound integer work, ~1.3 KB of algorithmic HBM bytes per simulation
            "tree_side_hbm": {"bytes_per_sim": 1300, "achieved_GBps": sims_all / dt * 1300 / 1e9, "peak_GBps": 8000.0,
                              "frac": sims_all / dt * 1300 / 8e12, "note": "not the binding roof; reported per SURVEY 8(d)"},
        }
        out["dtype"] = "f32" if args.precision == "f32" else "f32 (2xf16 split)"
        out["dtype_detail"] = ("fp32 operands and accumulators on v_mfma_f32_32x32x2_f32" if args.precision == "f32" else
                               "fp32 values carried as two fp16 planes, 3 fp16 MFMA products per fp32 product, fp32 accumulate; pi, v within 1e-5 of float64")
        wall = {"setup_s": round(t_setup, 2), "timed_region_s": round(dt, 2)}      # where this process's wall time goes (the driver clocks the whole run)
        out["wall_breakdown"] = wall
        if world == 1 and args.dedup == "off" and not args.no_dedup_compare:
            t_sec = time.perf_counter()
            # the same K steps with the library default (cross-game de-duplication on): identical records, fewer evaluations
            eng2 = make_engine(True)
            advance(args.warmup, True, eng2)
            eng2.sync()
            q0 = eng2.stats()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            advance(args.steps, False, eng2)
            eng2.sync()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            q1 = eng2.stats()
            out["cross_game_dedup"] = {
                "value": (q1["expansions"] - q0["expansions"]) / dt2, "unit": "node-expansions/s", "ms_per_step": dt2 / args.steps * 1e3,
                "expansions": int(q1["expansions"] - q0["expansions"]), "leaves_evaluated": int(q1["leaves_evaluated"] - q0["leaves_evaluated"]),
                "note": "same games, same records; concurrent games that reach the same board in a step share one network evaluation "
                        "(k_compact). Not the headline: `value` above evaluates every expansion"}
            wall["dedup_compare_s"] = round(time.perf_counter() - t_sec, 2)
        if world == 1 and layer == 3 and not args.no_dedup_compare:
            # the same K steps with conv1 / conv2 evaluated the plain way (conv1 kernel + conv2 as an MFMA implicit GEMM,
            # no pattern tables): what the table form buys, and a number for readers who want every layer as a GEMM
            t_sec = time.perf_counter()
            net.profile(False)
            net.set_tables(0)
            eng3 = make_engine(args.dedup == "on")
            advance(args.warmup, True, eng3)
            eng3.sync()
            g0 = eng3.stats()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            advance(args.steps, False, eng3)
            eng3.sync()
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t2
            g1 = eng3.stats()
            net.set_tables(-1)
            out["all_layers_as_gemm"] = {
                "value": (g1["expansions"] - g0["expansions"]) / dt3, "unit": "node-expansions/s", "ms_per_step": dt3 / args.steps * 1e3,
                "flop_per_expansion_executed": flop_ref,
                "note": "same games, conv1 as a kernel and conv2 as an MFMA implicit GEMM (oz_net_set_tables(net, 0)); "
                        "(pi, v) agree with the table form to 5e-7"}
            wall["gemm_compare_s"] = round(time.perf_counter() - t_sec, 2)
        if world == 1 and not args.no_cpu_baseline:
            t_sec = time.perf_counter()
            out["cpu_baseline"] = cpu_baseline(n, args.channels, args.sims)
            wall["cpu_baseline_s"] = round(time.perf_counter() - t_sec, 2)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
