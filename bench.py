#!/usr/bin/env python3
"""bench.py -- self-play hot path on MI355X: BASELINE.json configs[1]
(4096 concurrent 8x8 self-play games per GPU, 100 MCTS sims/move, random-init OthelloNN, batched leaf eval).

One *step* = 100 network batches (one per simulation of a move).  Every batch: each of the 4096 resident games advances on its
own -- plays, records its move when its 100 simulations are complete (a finished game is replaced by a fresh one in the same
slot: continuous self-play), runs simulations that end on finished boards, descends to its next first-visit leaf -- the leaves
are compacted into one batch (at most 3640 of them: conv3's grid is then 4.0 rounds of the chip; a leaf without a slot waits for
the next batch), ONE batched OthelloNN evaluation, expand / backup (oz_selfplay_run_steps, the free-running driver; every game's
simulations, moves and records are exactly those of the lock-step driver, `--driver lockstep`, whose rate rides along as
`other_driver`).  Before the timed region the slots are spread over the plies of a game (oz_selfplay_stagger: slot g starts
(g*60)/4096 plies into its first game, played untimed by the same searched self-play), so the engine is in the steady state of a
long-running service: games complete in any window, and games/s, sims/s and expansions/s are all measured.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher environment (RANK / WORLD_SIZE unset) this process only SPAWNS the N ranks (one child
process per GPU, RCCL rendezvous on 127.0.0.1) and relays rank 0's JSON line -- it never touches the GPU itself; under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks run directly.  A realised world
size different from --gpus is an error (exit 2), never a silent 1-GPU run.

Prints ONE JSON line of <= 6 KB on rank 0 (strict JSON, contract keys first; bench_legs.compact_line): value = node expansions per second
over all GPUs (the BASELINE metric; the same line carries games/s and sims/s) in the arithmetic `--precision` names -- default bf16x3: fp32
values carried exactly as three bf16 planes, six bf16 MFMA products per fp32 product (fp32-class results; exact fp32 on the fp32 matrix cores,
literally what the reference computes in, Net/NNet.py:85, rides beside it as value_f32 / roofline_f32) -- `roofline` for the dominant kernel (HIP events on the launch
stream, in the timed region; traffic from two rocprofv3 --pmc child passes of this run) and `cpu_baseline` (the CPU oracle -- the reference
algorithm with batch-1 leaf evaluation -- timed on this host's cores on a bounded sample), then scalars: the same workload in the library's
other precisions (value_f16x2, roofline_f16x2, ...), whole_path_frac, config4 / config5 / dropin_config0 figures, the parity sample.
Everything else -- `kernels` (every kernel of a step against its own roof), the nested legs (`precisions`, `config4`, `config5`,
`other_driver`, `cross_game_dedup`, `eval_cache`, `all_layers_as_gemm`, `dropin_config0`, `parity_sample`), notes, `wall_breakdown` -- is
written to bench_detail.json next to this file.  In the timed region the network evaluates EVERY expansion (cross-game de-duplication off).
"""
import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this host driver (set before torch loads)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench_legs import (DTYPE_DETAIL, DTYPE_LABEL, FLOP_PER_EXPANSION, PRECISIONS, _finite, compact_line, config5_arena, conv3_tile_rows,  # noqa: E402,F401
                        conv_flop_per_leaf, cpu_baseline, roofline, run_secondary, tree_side)

# ---------------------------------------------------------------------------------------------------------------- launcher
BENCH_TIMEOUT_S = float(os.environ.get("OZ_BENCH_TIMEOUT", "520"))          # below the driver's own 600 s limit: a hang is reported by us, with its phase
COLLECTIVE_TIMEOUT_S = float(os.environ.get("OZ_BENCH_COLLECTIVE_TIMEOUT", "150"))

_PHASE = {"name": "start", "t": time.time(), "rank": int(os.environ.get("RANK", "0")), "done": False}


def phase(name):
    """rank side: the phase this rank is in (init / build / stagger / warmup / timed / gather / reduce / report), for the watchdog below
    and -- when bench.py launched the ranks itself -- for the parent (one small file per rank)"""
    _PHASE.update(name=name, t=time.time())
    d = os.environ.get("OZ_BENCH_PHASE_DIR")
    if d:
        try:
            with open(os.path.join(d, f"rank{_PHASE['rank']}"), "w") as f:
                f.write(f"{name} {time.time():.1f}\n")
        except OSError:
            pass


def start_watchdog(limit_s):
    """a rank that has not finished after limit_s says WHICH rank is stuck in WHICH phase and exits 124 -- also under
    torch.distributed.run, where no parent of ours watches (a rank stuck inside a collective has released the GIL)"""
    import threading
    t_start = time.time()

    def run():
        while not _PHASE["done"]:
            if time.time() - t_start > limit_s:
                print(f"bench.py: rank {_PHASE['rank']} did not finish within {limit_s:.0f} s: stuck in phase '{_PHASE['name']}' for "
                      f"{time.time() - _PHASE['t']:.0f} s", file=sys.stderr, flush=True)
                os._exit(124)
            time.sleep(0.5)
    threading.Thread(target=run, daemon=True).start()


def launch_ranks(args):
    """--gpus N > 1 without a launcher environment: start the N ranks as child processes of THIS process (which never
    initialises HIP), relay rank 0's stdout, fail if any rank fails or the realised world size is not N; on a failure or a
    timeout say which rank, and the phase every unfinished rank was in.  The rendezvous is a FILE store in a private temporary
    directory (no TCP port to lose a race for).  Replaces the role of WorkerManager fan-out in the reference (workers.py:168-184,298-303)."""
    import tempfile
    n = args.gpus
    tmp = tempfile.mkdtemp(prefix="oz_bench_")
    procs = []
    out_file = tempfile.TemporaryFile(mode="w+")               # rank 0's stdout (a file, so a long line can never block the rank)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   OZ_BENCH_INIT_FILE=os.path.join(tmp, "rendezvous"), OZ_BENCH_PHASE_DIR=tmp, OZ_BENCH_CHILD="1")
        env.pop("MASTER_PORT", None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out_file if r == 0 else subprocess.DEVNULL, stderr=None, text=True))

    def phases(ranks):
        out = []
        for r in ranks:
            try:
                name, t = open(os.path.join(tmp, f"rank{r}")).read().split()
                out.append(f"rank {r}: phase '{name}' for {time.time() - float(t):.0f} s")
            except (OSError, ValueError):
                out.append(f"rank {r}: no phase reported (died or hung before the first one)")
        return "; ".join(out)
    rc = 0
    try:
        pending = set(range(n))
        deadline = time.time() + BENCH_TIMEOUT_S + 10         # the ranks' own watchdogs fire first and name their phase
        while pending:
            for r in list(pending):
                code = procs[r].poll()
                if code is not None:
                    pending.discard(r)
                    if code != 0 and rc == 0:
                        rc = code
                        print(f"bench.py: rank {r} exited with code {code}" + (f"; still running: {phases(sorted(pending))}" if pending else ""),
                              file=sys.stderr)
            if rc != 0 or time.time() > deadline:
                if rc == 0:
                    rc = 124
                    print(f"bench.py: ranks did not finish within {BENCH_TIMEOUT_S + 10:.0f} s: {phases(sorted(pending))}", file=sys.stderr)
                break
            time.sleep(0.2)
    finally:
        for p in procs:                      # the exact children started above, nothing else
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    out_file.seek(0)
    out0 = out_file.read()
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if rc == 0:
        if len(lines) != 1:
            print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
            rc = 3
        else:
            got = json.loads(lines[0]).get("n_gpus")
            if got != n:
                print(f"bench.py: asked for {n} GPUs, the ranks realised a world of {got}", file=sys.stderr)
                rc = 2
    sys.stdout.write(out0)
    sys.stdout.flush()
    return rc


def main():
    t_proc = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--board", type=int, default=8)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--precision", default="bf16x3", choices=list(PRECISIONS),
                    help="arithmetic of the timed region (top-level value / dtype / roofline): bf16x3 (default) = every fp32 value carried EXACTLY as three bf16 "
                         "planes, six bf16 MFMA products per fp32 product, fp32 accumulate -- fp32-class results (the dropped cross terms are below fp32's own "
                         "rounding of a product) at 2.67x the fp32 matrix roof; f32 = exact fp32 matrix cores, literally what the reference computes in "
                         "(Net/NNet.py:85); f16x2 = f32 via 2 x fp16 split (22 of 24 significand bits).  Every other mode is measured after the timed region "
                         "and rides beside it (value_<mode>, roofline_<mode>)")
    ap.add_argument("--stagger-sims", type=int, default=-1,
                    help="simulations per move of the untimed stagger phase that spreads the slots over the plies of a game "
                         "(-1 = --sims: the staggered plies are ordinary self-play at full strength; 0 = no stagger: all games start "
                         "at ply 0 and none completes for ~60 move rounds)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic in this run")
    ap.add_argument("--no-dedup-compare", "--no-compare", dest="no_compare", action="store_true",
                    help="skip the secondary measurements (kernels, exact fp32, 6x6, de-duplication on, conv2 as a GEMM, drop-in, parity sample): profiling runs")
    ap.add_argument("--dedup", default="off", choices=["off", "on"],
                    help="cross-game leaf de-duplication in the timed region.  off (default for the headline): the network evaluates "
                         "every expansion -- no output is shared or cached; on: the library default (a board reached by several "
                         "games in the same step is evaluated once).  With off, the on-rate is measured afterwards and reported "
                         "next to it (N = 1 only)")
    ap.add_argument("--batch-cap", type=int, default=-1,
                    help="free-running driver: leaves per network batch; -1 = the cap at which conv3's grid is a whole number of rounds "
                         "(training.preferred_batch_cap: 3640 for 4096 8x8 games, none on 6x6), 0 = none")
    ap.add_argument("--driver", default="free", choices=["free", "lockstep"],
                    help="free: oz_selfplay_run_steps (every game runs on by itself, full leaf batches; identical records); "
                         "lockstep: oz_selfplay_run (one simulation per game per step, moves aligned)")
    ap.add_argument("--arena-plies", type=int, default=0, help="config5 leg: 0 (default) = every arena game played to the end (512 games x 800 sims per move "
                                                               "and agent); a positive value bounds the games to that many plies (quick looks)")
    ap.add_argument("--b3-tile", type=int, default=0, choices=[0, 128, 256], help="precision bf16x3: force k_gemm_b3's tile for conv3 / conv4 (0 = the launcher picks; bit-identical results -- an A/B switch)")
    ap.add_argument("--arena-precision", default="f16x2", choices=list(PRECISIONS), help="config5 leg: arithmetic of the two arena networks (reported as config5_precision)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the control flow)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses GPU 0")
    ap.add_argument("--no-c-abi-gather", action="store_true", help="N > 1: skip the post-line check of the C ABI's own RCCL exchange step")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))                            # parent: spawns the ranks, never initialises the GPU

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_device:
        local_rank = 0
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher realised WORLD_SIZE={world}; refusing to report a {world}-GPU number as {args.gpus}",
              file=sys.stderr)
        sys.exit(2)

    _PHASE["rank"] = rank
    if world > 1 or "OZ_BENCH_TIMEOUT" in os.environ:          # (a one-rank run cannot hang in a collective; long profiling runs stay possible)
        start_watchdog(BENCH_TIMEOUT_S)
    phase("init")
    try:
        run_rank(args, rank, world, local_rank, t_proc)
    except SystemExit:
        raise
    except BaseException:
        import traceback
        traceback.print_exc()
        print(f"bench.py: rank {rank} failed in phase '{_PHASE['name']}'", file=sys.stderr, flush=True)
        sys.stderr.flush()
        os._exit(1)                                            # (a rank blocked in a collective elsewhere is ended by its own watchdog / the launcher)
    _PHASE["done"] = True


def write_detail(out):
    """the full record of the run (everything the stdout line leaves out) as JSON next to bench.py; returns the path written (or None)"""
    import tempfile
    for d in (ROOT, tempfile.gettempdir()):
        path = os.path.join(d, "bench_detail.json")
        try:
            with open(path, "w") as f:
                json.dump(_finite(out), f, indent=1, allow_nan=False)
            return path
        except OSError:
            continue
    return None


def run_rank(args, rank, world, local_rank, t_proc):
    import datetime
    import torch
    import torch.distributed as dist
    from othellozero_amd import _lib
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.distributed import engine_records_tensor, gather_records
    from othellozero_amd.training import SelfPlayEngine

    _lib.require_gpu()
    if not args.same_device and torch.cuda.device_count() < world:
        print(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPUs visible", file=sys.stderr)
        sys.exit(2)
    _lib.check(_lib.load().oz_set_device(local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a collective (or the rendezvous) that does not complete in COLLECTIVE_TIMEOUT_S fails instead of hanging: torch's watchdog
        # aborts the communicator and the rank exits non-zero -- the launcher / this rank's watchdog then names the phase
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        kw = dict(rank=rank, world_size=world, timeout=datetime.timedelta(seconds=COLLECTIVE_TIMEOUT_S))
        if os.environ.get("OZ_BENCH_INIT_FILE"):               # ranks started by launch_ranks: a file store, no TCP port
            kw["init_method"] = "file://" + os.environ["OZ_BENCH_INIT_FILE"]
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)
        else:
            dist.init_process_group(args.backend, **kw)
        if dist.get_world_size() != args.gpus:
            print(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}", file=sys.stderr)
            sys.exit(2)

    phase("build")
    n, G = args.board, args.games
    stagger_sims = args.sims if args.stagger_sims < 0 else args.stagger_sims
    period = n * n - 4
    net = NNetWrapper((n, n), num_channels_1=args.channels, max_batch=G, seed=0, precision=args.precision)   # same weights on every rank
    if args.b3_tile:
        net.set_option(_lib.NET_OPT_B3_TILE, args.b3_tile)
    net.profile(1)          # HIP events around the dominant launch from the first launch on; the timed region is the difference of two readings

    from othellozero_amd.training import preferred_batch_cap

    def batch_cap(board, games, the_precision=None):
        """leaves per network batch of the free-running driver (0: none) for a network of `the_precision` (default: the timed region's)"""
        return preferred_batch_cap(board, games, args.channels, the_precision or args.precision) if args.batch_cap < 0 else args.batch_cap

    def make_engine(dedup, the_net=net, board=n, games=G, steps=args.steps, eval_cache=False):
        # (oz_selfplay_config.dedup / .batch_cap; the cap is used by the free-running driver only)
        return SelfPlayEngine(the_net, board, games, args.sims, 1.0, 1.0, 0.9, seed=1234, first_game_id=rank * games,
                              game_id_stride=world * games, q_mode=_lib.QMODE_F64, refill=True,
                              record_cap=int(games * (steps + args.warmup + board * board + 2) * 1.25),
                              dedup=dedup, batch_cap=batch_cap(board, games, getattr(the_net, "precision", None)), eval_cache=eval_cache)
    eng = make_engine(args.dedup == "on")
    cap_main = batch_cap(n, G) if args.driver == "free" else 0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # one bench step = `sims` network batches: in free-running mode (default) every batch holds exactly the batch cap (3640 leaves at
    # 4096 8x8 games: conv3 = 1024 tiles = 4.0 rounds of the chip); in lock step a step is a move round whose batches are ~91 % full and
    # whose conv3 grid pays 6 rounds for 5.5 (the `other_driver` leg: -4 % expansions/s)
    def advance(e, k, sync, driver=None):
        if (driver or args.driver) == "free":
            e.run_steps(k * args.sims, sync=sync)
        else:
            e.run(k, sync=sync)

    def measure(e, steps, the_net=None, driver=None):
        """warm-up + `steps` timed move rounds on engine e (single rank, secondary legs) -> (stats delta, seconds);
        with the_net: HIP events around its dominant launch during the timed rounds only"""
        advance(e, args.warmup, True, driver)
        e.sync()
        if the_net is not None:
            the_net.profile_kernels(reset=True)
            the_net.profile(1)
        a = e.stats()
        torch.cuda.synchronize()
        t = time.perf_counter()
        advance(e, steps, False, driver)
        e.sync()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t
        b = e.stats()
        if the_net is not None:
            the_net.profile(0)
        return {k: b[k] - a[k] for k in ("simulations", "expansions", "games_completed", "moves", "leaves_evaluated", "node_visits")}, dt_

    t_setup = time.perf_counter() - t_proc                      # imports, network + table build, engine allocation
    t_sec = time.perf_counter()
    phase("stagger")
    if stagger_sims >= 2:
        eng.stagger(stagger_sims)                               # untimed: slot g is (g * 60) / G plies into its first game
    t_stagger = time.perf_counter() - t_sec
    phase("warmup")
    advance(eng, args.warmup, True)
    eng.sync()
    if world > 1:           # warm-up of the exchange step too (communicator channels for both collectives), like the W untimed steps
        phase("gather-warmup")
        gather_records(torch.zeros((8, 48), dtype=torch.uint8, device=dev))
    dom0_ms, dom0_launches = net.profile_read()
    s0 = eng.stats()
    ev0 = eng.eval_time()
    phase("barrier")
    barrier()
    phase("timed")
    t0 = time.perf_counter()
    advance(eng, args.steps, False)
    eng.sync()
    # the path's only exchange step: the move records of the games that ended inside the timed region, pooled over the ranks
    phase("gather")
    t_g = time.perf_counter()
    local_records = engine_records_tensor(eng, dev)[s0["records"]:]
    pooled = gather_records(local_records)
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - t_g) * 1e3
    barrier()
    dt = time.perf_counter() - t0
    phase("reduce")
    s1 = eng.stats()
    ev1 = eng.eval_time()
    dom_all_ms, dom_all_launches = net.profile_read()           # the dominant launch: conv3 (conv2 is a gather-sum), or conv2
    dom_ms, dom_launches = dom_all_ms - dom0_ms, dom_all_launches - dom0_launches
    layer = net.profiled_layer()
    net.profile(0)

    d = {k: s1[k] - s0[k] for k in ("simulations", "expansions", "terminal_hits", "node_visits", "moves", "games_completed", "leaves_evaluated")}
    vec = torch.tensor([d["expansions"], d["simulations"], d["games_completed"], d["moves"], d["node_visits"]],
                       dtype=torch.float64, device=dev)
    tmax = torch.tensor([dt, gather_ms], dtype=torch.float64, device=dev)
    # per rank: records contributed to the pool, its OWN time of the timed steps (before the gather: t_g - t0), its own gather time, its own
    # expansions -- so that the first real multi-GPU run shows skew between the ranks, not just a sum and a maximum
    per_rank = torch.zeros((4, world), dtype=torch.float64, device=dev)
    per_rank[0, rank] = float(local_records.shape[0])
    per_rank[1, rank] = (t_g - t0) * 1e3 / args.steps
    per_rank[2, rank] = gather_ms
    per_rank[3, rank] = float(d["expansions"])
    if world > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
    dt, gather_ms = (float(x) for x in tmax.tolist())
    per_rank_records = [int(x) for x in per_rank[0].tolist()]
    per_rank_step_ms = [float(x) for x in per_rank[1].tolist()]
    per_rank_gather_ms = [float(x) for x in per_rank[2].tolist()]
    per_rank_expansions = [int(x) for x in per_rank[3].tolist()]
    if sum(per_rank_records) != int(pooled.shape[0]):                        # the all-gather lost or duplicated records: not a result
        print(f"bench.py: rank {rank}: pooled {int(pooled.shape[0])} records, the ranks contributed {per_rank_records} (sum {sum(per_rank_records)})",
              file=sys.stderr, flush=True)
        sys.exit(4)
    eng_for_c_abi = eng if world > 1 else None
    phase("report")
    exp_all, sims_all, games_all, moves_all, visits_all = (float(x) for x in vec.tolist())

    if rank == 0:
        # network work is counted per position actually evaluated (a board reached by several games in one step is evaluated once)
        flop_dom = d["leaves_evaluated"] * conv_flop_per_leaf(layer, n, args.channels)
        achieved = flop_dom / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        # FLOP the GPU executes per expansion: the reference network's, minus conv1 + conv2 when they run as table lookups
        flop_ref = FLOP_PER_EXPANSION.get(n, 0)
        flop_exec = flop_ref - (conv_flop_per_leaf(2, n, args.channels) + 2 * n * n * 18 * args.channels if layer == 3 else 0)
        nn_ms = ev1["ms"] - ev0["ms"]
        ply_now = eng.state()["ply"]
        out = {
            "metric": "mcts_node_expansions_per_sec", "value": exp_all / dt, "unit": "node-expansions/s",
            "n_gpus": world, "rccl_ranks": world if (world > 1 and args.backend == "nccl") else (0 if world > 1 else 1),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{G} concurrent {n}x{n} self-play games per GPU, {args.sims} sims/move, batched leaf eval "
                            "(BASELINE configs[1]); step = " + (f"{args.sims} network batches of up to {cap_main or G} leaves, games free-running "
                                                                  "(a game plays its move as soon as its simulations are complete, a leaf that finds no slot "
                                                                  "waits for the next batch; every game's records identical to lock step), "
                                                                  if args.driver == "free" else
                                                                  "one move round (100 lock-step simulations + one move per game), ") +
                            f"finished games refilled; leaf evaluator = the reference's OthelloNN ({args.channels} filters), random init seed 0",
                "workload_short": f"BASELINE configs[1]: {G} concurrent {n}x{n} self-play games per GPU, {args.sims} sims/move, batched leaf eval, OthelloNN "
                                  f"{args.channels} filters random init; step = " + (f"{args.sims} batches of <= {cap_main or G} leaves (free-running games, records = lock step)"
                                                                                     if args.driver == "free" else "one lock-step move round"),
                "precision": args.precision,
                "games_per_gpu": G, "sims_per_move": args.sims, "board": n,
                "q_mode": "float64 (NumPy 1.18.5 promotion)", "driver": args.driver, "batch_cap": cap_main or None,
                "parallelism": f"games sharded x{world}, all-gather of move records",
                "backend": (args.backend if world > 1 else None),
                "steady_state": (f"untimed oz_selfplay_stagger({stagger_sims}): slot g starts (g*{period})/{G} plies into its first game, those plies "
                                 f"played by searched self-play at {stagger_sims} sims/move; then {args.warmup} untimed warm-up rounds"
                                 if stagger_sims >= 2 else "none: all games start at ply 0 together"),
                "leaf_dedup": ("on: a board reached by several games in the same step is evaluated once (library default)" if args.dedup == "on" else
                               "off: the network evaluates every expansion, nothing is shared or cached between games "
                               "(the library default is on -- see cross_game_dedup for that rate)"),
            },
            "games_per_s": games_all / dt, "sims_per_s": sims_all / dt, "moves_per_s": moves_all / dt,
            "games_completed": int(games_all), "expansions": int(exp_all), "simulations": int(sims_all),
            "expansions_per_sim": exp_all / max(sims_all, 1), "node_visits_per_sim": visits_all / max(sims_all, 1),
            "pooled_records": int(pooled.shape[0]), "per_rank_records": per_rank_records, "gather_ms": gather_ms,
            "per_rank": {"ms_per_step": per_rank_step_ms, "ms_per_step_min": min(per_rank_step_ms), "ms_per_step_max": max(per_rank_step_ms),
                         "gather_ms": per_rank_gather_ms, "expansions": per_rank_expansions,
                         "note": "each rank's own wall time of the timed steps before the exchange step (ms_per_step above is the max over ranks of "
                                 "steps + gather + closing barrier), its own time in the exchange step, its own expansions"},
            "gather_note": "the path's one exchange step, inside the timed region: counts all-gather + one padded all-gather of 48-byte move "
                           "records copied device to device out of each engine's HBM buffer (max over ranks; pooled == sum of per_rank checked)",
            "slot_ply_spread_rank0": [int(ply_now.min()), int(ply_now.max())],
            "nn_forward_ms_total_rank0": nn_ms, "nn_fraction_of_wall_rank0": nn_ms * 1e-3 / dt,
            "leaves_evaluated_rank0": int(d["leaves_evaluated"]),
            "whole_net_tflops_rank0": d["leaves_evaluated"] * flop_exec / max(nn_ms * 1e-3, 1e-9) / 1e12,
            "flop_per_expansion": {"reference_network": flop_ref, "executed": flop_exec,
                                   "note": "executed < reference when conv1 + conv2 are evaluated as pattern-table lookups (exact refactoring, no GEMM)"},
            "roofline": dict(roofline(net.arithmetic(), layer, achieved, dom_ms, dom_launches, d["leaves_evaluated"], n, args.channels,
                                      conv3_rows=conv3_tile_rows(n, cap_main or G, args.channels)),
                             all_launches={"launches": int(dom_all_launches), "avg_launch_ms": dom_all_ms / max(dom_all_launches, 1),
                                           "leaves_per_launch": s1["leaves_evaluated"] / max(dom_all_launches, 1),
                                           "note": "every launch of this kernel since process start, incl. the untimed stagger and warm-up rounds "
                                                   "(partly filled batches): the population `rocprofv3 --stats` of the same command averages over; "
                                                   "avg_launch_ms above is the timed region only"}),
            # the tree / rules side is latency-bound integer work, ~1.3 KB of algorithmic HBM bytes per simulation
            "tree_side_hbm": tree_side(sims_all / dt),
        }
        out["dtype"] = DTYPE_LABEL[net.arithmetic()]
        out["dtype_detail"] = DTYPE_DETAIL[net.arithmetic()]
        out["roofline"]["leaves_per_launch"] = d["leaves_evaluated"] / max(dom_launches, 1)
        # SURVEY 8(d)'s whole-path figure: expansions/s x FLOP per expansion / the matrix peak of the arithmetic the timed region ran in.  On the FLOP the GPU
        # executes it is a fraction of the roof; on the reference network's FLOP it can exceed 1 by construction (conv1 + conv2 are exact table lookups)
        peak_tf = out["roofline"]["peak"]
        out["whole_path"] = {"frac_executed_flop": out["value"] / world * flop_exec / (peak_tf * 1e12),
                             "frac_reference_flop": out["value"] / world * flop_ref / (peak_tf * 1e12), "peak_tflops": peak_tf,
                             "per_gpu_expansions_per_s": out["value"] / world,
                             "note": "per GPU; frac_reference_flop counts conv1 + conv2 as GEMM FLOP although they run as exact pattern-table gathers: > 1 by construction "
                                     "when the tables are on (all_layers_as_gemm is the rate with every layer as a kernel / GEMM)"}
        wall = {"setup_s": round(t_setup, 2), "stagger_s": round(t_stagger, 2), "timed_region_s": round(dt, 2)}      # where this process's wall time goes
        out["wall_breakdown"] = wall
        # what THIS device's matrix pipes sustain right now, measured right after the timed region: a pure-MFMA loop (no LDS, no loads, no barriers) per
        # instruction the two precisions run on.  The GEMM kernels' rate moves with it from box to box (and with a power cap), so a reader can
        # tell a slow box from a regression: roofline.achieved x 3 / device_calibration.f16.tflops is the kernel's share of what the pipe gives HERE.
        try:
            cal = {k: _lib.mfma_rate(k, 50.0) for k in ("f16", "bf16", "f32")}
            out["device_calibration"] = {
                "f16": {"instruction": "v_mfma_f32_16x16x32_f16, operands with busy mantissas in [2^-3, 2^-2)", "sustained_tflops": cal["f16"]["tflops"],
                        "clock_ghz_at_back_to_back_issue": cal["f16"]["clock_ghz"], "ms": cal["f16"]["ms"], "nominal_peak_tflops": 2500.0},
                "bf16": {"instruction": "v_mfma_f32_16x16x32_bf16, operands with random 7-bit mantissas in [2^-3, 2^-2) (what the planes of bf16x3 hold)",
                         "sustained_tflops": cal["bf16"]["tflops"], "clock_ghz_at_back_to_back_issue": cal["bf16"]["clock_ghz"], "ms": cal["bf16"]["ms"],
                         "nominal_peak_tflops": 2500.0},
                "f32": {"instruction": "v_mfma_f32_32x32x2_f32", "sustained_tflops": cal["f32"]["tflops"],
                        "clock_ghz_at_back_to_back_issue": cal["f32"]["clock_ghz"], "ms": cal["f32"]["ms"], "nominal_peak_tflops": 157.3},
                "dominant_kernel_share_of_sustained": ({"f32": 1.0, "f16x2": 3.0, "bf16x3": 6.0}[net.arithmetic()] * achieved /
                                                       max(cal[{"f32": "f32", "f16x2": "f16", "bf16x3": "bf16"}[net.arithmetic()]]["tflops"], 1e-9)),
                "note": "oz_selftest_mfma_rate: one block per CU, one wave per SIMD, four independent accumulators back to back, ~50 ms each, rank 0, right "
                        "after the timed region; roofline.peak stays the nominal figure of MI355X_MICROARCH.md"}
        except Exception as e:                                   # noqa: BLE001 -- a diagnostic, never a reason to lose the line
            out["device_calibration"] = {"error": repr(e)}
        # ---- everything else of the line is measured AFTER the timed region, on fresh engines (bench_legs.py): kernels[] against their own roofs,
        # roofline.traffic from PMC child runs, the parity sample of both precisions, cross_game_dedup / eval_cache / other_driver /
        # all_layers_as_gemm, precisions (the other arithmetic modes: value_<mode> ...), config4 (6x6), config5 (BASELINE configs[4], whole games),
        # dropin_config0, cpu_baseline
        if world == 1:
            run_secondary(dict(args=args, out=out, wall=wall, world=world, net=net, eng=eng, make_engine=make_engine, measure=measure, advance=advance,
                               layer=layer, n=n, G=G, cap_main=cap_main, d=d, dom_launches=dom_launches, flop_ref=flop_ref, batch_cap=batch_cap))
            del eng
        # ---- output.  stdout: ONE line of <= 6 KB, contract keys first (bench_legs.compact_line) -- printed LAST, so that it is the tail of this
        # process's output whatever a reader keeps.  Everything else (kernels[], the nested legs, notes, wall_breakdown) goes to bench_detail.json next
        # to bench.py (a temporary directory if that is not writable) and, in a few lines, to stderr BEFORE the line.
        detail_path = write_detail(out)
        out["detail_file"] = detail_path
        text = compact_line(out)
        print(f"bench.py: detail -> {detail_path}; wall {json.dumps(wall)}", file=sys.stderr, flush=True)
        print(text, flush=True)
    if world > 1 and args.backend == "nccl" and not args.no_c_abi_gather:
        # AFTER the line is out (nothing here can cost the measurement): the same exchange step through the C ABI's own RCCL communicator
        # (oz_comm_* / oz_selfplay_gather_records), checked against the torch.distributed pool; outcome on stderr.  Guarded by a
        # timeout: a rank that cannot finish it in 60 s says so and the run still ends with exit code 0.
        import threading
        phase("c-abi-gather")
        result = {}

        def c_abi():
            try:
                from othellozero_amd.distributed import Comm, torch_share
                _lib.check(_lib.load().oz_set_device(local_rank))     # a new host thread starts on device 0: select this rank's GPU
                torch.cuda.set_device(local_rank)
                comm = Comm(rank, world, torch_share(dev))
                t_c = time.perf_counter()
                rec, per = comm.gather_records(eng_for_c_abi, first_record=s0["records"])
                result.update(ms=(time.perf_counter() - t_c) * 1e3, records=int(rec.size), per_rank=[int(x) for x in per],
                              same=bool(rec.tobytes() == pooled.cpu().numpy().tobytes()))
                comm.close()
            except BaseException as e:                          # noqa: BLE001 -- report, never fail the run
                result["error"] = repr(e)
        th = threading.Thread(target=c_abi, daemon=True)
        th.start()
        th.join(60)
        if th.is_alive():
            print(f"bench.py: rank {rank}: c-abi gather over RCCL did not finish in 60 s (skipped; the line above stands)", file=sys.stderr, flush=True)
            os._exit(0)
        if rank == 0:
            print(f"bench.py: c-abi gather over RCCL ({world} ranks, oz_selfplay_gather_records): {json.dumps(result)}", file=sys.stderr, flush=True)
    if world > 1:
        phase("teardown")
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
