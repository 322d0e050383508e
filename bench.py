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

Prints ONE JSON line on rank 0: value = node expansions per second over all GPUs (the BASELINE metric; the same line
carries games/s and sims/s), `roofline` for the dominant kernel (HIP events on the launch stream, in the timed region),
and -- at N = 1 -- `kernels` (every kernel of a step against its own roof), `exact_fp32` (the same workload in exact fp32
arithmetic), `config4` (6x6 boards), `other_driver` (the lock-step driver), `cross_game_dedup`, `all_layers_as_gemm`, `dropin_config0` (configs[0] through the
reference's Python surface), `parity_sample_max_err` and `cpu_baseline` (the CPU oracle -- the reference algorithm with
batch-1 leaf evaluation -- timed on this host's cores on a bounded sample).  In the timed region the network evaluates
EVERY expansion (cross-game de-duplication off).
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

FLOP_PER_EXPANSION = {8: 566428672, 6: 270185472}          # SURVEY.md 8(d), whole OthelloNN forward
PEAK_F32_MATRIX_TFLOPS = 157.3                             # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_F16_MATRIX_TFLOPS = 2500.0                            # MI355X_MICROARCH.md: ~2.5 PF dense fp16/bf16 MFMA
PEAK_HBM_GBPS = 8000.0                                     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_L2_GBPS = 34500.0                                     # MI355X_MICROARCH.md: L2 aggregate ~34.5 TB/s (8 XCDs x 4 MiB)
TREE_BYTES_PER_SIM = 1300                                  # algorithmic HBM bytes per simulation on the tree side


def conv_flop_per_leaf(layer, n, C):
    """implicit-GEMM FLOP per leaf of conv2 (layer 2: n x n outputs), conv3 ((n-2)^2 outputs), conv4 ((n-4)^2 outputs)"""
    px = {2: n * n, 3: (n - 2) * (n - 2), 4: (n - 4) * (n - 4)}[layer]
    return 2 * px * (9 * C) * C                            # 8x8: conv2 301 989 888, conv3 169 869 312, conv4 75 497 472 FLOP


def conv3_tile_rows(n, capacity, C):
    """the row-tile height oz_net.hip picks for conv3 at a launch capacity of `capacity` leaves: the one that pays fewer tile rows
    (rounds of 256 CUs x tile height); 6x6 boards always use 256"""
    def cost(bm):
        blocks = -(-capacity * (n - 2) ** 2 // bm) * (C // 256)
        return -(-blocks // 256) * bm
    return 256 if n == 6 or cost(256) < cost(192) else 192


def roofline(precision, layer, achieved, layer_ms, launches, expansions, n, channels, conv3_rows=192):
    """Dominant kernel.  precision f32 / f16x2 with the pattern tables (default): conv1 + conv2 run as a table gather-sum
    (k_conv2_lut), so the dominant launch is the conv3 implicit GEMM; without the tables it is the conv2 implicit GEMM.
    `achieved` is ALGORITHMIC fp32 TFLOP/s (2*M*K*N per launch / HIP-event time on the launch stream).
    f32: v_mfma_f32_32x32x2_f32, peak 157.3.  f16x2: every fp32 product costs 3 fp16 MFMA products, so the matrix pipe
    executes 3x `achieved`; both fractions are reported against the 2.5 PFLOP/s dense fp16 peak."""
    hbm, src = None, None
    try:      # HBM-side bytes per leaf from the committed PMC profile of this kernel (profiles/), scaled per launch
        rel = os.path.join("profiles", f"conv{layer}_traffic_{precision}.json")
        alt = os.path.join("profiles", f"conv{layer}_traffic_{precision}_lockstep_192row_tile.json")      # the counters of the 192-row tile (lock-step batches)
        if layer == 3 and conv3_rows == 192 and os.path.exists(os.path.join(ROOT, alt)):
            rel = alt
        tj = json.load(open(os.path.join(ROOT, rel)))
        hbm = tj["hbm_bytes_per_leaf"] * expansions / max(launches, 1)
        src = f"{rel} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this kernel, separate passes, gfx950 x2 read correction) x leaves per launch of this run; not re-measured in this run"
    except Exception:
        pass
    r = {"bound": "mfma", "achieved": achieved, "unit": "TFLOP/s", "traffic": hbm, "traffic_source": src, "launches": int(launches),
         "avg_launch_ms": layer_ms / max(launches, 1), "flop_per_leaf": conv_flop_per_leaf(layer, n, channels),
         "algorithmic_bytes_per_launch": None}
    px_in, px_out = {2: (n * n, n * n), 3: (n * n, (n - 2) ** 2), 4: ((n - 2) ** 2, (n - 4) ** 2)}[layer]
    r["algorithmic_bytes_per_launch"] = (expansions / max(launches, 1)) * (px_in + px_out) * channels * 4 + 9 * channels * channels * 4
    if precision == "f32":
        r.update(kernel=("k_gemm_f32 (conv3: 3x3 valid, 512->512, 8x8 -> 6x6, implicit GEMM, v_mfma_f32_32x32x2_f32); conv1 + conv2 = k_conv2_lut_f32 table gather-sum"
                         if layer == 3 else "k_gemm_f32 (conv2: 3x3 same, 512->512, implicit GEMM, v_mfma_f32_32x32x2_f32)"),
                 peak=PEAK_F32_MATRIX_TFLOPS, frac=achieved / PEAK_F32_MATRIX_TFLOPS)
    else:
        tail = "implicit GEMM, f32 as 2xfp16 split, v_mfma_f32_16x16x32_f16, 4-phase ping-pong loop"
        if layer == 3:
            cfg = "H2BigPP" if conv3_rows == 256 else "H2MidPP"
            kernel = (f"k_gemm_h2<{cfg}, 3> (conv3: 3x3 valid, 512->512, {n}x{n} -> {n - 2}x{n - 2}, {conv3_rows} x 256 tiles, {tail}); "
                      "conv1 + conv2 = k_conv2_lut table gather-sum")
        else:
            kernel = f"k_gemm_h2<H2BigPP, 2> (conv2: 3x3 same, 512->512, {tail}; oz_net_set_tables(0): conv1 as a kernel)"
        r.update(kernel=kernel, peak=PEAK_F16_MATRIX_TFLOPS, frac=achieved / PEAK_F16_MATRIX_TFLOPS,
                 mfma_products_per_fp32_product=3, matrix_pipe_tflops=3 * achieved,
                 matrix_pipe_frac=3 * achieved / PEAK_F16_MATRIX_TFLOPS,
                 vs_fp32_matrix_peak=achieved / PEAK_F32_MATRIX_TFLOPS)
    return r


def measured_tree_bytes():
    """HBM-side bytes per simulation of the tree kernels from the committed PMC profile (profiles/tree_traffic.json: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE over the 100 launches of one timed step, tools/profile_round.sh) -> (bytes with the gfx950 x2 read
    correction, source label); falls back to SURVEY 8(d)'s algorithmic estimate when the file is absent"""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "tree_traffic.json")))
        t = tj["tree_side_bytes_per_sim"]
        return float(t["total_with_x2_fetch"]), ("profiles/tree_traffic.json: measured FETCH_SIZE x2 + WRITE_SIZE per simulation (PMC, one timed step of a "
                                                   "profiling run of this command; not re-measured in this run)")
    except Exception:
        return float(TREE_BYTES_PER_SIM), "SURVEY.md 8(d) algorithmic estimate (no measured profile found)"


def kernel_table(net_k, tree_k, rounds, leaves, sims_done, n, C, precision, tables, driver="lockstep"):
    """every kernel of a move round against its own roof.  net_k / tree_k: {name: (ms_total, launches)} over `rounds` move
    rounds that evaluated `leaves` positions and ran `sims_done` simulations."""
    peak_mm = PEAK_F32_MATRIX_TFLOPS if precision == "f32" else PEAK_F16_MATRIX_TFLOPS
    F = (n - 4) * (n - 4) * C
    flop = {"conv2": conv_flop_per_leaf(2, n, C), "conv3": conv_flop_per_leaf(3, n, C), "conv4": conv_flop_per_leaf(4, n, C),
            "fc1": 2 * F * 1024, "fc2": 2 * 1024 * 512}
    out = []
    for name, (ms, cnt) in net_k.items():
        if cnt == 0:
            continue
        row = {"name": name, "ms_per_step": ms / rounds, "launches_per_step": cnt / rounds}
        sec = ms * 1e-3
        if name == "conv2" and tables:
            # the table gather: bound by the bytes that miss L2 (the 363 MB table cannot live in the 256 MB Infinity Cache, so they are HBM
            # bytes).  `achieved` = HBM-side bytes per leaf from the committed PMC summary of this command (FETCH_SIZE x2 + WRITE_SIZE of
            # the gather's launches in the timed step, profiles/r3_*_bench_pmc_by_shape.csv) x the leaves of this run / this run's time
            per_leaf, src = None, None
            try:
                import csv
                rel = os.path.join("profiles", f"r3_{precision}_bench_pmc_by_shape.csv")
                rows = [r for r in csv.DictReader(l for l in open(os.path.join(ROOT, rel)) if not l.startswith("#")) if "k_conv2_lut" in r["kernel"]]
                fetch = max(float(r["avg_per_launch"]) for r in rows if r["counter"] == "FETCH_SIZE")
                write = max(float(r["avg_per_launch"]) for r in rows if r["counter"] == "WRITE_SIZE")
                per_leaf = (fetch * 2048 + write * 1024) / 3640.0 * (n * n / 64.0) * (C / 512.0)
                src = f"{rel}: FETCH_SIZE x2 + WRITE_SIZE of the gather per 3640-leaf launch (PMC passes of this command); not re-measured in this run"
            except Exception:
                per_leaf = 0.58e9 / 3640 * (n * n / 64.0) * (C / 512.0) + n * n * C * 4
                src = "0.58 GB read (PMC, round 3) + the compulsory output row per pixel per 3640-leaf launch; not re-measured in this run"
            byts = leaves * per_leaf
            row.update(kernel=("k_conv2_lut_xcd" if C == 512 else "k_conv2_lut") + (" (fp32 rows)" if precision == "f32" else " (h2 rows)"), bound="hbm",
                       achieved=byts / sec / 1e9, peak=PEAK_HBM_GBPS, unit="GB/s", bytes_source=src)
        elif name in flop:
            kern = "k_gemm_f32" if precision == "f32" else "k_gemm_h2"
            row.update(kernel=f"{kern} ({name})", bound="mfma", achieved=leaves * flop[name] / sec / 1e12, peak=peak_mm, unit="TFLOP/s")
        elif name == "input":
            byts = leaves * (16 + n * n * 2) if tables else leaves * (16 + n * n * C * 4)
            row.update(kernel="k_lut_ids" if tables else "k_conv1", bound="hbm", achieved=byts / sec / 1e9, peak=PEAK_HBM_GBPS, unit="GB/s")
        elif name == "heads":
            byts = leaves * (512 * 4 + (n * n + 1) * 4)
            row.update(kernel="k_heads", bound="hbm", achieved=byts / sec / 1e9, peak=PEAK_HBM_GBPS, unit="GB/s")
        row["frac"] = row["achieved"] / row["peak"]
        out.append(row)
    tree_ms = sum(ms for name, (ms, cnt) in tree_k.items() if name != "network")
    tree_bytes, tree_src = measured_tree_bytes()
    for name, (ms, cnt) in tree_k.items():
        if name == "network" or cnt == 0:
            continue
        label = ({"select": "k_advance (first batch of a call) / k_backup_advance (expand + backup of the previous batch's leaves fused with every game's "
                            "advance: its move when due, network-free simulations, the descent to its next leaf)",
                  "compact": "k_compact (slot order rotating under the batch cap)", "expand_backup": "k_expand_backup (closing one of a call)",
                  "roots_move": "k_sp_roots + k_sp_move"} if driver == "free" else
                 {"select": "k_select (first simulation of a round) / k_backup_select (expand + backup of simulation s-1 fused with the descent of s)",
                  "compact": "k_compact", "expand_backup": "k_expand_backup (closing one of a round)", "roots_move": "k_sp_roots + k_sp_move"})
        out.append({"name": name, "kernel": label[name],
                    "ms_per_step": ms / rounds, "launches_per_step": cnt / rounds, "bound": "hbm (latency-bound integer work)",
                    # the tree side as a whole: measured HBM-side bytes per simulation (PMC profile) over the time of the tree kernels together
                    "achieved": sims_done * tree_bytes / (tree_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                    "frac": sims_done * tree_bytes / (tree_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                    "bytes_per_sim": tree_bytes, "bytes_source": tree_src,
                    "note": "achieved / frac are those of the tree kernels together (latency-bound pointer chasing: one dependent HBM round trip per tree level)"})
    return out


def tree_side(sims_per_s):
    """the tree / rules side is latency-bound integer work; ~1.3 KB of algorithmic HBM bytes per simulation.
    The measured bytes (PMC FETCH_SIZE / WRITE_SIZE of k_select / k_expand_backup / k_compact over one timed move round) come
    from the committed profile, labelled as such."""
    tree_bytes, tree_src = measured_tree_bytes()
    r = {"bytes_per_sim": tree_bytes, "bytes_per_sim_source": tree_src, "algorithmic_bytes_per_sim": TREE_BYTES_PER_SIM,
         "achieved_GBps": sims_per_s * tree_bytes / 1e9, "peak_GBps": PEAK_HBM_GBPS,
         "frac": sims_per_s * tree_bytes / (PEAK_HBM_GBPS * 1e9), "note": "not the binding roof; reported per SURVEY 8(d)"}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "tree_traffic.json")))
        t = tj["tree_side_bytes_per_sim"]
        r["measured"] = {"fetch_bytes_per_sim_raw": t["fetch_raw"], "fetch_bytes_per_sim_x2": t["fetch_x2"], "write_bytes_per_sim": t["write"],
                         "descent_kernel_fetch_bytes_per_sim_x2": (tj["kernels"].get("k_backup_advance") or tj["kernels"].get("k_backup_select") or tj["kernels"]["k_select"])["fetch_bytes_per_sim_x2"],
                         "source": "profiles/tree_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, the 100 launches of one timed move round; "
                                   "x2 = the gfx950 rule for 16-byte-per-lane reads, an upper bound for this access mix); not re-measured in this run"}
    except Exception:
        pass
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
        "sims_per_s": plies * sims / t, "games_per_s": games / t,
        "value_1_thread": ep1["stats"]["expansions"] / t1, "sample_1_thread": f"first 2 plies of game 0 ({t1:.1f} s)",
        "cpu_model": model, "host_cpus": os.cpu_count(),
    }


def cpu_baseline_config(n, channels, sims, budget_s, threads):
    """the same port on another config (configs 1, 2 and 4): whole sequential games within budget_s"""
    import oracle
    from othellozero_amd.weights import init_weights
    w = init_weights(n, seed=0, channels=channels)
    net = oracle.CNet(w, n, channels=channels, nthreads=threads)
    t, exp, plies, games = 0.0, 0, 0, 0
    while t < budget_s and games < 64:
        m = oracle.Mcts(n, 1.0, oracle.QMODE_F64, evaluator=net.evaluator())
        t0 = time.perf_counter()
        ep = m.episode(sims, 1.0, 0.9, 1234, games)
        t += time.perf_counter() - t0
        exp, plies, games = exp + ep["stats"]["expansions"], plies + ep["n_moves"], games + 1
    return {"value": exp / t, "unit": "node-expansions/s", "cores": threads, "kind": "port", "games_per_s": games / t,
            "sample": f"{games} sequential {n}x{n} game(s) at {sims} sims/move ({exp} expansions, {t:.1f} s), batch-1 leaf eval, OpenMP x{threads}"}


def dropin_config0(channels, precision):
    """BASELINE configs[0] through the reference's own Python surface (training.execute_episode -> OthelloMCTS.simulate ->
    NNetWrapper.predict, one position per call: training.py:26-72) on the GPU library, and the CPU port of the same
    episode (same RNG streams, so the same game when the two networks agree on every arg-max) beside it."""
    import random
    import numpy as np
    import oracle
    from othellozero_amd import training
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.weights import init_weights
    n, sims = 8, 25
    # the network a drop-in caller gets: NNetWrapper's defaults (max_batch 1 = the library's latency path; exact fp32 arithmetic,
    # which is also the faster one for a single position: its layers of <= 64 rows run as weight streams)
    net = NNetWrapper((n, n), num_channels_1=channels, max_batch=1, seed=0)
    precision = net.precision
    random.seed(0); np.random.seed(0)
    training.execute_episode(n, net, 1, 2, 1, 0.9)                 # untimed: first-call allocations, code object load
    random.seed(1); np.random.seed(1)
    t0 = time.perf_counter()
    ex = training.execute_episode(n, net, 1, sims, 1, 0.9)
    t_gpu = time.perf_counter() - t0
    moves = len(ex) // 8
    w = init_weights(n, seed=0, channels=channels)
    threads = min(oracle.lib().orc_nn_max_threads(), 16)
    cnet = oracle.CNet(w, n, channels=channels, nthreads=threads)
    m = oracle.Mcts(n, 1.0, oracle.QMODE_F64, evaluator=cnet.evaluator())
    t0 = time.perf_counter()
    ep = m.episode(sims, 1.0, 0.9, 1234, 0)
    t_cpu = time.perf_counter() - t0
    return {"workload": "BASELINE configs[0]: one 8x8 self-play game, 25 sims/move, random-init OthelloNN, through the reference's "
                        "execute_episode / OthelloMCTS / NNetWrapper.predict surface (one position per call)",
            "gpu_dropin": {"seconds": t_gpu, "moves": moves, "sims": moves * sims, "sims_per_s": moves * sims / t_gpu, "games_per_s": 1.0 / t_gpu,
                           "precision": precision,
                           "path": "Python drop-in over the C ABI: OthelloMCTS.simulate -> select + NNetWrapper.predict (latency kernels: one position per call) + backup"},
            "cpu_port": {"seconds": t_cpu, "moves": int(ep["n_moves"]), "sims": int(ep["n_moves"]) * sims, "expansions": int(ep["stats"]["expansions"]),
                         "sims_per_s": int(ep["n_moves"]) * sims / t_cpu, "games_per_s": 1.0 / t_cpu, "cores": threads, "kind": "port"}}


def parity_sample(net, eng, n, channels, call_size, check=384):
    """the checker, after the timed region: ONE network call of `call_size` positions -- the size of the timed region's batches (the
    batch cap: conv3 on the tile the roofline row is about), positions the engine holds right now, evaluated by the network object
    the timed region used (same kernels, same max_batch) -- against the float64 oracle on `check` rows spread over the call (every
    output row is an independent accumulation), and bit-compared with a second, shorter call (another conv3 tile, same sums)"""
    import numpy as np
    from oracle import nn_numpy
    from othellozero_amd.weights import init_weights
    st = eng.state()
    idx = np.linspace(0, st["black"].size - 1, call_size).astype(np.int64)
    own = np.where(st["player"][idx] == 1, st["black"][idx], st["white"][idx])
    opp = np.where(st["player"][idx] == 1, st["white"][idx], st["black"][idx])
    pi, v = net.predict_batch(own, opp)                                       # one call: call_size <= max_batch
    tile = net.conv3_tile_rows()
    rows = np.linspace(0, call_size - 1, min(check, call_size)).astype(np.int64)
    pi64, v64 = nn_numpy.forward_chunked(init_weights(n, seed=0, channels=channels), own[rows], opp[rows], n, chunk=128)
    short = min(256, call_size)
    p2, v2 = net.predict_batch(own[:short], opp[:short])
    return {"max_abs_err_pi": float(np.abs(pi.reshape(call_size, -1)[rows] - pi64).max()), "max_abs_err_v": float(np.abs(v[rows] - v64).max()),
            "positions_in_the_call": int(call_size), "conv3_tile_rows_of_the_call": int(tile), "rows_checked_vs_float64": int(rows.size),
            "bit_identical_to_a_shorter_call": bool(np.array_equal(p2, pi[:short]) and np.array_equal(v2, v[:short])),
            "shorter_call": {"positions": int(short), "conv3_tile_rows": int(net.conv3_tile_rows())},
            "plies_sampled": [int(st["ply"][idx].min()), int(st["ply"][idx].max())], "tolerance": 1e-5,
            "checker": "oracle/nn_numpy.py (float64 restatement of Net/OthelloNN.py:42-56)"}


def live_traffic(args, layer, grid_leaves, timeout_s=150):
    """roofline.traffic measured IN this run: two child processes -- `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE`,
    separate passes as MI355X_MICROARCH.md prescribes, never combined with another trace domain -- each running this bench with one timed
    step (100 batches of the same capped size; secondary legs off), and the dominant kernel's counters averaged over the launches of
    that step.  FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads (x2).
    Returns (bytes per launch, description) or (None, reason).  The children are ordinary child processes of this one."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="oz_bench_pmc_")
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out_dir, "-o", "run", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--stagger-sims", "8", "--no-compare", "--no-cpu-baseline",
                   "--precision", args.precision, "--games", str(args.games), "--sims", str(args.sims), "--board", str(args.board),
                   "--channels", str(args.channels), "--driver", args.driver, "--batch-cap", str(args.batch_cap), "--dedup", args.dedup]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=env)
            files = glob.glob(os.path.join(out_dir, "**", "*_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}"
            # the dominant launch of the timed step: the layer-th GEMM after each k_lut_ids, in the last `sims` forwards (launch order, as
            # tools/summarize_prof.py labels them)
            rows = sorted(csv.DictReader(open(files[0])), key=lambda x: int(x["Dispatch_Id"]))
            picked, order, nth = [], -1, 0
            want = {3: 1, 2: 1}[layer]                           # conv3 is the first GEMM of a forward that uses the tables; conv2 the first without
            for x in rows:
                k = x["Kernel_Name"]
                if "k_lut_ids" in k or "k_conv1" in k:
                    nth = 0
                elif "k_gemm" in k:
                    nth += 1
                    if nth == want and x["Counter_Name"] == counter:
                        picked.append(float(x["Counter_Value"]))
            picked = picked[-args.sims:]
            if not picked:
                return None, f"no dominant-kernel rows in the {counter} pass"
            vals[counter] = sum(picked) / len(picked)
    except Exception as e:                                       # noqa: BLE001 -- a measurement aid, never a reason to fail the bench
        return None, f"live PMC pass failed: {e!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    byts = vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024
    return byts, (f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two separate child runs of bench.py --steps 1 --stagger-sims 8 "
                  f"--no-compare), average over the {args.sims} launches of the timed step at {grid_leaves} leaves per launch; FETCH_SIZE x2 (gfx950) "
                  f"= {vals['FETCH_SIZE'] * 2048 / 1e9:.3f} GB read + {vals['WRITE_SIZE'] * 1024 / 1e9:.3f} GB written")


def config5_arena(channels, precision, plies, games=512, sims=800, sample=2):
    """BASELINE configs[4]: arena evaluation (agents.py:44-84, duel_between_agents) -- `games` parallel 8x8 games, `sims` simulations per
    move per agent, two REAL networks (seeds 0 / 1: best vs candidate), deterministic play (temperature 0, RNG_TIE stream), bounded
    to `plies` plies per game so that the leg fits the run's time budget; `sample` games are replayed by the CPU oracle's arena fed
    with the GPU networks' own (pi, v) -> sample_mismatches must be 0."""
    import numpy as np
    import oracle
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.agents import arena_batch
    n = 8
    nets = [NNetWrapper((n, n), num_channels_1=channels, max_batch=games, seed=sd, precision=precision) for sd in (0, 1)]
    arena_batch(nets[0], nets[1], n, games, 8, 1.0, seed=11, first_game_id=0, q_mode=1, max_rounds=2)      # untimed: allocation, code load
    # headline of the leg: every expansion evaluated by itself (cross-game de-duplication off, as in the self-play headline); then the
    # library default (on) -- arena games start from ONE opening and play deterministically, so most expansions of a step share a board
    t0 = time.perf_counter()
    r = arena_batch(nets[0], nets[1], n, games, sims, 1.0, seed=11, first_game_id=0, q_mode=1, max_rounds=plies, dedup=False)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    rd = arena_batch(nets[0], nets[1], n, games, sims, 1.0, seed=11, first_game_id=0, q_mode=1, max_rounds=plies, dedup=True)
    dtd = time.perf_counter() - t0
    live = np.arange(128)[None, :] < r["n_moves"][:, None]                     # the moves actually played
    same = bool(np.array_equal(r["n_moves"], rd["n_moves"]) and np.array_equal(r["actions"][live], rd["actions"][live])
                and np.array_equal(r["players"][live], rd["players"][live])
                and np.array_equal(r["final_black"], rd["final_black"]) and np.array_equal(r["final_white"], rd["final_white"]))
    moves = int(r["n_moves"].sum())
    st = r["stats_black"] + r["stats_white"]
    caches = [{}, {}]

    def evaluator(k):
        def ev(own, opp, nn):
            if (own, opp) not in caches[k]:
                p, v = nets[k].predict_batch([own], [opp])
                caches[k][(own, opp)] = (p[0].ravel(), float(v[0]))
            return caches[k][(own, opp)]
        return ev
    t1 = time.perf_counter()
    bad = 0
    for gi in range(sample):
        o = oracle.arena(oracle.Mcts(n, 1.0, 1, evaluator=evaluator(0)), oracle.Mcts(n, 1.0, 1, evaluator=evaluator(1)), sims, 11, gi, max_plies=plies)
        k = o["n_moves"]
        ok = (int(r["n_moves"][gi]) == k and np.array_equal(r["actions"][gi][:k], o["action"]) and np.array_equal(r["players"][gi][:k], o["player"])
              and (int(r["final_black"][gi]), int(r["final_white"][gi])) == (int(o["final_black"]), int(o["final_white"])))
        bad += 0 if ok else 1
    return {"workload": f"BASELINE configs[4]: {games} parallel 8x8 arena games, {sims} sims/move per agent, two {channels}-filter OthelloNN "
                        f"(random init, seeds 0 / 1), temperature 0, deterministic; first {plies} plies of every game (bounded for the run's time budget)",
            "seconds": dt, "plies_per_game": plies, "moves": moves, "simulations": int(st[0]), "expansions": int(st[2]),
            "sims_per_s": float(st[0]) / dt, "value": float(st[2]) / dt, "unit": "node-expansions/s", "moves_per_s": moves / dt,
            "games_per_s_extrapolated_60_plies": games / (dt * 60.0 / plies), "precision": precision,
            "leaves_evaluated": int(r["leaves_evaluated"]),
            "with_cross_game_dedup": {"seconds": dtd, "sims_per_s": float(rd["stats_black"][0] + rd["stats_white"][0]) / dtd,
                                      "leaves_evaluated": int(rd["leaves_evaluated"]), "identical_games": same,
                                      "note": "library default: a board several games reach in one step is evaluated once"},
            "sample_games_replayed_by_oracle": sample, "sample_mismatches": bad, "oracle_replay_s": round(time.perf_counter() - t1, 2),
            "note": "oz_arena_run_rounds: BLACK movers search in net A's trees, WHITE movers in net B's, one searched ply per game and round; "
                    "the oracle's arena (agents.py restated, oracle/oz_oracle.c orc_arena_plies) is fed the GPU networks' own (pi, v), so "
                    "actions, movers and boards must agree bit for bit"}


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
    ap.add_argument("--precision", default="f16x2", choices=["f32", "f16x2"],
                    help="conv arithmetic: exact fp32 matrix cores, or f32 via 2 x fp16 split (same 1e-5 parity tolerance)")
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
    ap.add_argument("--arena-plies", type=int, default=6, help="config5 leg: plies per arena game (512 games x 800 sims per move and agent)")
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
    net.profile(1)          # HIP events around the dominant launch from the first launch on; the timed region is the difference of two readings

    from othellozero_amd.training import preferred_batch_cap

    def batch_cap(board, games):
        """leaves per network batch of the free-running driver (0: none)"""
        return preferred_batch_cap(board, games, args.channels) if args.batch_cap < 0 else args.batch_cap

    def make_engine(dedup, the_net=net, board=n, games=G, steps=args.steps, eval_cache=False):
        # (oz_selfplay_config.dedup / .batch_cap; the cap is used by the free-running driver only)
        return SelfPlayEngine(the_net, board, games, args.sims, 1.0, 1.0, 0.9, seed=1234, first_game_id=rank * games,
                              game_id_stride=world * games, q_mode=_lib.QMODE_F64, refill=True,
                              record_cap=int(games * (steps + args.warmup + board * board + 2) * 1.25),
                              dedup=dedup, batch_cap=batch_cap(board, games), eval_cache=eval_cache)
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
    per_rank = torch.zeros(world, dtype=torch.float64, device=dev)           # records every rank contributed to the pool
    per_rank[rank] = float(local_records.shape[0])
    if world > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
    dt, gather_ms = (float(x) for x in tmax.tolist())
    per_rank_records = [int(x) for x in per_rank.tolist()]
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
            "gather_note": "the path's one exchange step, inside the timed region: counts all-gather + one padded all-gather of 48-byte move "
                           "records copied device to device out of each engine's HBM buffer (max over ranks; pooled == sum of per_rank checked)",
            "slot_ply_spread_rank0": [int(ply_now.min()), int(ply_now.max())],
            "nn_forward_ms_total_rank0": nn_ms, "nn_fraction_of_wall_rank0": nn_ms * 1e-3 / dt,
            "leaves_evaluated_rank0": int(d["leaves_evaluated"]),
            "whole_net_tflops_rank0": d["leaves_evaluated"] * flop_exec / max(nn_ms * 1e-3, 1e-9) / 1e12,
            "flop_per_expansion": {"reference_network": flop_ref, "executed": flop_exec,
                                   "note": "executed < reference when conv1 + conv2 are evaluated as pattern-table lookups (exact refactoring, no GEMM)"},
            "roofline": dict(roofline(args.precision, layer, achieved, dom_ms, dom_launches, d["leaves_evaluated"], n, args.channels,
                                      conv3_rows=conv3_tile_rows(n, cap_main or G, args.channels)),
                             all_launches={"launches": int(dom_all_launches), "avg_launch_ms": dom_all_ms / max(dom_all_launches, 1),
                                           "leaves_per_launch": s1["leaves_evaluated"] / max(dom_all_launches, 1),
                                           "note": "every launch of this kernel since process start, incl. the untimed stagger and warm-up rounds "
                                                   "(partly filled batches): the population `rocprofv3 --stats` of the same command averages over; "
                                                   "avg_launch_ms above is the timed region only"}),
            # the tree / rules side is latency-bound integer work, ~1.3 KB of algorithmic HBM bytes per simulation
            "tree_side_hbm": tree_side(sims_all / dt),
        }
        out["dtype"] = "f32" if args.precision == "f32" else "f32 (2xf16 split)"
        out["dtype_detail"] = ("fp32 operands and accumulators on v_mfma_f32_32x32x2_f32" if args.precision == "f32" else
                               "fp32 values carried as two fp16 planes, 3 fp16 MFMA products per fp32 product, fp32 accumulate; pi, v within 1e-5 of float64")
        wall = {"setup_s": round(t_setup, 2), "stagger_s": round(t_stagger, 2), "timed_region_s": round(dt, 2)}      # where this process's wall time goes
        out["wall_breakdown"] = wall
        secondary = world == 1 and not args.no_compare
        cheap_pre = max(2, min(8, args.sims))                   # stagger of the secondary legs: 8 sims/move (a few tenths of a second)
        if secondary:
            # ---- every kernel of a step against its own roof: 2 move rounds on the SAME engine with events around every launch
            t_sec = time.perf_counter()
            rounds = 2
            eng.profile(True); net.profile(2)
            eng.profile_read(reset=True); net.profile_kernels(reset=True)
            a = eng.stats()
            advance(eng, rounds, True)
            b = eng.stats()
            tree_k, net_k = eng.profile_read(), net.profile_kernels()
            eng.profile(False); net.profile(0)
            out["kernels"] = kernel_table(net_k, tree_k, rounds, b["leaves_evaluated"] - a["leaves_evaluated"], b["simulations"] - a["simulations"],
                                          n, args.channels, args.precision, layer == 3, args.driver)
            out["kernels_note"] = (f"HIP events around every launch, {rounds} further move rounds of the same engine after the timed region "
                                   "(events between launches add a few us each: the sum is slightly above ms_per_step)")
            wall["kernels_s"] = round(time.perf_counter() - t_sec, 2)
        if secondary and not args.no_live_traffic:
            # roofline.traffic: measured now (PMC passes in child processes), not taken from the committed profile
            t_sec = time.perf_counter()
            byts, how = live_traffic(args, layer, cap_main or G)
            if byts is not None:
                out["roofline"]["traffic_from_committed_profile"] = out["roofline"]["traffic"]
                out["roofline"]["traffic"] = byts * (d["leaves_evaluated"] / max(dom_launches, 1)) / float(cap_main or G)
                out["roofline"]["traffic_source"] = how
            else:
                out["roofline"]["live_traffic_error"] = how
            wall["live_traffic_s"] = round(time.perf_counter() - t_sec, 2)
        if secondary and not args.no_cpu_baseline:
            t_sec = time.perf_counter()
            ps = parity_sample(net, eng, n, args.channels, cap_main or G)
            out["parity_sample_max_err"] = max(ps["max_abs_err_pi"], ps["max_abs_err_v"])
            out["parity_sample"] = ps
            wall["parity_sample_s"] = round(time.perf_counter() - t_sec, 2)
        if world == 1:
            del eng                                               # (N > 1: kept for the C-ABI exchange check after the line)
        if secondary and args.dedup == "off":
            t_sec = time.perf_counter()
            # the same workload with the library default (cross-game de-duplication on): identical records, fewer evaluations
            eng2 = make_engine(True)
            eng2.stagger(cheap_pre)
            q, dt2 = measure(eng2, args.steps)
            out["cross_game_dedup"] = {
                "value": q["expansions"] / dt2, "unit": "node-expansions/s", "ms_per_step": dt2 / args.steps * 1e3,
                "games_per_s": q["games_completed"] / dt2, "expansions": int(q["expansions"]), "leaves_evaluated": int(q["leaves_evaluated"]),
                "note": f"same workload (slots staggered at {cheap_pre} sims/move); concurrent games that reach the same board in a step share one "
                        "network evaluation (k_compact). Not the headline: `value` above evaluates every expansion"}
            del eng2
            wall["dedup_compare_s"] = round(time.perf_counter() - t_sec, 2)
        if secondary and args.dedup == "off":
            # the same workload with every sharing the library offers: cross-game de-duplication AND the network's persistent exact-key
            # evaluation cache (the reference's per-search _predict_cache, othelo_mcts.py:82-88, across batches / games / refilled slots)
            t_sec = time.perf_counter()
            net.set_eval_cache(1 << 23)
            engc = make_engine(True, eval_cache=True)
            engc.stagger(cheap_pre)
            c0 = net.eval_cache_stats()
            qc, dtc = measure(engc, args.steps)
            c1 = net.eval_cache_stats()
            out["eval_cache"] = {
                "value": qc["expansions"] / dtc, "unit": "node-expansions/s", "ms_per_step": dtc / args.steps * 1e3,
                "games_per_s": qc["games_completed"] / dtc, "sims_per_s": qc["simulations"] / dtc,
                "expansions": int(qc["expansions"]), "leaves_evaluated": int(qc["leaves_evaluated"]),
                "hit_rate": (c1["hits"] - c0["hits"]) / max(c1["lookups"] - c0["lookups"], 1), "cache_entries": c1["entries"],
                "note": f"same workload (slots staggered at {cheap_pre} sims/move, which also warms the cache), de-duplication on, "
                        "oz_selfplay_config.eval_cache = 1: a leaf whose board this network has evaluated before takes (pi, v) from the "
                        "network's HBM cache and needs no batch slot; records identical (tests/test_gpu_bench_config.py). Not the headline: "
                        "`value` above evaluates every expansion"}
            del engc
            net.set_eval_cache(0)
            wall["eval_cache_s"] = round(time.perf_counter() - t_sec, 2)
        if secondary:
            # the same workload under the library's other driver (identical records per game: tests/test_gpu_bench_config.py)
            t_sec = time.perf_counter()
            other = "free" if args.driver == "lockstep" else "lockstep"
            engo = make_engine(args.dedup == "on")
            engo.stagger(cheap_pre)
            qo, dto = measure(engo, args.steps, driver=other)
            out["other_driver"] = {
                "driver": other, "value": qo["expansions"] / dto, "unit": "node-expansions/s", "ms_per_step": dto / args.steps * 1e3,
                "games_per_s": qo["games_completed"] / dto, "sims_per_s": qo["simulations"] / dto,
                "leaves_per_batch": qo["leaves_evaluated"] / (args.steps * args.sims),
                "note": ("oz_selfplay_run_steps: every game runs on by itself (network-free simulations and its move ride in the same launch), "
                         "so nearly every slot of a batch carries a leaf; a game's records are those of lock step bit for bit"
                         if other == "free" else "oz_selfplay_run: one simulation per game per batch, moves aligned")
                        + f"; slots staggered at {cheap_pre} sims/move"}
            del engo
            wall["other_driver_s"] = round(time.perf_counter() - t_sec, 2)
        if secondary and layer == 3:
            # the same steps with conv1 / conv2 evaluated the plain way (conv1 kernel + conv2 as an MFMA implicit GEMM,
            # no pattern tables): what the table form buys, and a number for readers who want every layer as a GEMM
            t_sec = time.perf_counter()
            net.set_tables(0)
            eng3 = make_engine(args.dedup == "on")
            eng3.stagger(cheap_pre)
            g3, dt3 = measure(eng3, max(args.steps // 2, 1))
            net.set_tables(-1)
            out["all_layers_as_gemm"] = {
                "value": g3["expansions"] / dt3, "unit": "node-expansions/s", "ms_per_step": dt3 / max(args.steps // 2, 1) * 1e3,
                "games_per_s": g3["games_completed"] / dt3, "flop_per_expansion_executed": flop_ref,
                "note": "same workload, conv1 as a kernel and conv2 as an MFMA implicit GEMM (oz_net_set_tables(net, 0)); "
                        "(pi, v) agree with the table form to 5e-7"}
            del eng3
            wall["gemm_compare_s"] = round(time.perf_counter() - t_sec, 2)
        if secondary and args.precision == "f16x2":
            # ---- exact fp32 arithmetic (what the reference computes in: Net/NNet.py:85), same workload, same run
            t_sec = time.perf_counter()
            steps32 = max(args.steps // 4, 2)
            net32 = NNetWrapper((n, n), num_channels_1=args.channels, max_batch=G, seed=0, precision="f32")
            e32 = make_engine(args.dedup == "on", the_net=net32)
            e32.stagger(cheap_pre)
            q32, dt32 = measure(e32, steps32, net32)
            ms32, l32 = net32.profile_read()
            layer32 = net32.profiled_layer()
            ach32 = q32["leaves_evaluated"] * conv_flop_per_leaf(layer32, n, args.channels) / max(ms32 * 1e-3, 1e-9) / 1e12
            out["exact_fp32"] = {
                "value": q32["expansions"] / dt32, "unit": "node-expansions/s", "ms_per_step": dt32 / steps32 * 1e3, "steps": steps32,
                "games_per_s": q32["games_completed"] / dt32, "sims_per_s": q32["simulations"] / dt32, "dtype": "f32",
                "dtype_detail": "fp32 operands and accumulators on v_mfma_f32_32x32x2_f32 (conv1 + conv2 from exact-fp32 pattern tables)",
                "roofline": roofline("f32", layer32, ach32, ms32, l32, q32["leaves_evaluated"], n, args.channels),
                "note": f"same workload and run as the headline (slots staggered at {cheap_pre} sims/move), precision='f32'"}
            del e32, net32
            wall["exact_fp32_s"] = round(time.perf_counter() - t_sec, 2)
        if secondary and n == 8:
            # ---- BASELINE configs[3]: 6x6 boards, same network family, same engine
            t_sec = time.perf_counter()
            net6 = NNetWrapper((6, 6), num_channels_1=args.channels, max_batch=G, seed=0, precision=args.precision)
            e6 = make_engine(args.dedup == "on", the_net=net6, board=6)
            e6.stagger(cheap_pre)
            q6, dt6 = measure(e6, args.steps)
            out["config4"] = {
                "workload": f"{G} concurrent 6x6 self-play games, {args.sims} sims/move (BASELINE configs[3])",
                "value": q6["expansions"] / dt6, "unit": "node-expansions/s", "ms_per_step": dt6 / args.steps * 1e3,
                "games_per_s": q6["games_completed"] / dt6, "sims_per_s": q6["simulations"] / dt6,
                "flop_per_expansion_reference": FLOP_PER_EXPANSION[6]}
            del e6, net6
            wall["config4_s"] = round(time.perf_counter() - t_sec, 2)
        if secondary and n == 8 and not args.no_cpu_baseline:
            # ---- BASELINE configs[4]: arena evaluation with two real networks (bounded plies), sampled games replayed by the oracle
            t_sec = time.perf_counter()
            out["config5"] = config5_arena(args.channels, args.precision, args.arena_plies)
            wall["config5_s"] = round(time.perf_counter() - t_sec, 2)
        if secondary and n == 8 and not args.no_cpu_baseline:
            t_sec = time.perf_counter()
            out["dropin_config0"] = dropin_config0(args.channels, args.precision)
            wall["dropin_config0_s"] = round(time.perf_counter() - t_sec, 2)
        if world == 1 and not args.no_cpu_baseline:
            t_sec = time.perf_counter()
            out["cpu_baseline"] = cpu_baseline(n, args.channels, args.sims)
            if secondary and n == 8:
                out["config4"]["cpu_baseline"] = cpu_baseline_config(6, args.channels, args.sims, 5.0, out["cpu_baseline"]["cores"])
            wall["cpu_baseline_s"] = round(time.perf_counter() - t_sec, 2)
        # the whole metric once more as the LAST key of the line: a reader (or a log) that keeps only the tail of stdout still sees it
        out["headline"] = {"node_expansions_per_s": out["value"], "games_per_s": out["games_per_s"], "sims_per_s": out["sims_per_s"],
                           "ms_per_step": out["ms_per_step"], "n_gpus": world, "roofline_frac": out["roofline"]["frac"],
                           "roofline_avg_launch_ms": out["roofline"]["avg_launch_ms"], "pooled_records": out["pooled_records"],
                           "games_completed": out["games_completed"]}
        print(json.dumps(out), flush=True)
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
