"""bench_legs.py -- everything of bench.py that is NOT the timed region: the roofline arithmetic, the per-kernel table, and the secondary
legs that ride in the same JSON line at N = 1 (exact fp32, 6x6, the other driver, de-duplication / cache, conv2 as a GEMM, BASELINE
configs[4] and configs[0], the parity sample of both precisions, the CPU baselines).  bench.py holds the launcher and the timed region
and calls run_secondary() after it; nothing here runs inside the timed region."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
BENCH_PY = os.path.join(ROOT, "bench.py")

FLOP_PER_EXPANSION = {8: 566428672, 6: 270185472}          # SURVEY.md 8(d), whole OthelloNN forward
PEAK_F32_MATRIX_TFLOPS = 157.3                             # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_F16_MATRIX_TFLOPS = 2500.0                            # MI355X_MICROARCH.md: ~2.5 PF dense fp16/bf16 MFMA
PEAK_HBM_GBPS = 8000.0                                     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_L2_GBPS = 34500.0                                     # MI355X_MICROARCH.md: L2 aggregate ~34.5 TB/s (8 XCDs x 4 MiB)
TREE_BYTES_PER_SIM = 1300
PRECISIONS = ("f32", "f16x2", "bf16x3")                    # oz_net_set_precision modes bench.py measures
DTYPE_LABEL = {"f32": "f32", "f16x2": "f32 (2xf16 split)", "bf16x3": "f32 (3xbf16 split)"}
DTYPE_DETAIL = {
    "f32": "fp32 operands and accumulators on v_mfma_f32_32x32x2_f32 (conv1 + conv2 from exact-fp32 pattern tables): the reference's arithmetic (Net/NNet.py:85)",
    "f16x2": "fp32 values carried as two fp16 planes (22 of 24 significand bits, placed per channel at commit), 3 fp16 MFMA products per fp32 product, fp32 "
             "accumulate; pi, v within 1e-5 of float64 -- fp32-equivalent for this network, narrower than fp32",
    "bf16x3": "fp32 values carried as three bf16 planes (8 + 8 + 8 significand bits: every normal fp32 value exactly, fp32's exponent range, no scaling), 6 bf16 MFMA "
              "products per fp32 product (a2b3, a3b2, a3b3 <= 2^-24 relative dropped), fp32 accumulate"}                                  # algorithmic HBM bytes per simulation on the tree side


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
    el = 6 if precision == "bf16x3" else 4                  # bytes per tensor element: three bf16 planes, or fp32 / two fp16 planes
    r["algorithmic_bytes_per_launch"] = (expansions / max(launches, 1)) * (px_in + px_out) * channels * el + 9 * channels * channels * el
    if precision == "bf16x3":
        kernel = (f"k_gemm_b3<{layer}> (conv{layer}: 3x3, 512->512, 128 x 256 tiles, implicit GEMM, f32 as 3xbf16 split, 6 products on v_mfma_f32_16x16x32_bf16, "
                  "2-phase ping-pong loop); conv1 + conv2 = exact-fp32 table gather-sum")
        r.update(kernel=kernel, peak=PEAK_F16_MATRIX_TFLOPS, frac=achieved / PEAK_F16_MATRIX_TFLOPS, mfma_products_per_fp32_product=6,
                 matrix_pipe_tflops=6 * achieved, matrix_pipe_frac=6 * achieved / PEAK_F16_MATRIX_TFLOPS, vs_fp32_matrix_peak=achieved / PEAK_F32_MATRIX_TFLOPS)
    elif precision == "f32":
        r.update(kernel=("k_gemm_f32 (conv3: 3x3 valid, 512->512, 8x8 -> 6x6, implicit GEMM, v_mfma_f32_32x32x2_f32); conv1 + conv2 = k_conv2_lut_f32 table gather-sum"
                         if layer == 3 else "k_gemm_f32 (conv2: 3x3 same, 512->512, implicit GEMM, v_mfma_f32_32x32x2_f32)"),
                 peak=PEAK_F32_MATRIX_TFLOPS, frac=achieved / PEAK_F32_MATRIX_TFLOPS)
    else:
        loop = ("4-phase ping-pong loop" if (layer != 3 or conv3_rows == 256) else "2-phase ping-pong loop" if conv3_rows == 192 else
                "1-phase ping-pong loop on three LDS stages")
        tail = f"implicit GEMM, f32 as 2xfp16 split, v_mfma_f32_16x16x32_f16, {loop}"
        if layer == 3:
            cfg = {256: "H2BigPP", 192: "H2MidPP", 128: "H2LowPP1"}.get(conv3_rows, "H2MidPP")
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
                rel = os.path.join("profiles", f"r5_{'f32' if precision == 'bf16x3' else precision}_bench_pmc_by_shape.csv")
                rows = [r for r in csv.DictReader(l for l in open(os.path.join(ROOT, rel)) if not l.startswith("#")) if "k_conv2_lut" in r["kernel"]]
                fetch = max(float(r["avg_per_launch"]) for r in rows if r["counter"] == "FETCH_SIZE")
                write = max(float(r["avg_per_launch"]) for r in rows if r["counter"] == "WRITE_SIZE")
                per_leaf = (fetch * 2048 + write * 1024) / 3640.0 * (n * n / 64.0) * (C / 512.0)
                src = f"{rel}: FETCH_SIZE x2 + WRITE_SIZE of the gather per 3640-leaf launch (PMC passes of this command); not re-measured in this run"
            except Exception:
                per_leaf = 0.58e9 / 3640 * (n * n / 64.0) * (C / 512.0) + n * n * C * 4
                src = "0.58 GB read (PMC, round 3) + the compulsory output row per pixel per 3640-leaf launch; not re-measured in this run"
            byts = leaves * per_leaf
            row.update(kernel=("k_conv2_lut_xcd" if C == 512 else "k_conv2_lut") + {"f32": " (fp32 rows)", "f16x2": " (h2 rows)", "bf16x3": " (fp32 tables, b3 rows)"}[precision], bound="hbm",
                       achieved=byts / sec / 1e9, peak=PEAK_HBM_GBPS, unit="GB/s", bytes_source=src)
        elif name in flop:
            kern = {"f32": "k_gemm_f32", "f16x2": "k_gemm_h2", "bf16x3": "k_gemm_b3" if name in ("conv3", "conv4", "fc1") else "k_gemm_f32"}[precision]
            peak_row = PEAK_F32_MATRIX_TFLOPS if kern == "k_gemm_f32" else peak_mm
            row.update(kernel=f"{kern} ({name})", bound="mfma", achieved=leaves * flop[name] / sec / 1e12, peak=peak_row, unit="TFLOP/s")
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


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            return next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        return ""


GPU_BOX_CPU_SHARE_PER_GPU = 16     # the pool's stated CPU share of a one-GPU box ("size worker pools to the box's CPU share: 16 for one GPU")


def host_cpu_share():
    """the host threads this process may use, and why: its scheduler affinity, its cgroup CPU quota (cpu.max), what OpenMP would start --
    the CPU legs use the smallest of them; only when NEITHER the affinity nor the cgroup says less than the whole host (the GPU boxes show all
    256 CPUs of the node to every lease) the pool's stated share per GPU caps it.  Returned with every CPU baseline (`host_share`)."""
    import oracle
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            parts = open(path).read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    quota = float(parts[0]) / float(parts[1])
            else:
                q = float(parts[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    omp = int(oracle.lib().orc_nn_max_threads())
    limits = [affinity, omp] + ([max(1, int(quota))] if quota else [])
    used, rule = max(1, min(limits)), "min(sched_getaffinity, cgroup cpu.max, OpenMP max threads)"
    if affinity >= (os.cpu_count() or 1) and not quota and used > GPU_BOX_CPU_SHARE_PER_GPU:
        used, rule = GPU_BOX_CPU_SHARE_PER_GPU, (f"neither the affinity mask nor a cgroup quota restricts this process (all {affinity} host CPUs visible): "
                                                 f"the pool's stated share of {GPU_BOX_CPU_SHARE_PER_GPU} CPUs per GPU")
    return {"threads_used": used, "sched_getaffinity": affinity, "cgroup_cpu_quota": quota, "openmp_max_threads": omp,
            "host_cpus": os.cpu_count(), "rule": rule}


def host_threads():
    return host_cpu_share()["threads_used"]


def cpu_baseline(n, channels, sims, budget_s=12.0):
    """The oracle port of the reference path on this host's cores, three ways:
      value                -- sequential games, batch-1 leaf evaluation by the float32 C restatement of OthelloNN with OpenMP INSIDE each
                              evaluation (what one reference process with a multi-threaded TensorFlow does), whole games for ~budget_s;
      one_game_per_thread  -- as many concurrent games as host threads, every game on its own thread with a single-threaded evaluator
                              (the reference's ThreadWorker / one-VM-per-worker fan-out, workers.py:33-37,82-90): a bounded number of plies each;
      one_thread           -- ONE whole game on ONE thread at BASELINE configs[0]'s 25 sims/move (a whole 100-sims game takes minutes there).
    `reference_python` echoes the measurement of the reference's OWN Python in the build container (tools/time_reference_python.py)."""
    import threading
    import oracle
    from othellozero_amd.weights import init_weights
    w = init_weights(n, seed=0, channels=channels)
    threads = host_threads()
    net = oracle.CNet(w, n, channels=channels, nthreads=threads)

    def run(game_id, max_moves, the_net=net, the_sims=sims):
        m = oracle.Mcts(n, 1.0, oracle.QMODE_F64, evaluator=the_net.evaluator())    # a fresh tree per episode (training.py:29)
        t0 = time.perf_counter()
        ep = m.episode(the_sims, 1.0, 0.9, 1234, game_id, max_moves=max_moves)
        return time.perf_counter() - t0, ep
    run(0, 1)                                                                        # untimed: thread pool / cache warm-up
    t, exp, plies, games = 0.0, 0, 0, 0
    while t < budget_s and games < 64:                                               # whole games until the budget is used
        dt, ep = run(games, n * n)
        t, exp, plies, games = t + dt, exp + ep["stats"]["expansions"], plies + ep["n_moves"], games + 1
    out = {
        "value": exp / t, "unit": "node-expansions/s", "cores": threads, "kind": "port",
        "sample": f"{plies} plies of {games} sequential {n}x{n} game(s) at {sims} sims/move "
                  f"({exp} expansions, {plies * sims} sims, {t:.1f} s), batch-1 leaf eval, OpenMP x{threads} inside every evaluation",
        "sims_per_s": plies * sims / t, "games_per_s": games / t, "cpu_model": _cpu_model(), "host_cpus": os.cpu_count(),
        "host_share": host_cpu_share(),
    }
    # ---- one game per thread, all the host threads this process has: every thread its own tree and its own single-threaded evaluator
    nets1 = [oracle.CNet(w, n, channels=channels, nthreads=1) for _ in range(threads)]
    plies_each = 3
    res = [None] * threads

    def worker(k):
        res[k] = run(1000 + k, plies_each, nets1[k])
    t0 = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    wall = time.perf_counter() - t0
    exp_t = sum(r[1]["stats"]["expansions"] for r in res)
    plies_t = sum(r[1]["n_moves"] for r in res)
    out["one_game_per_thread"] = {
        "value": exp_t / wall, "unit": "node-expansions/s", "cores": threads, "kind": "port", "sims_per_s": plies_t * sims / wall,
        "games_per_s_at_60_plies": plies_t / 60.0 / wall,
        "sample": f"{threads} concurrent {n}x{n} games, one per host thread, the first {plies_each} plies of each at {sims} sims/move "
                  f"({exp_t} expansions, {wall:.1f} s wall), batch-1 leaf eval on the game's own thread"}
    # ---- one whole game on one thread (BASELINE configs[0]: 25 sims/move)
    dt1, ep1 = run(0, n * n, nets1[0], 25)
    out["one_thread"] = {
        "value": ep1["stats"]["expansions"] / dt1, "unit": "node-expansions/s", "cores": 1, "kind": "port",
        "games_per_s": 1.0 / dt1, "sims_per_s": ep1["n_moves"] * 25 / dt1,
        "sample": f"ONE whole {n}x{n} game at 25 sims/move (BASELINE configs[0]) on one thread: {ep1['n_moves']} plies, "
                  f"{ep1['stats']['expansions']} expansions, {dt1:.1f} s"}
    out["value_1_thread"] = out["one_thread"]["value"]
    try:
        ref = json.load(open(os.path.join(ROOT, "profiles", "reference_python_cpu.json")))
        pick = next(r for r in ref["runs"] if r["board"] == n and r["sims_per_move"] == sims)
        out["reference_python"] = {
            "value": pick["node_expansions_per_s"], "unit": "node-expansions/s", "games_per_s": pick["games_per_s"], "sims_per_s": pick["sims_per_s"],
            "kind": "reference", "cores": ref["nn_threads"], "cpu_model": ref["cpu_model"],
            "measured_in": "the BUILD CONTAINER, not on this box (the reference never travels): tools/time_reference_python.py -> "
                           "profiles/reference_python_cpu.json",
            "sample": f"{pick['workload']}: training.execute_episode of the reference itself, {pick['moves']} moves, {pick['node_expansions']} expansions, "
                      f"{pick['seconds']} s ({pick['seconds_in_python_tree_and_rules']} s in the reference's Python tree / rules code, the rest in the leaf "
                      "evaluator = the build's float32 C restatement of OthelloNN behind neural_network.predict)",
            "all_runs": ref["runs"]}
    except Exception as e:                                       # noqa: BLE001 -- an echo of a committed file, never a reason to fail
        out["reference_python"] = {"error": repr(e)}
    return out


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
    threads = host_threads()
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
    output row is an independent accumulation), and bit-compared with a second, shorter call (another conv3 tile, same sums).
    Returns (report, sample) -- `sample` = (own, opp, rows, pi64, v64) so that the other precision is checked on the SAME rows
    (parity_rows, called by the exact-fp32 leg)."""
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
    rep = {"max_abs_err_pi": float(np.abs(pi.reshape(call_size, -1)[rows] - pi64).max()), "max_abs_err_v": float(np.abs(v[rows] - v64).max()),
           "precision": net.precision,
           "positions_in_the_call": int(call_size), "conv3_tile_rows_of_the_call": int(tile), "rows_checked_vs_float64": int(rows.size),
           "bit_identical_to_a_shorter_call": bool(np.array_equal(p2, pi[:short]) and np.array_equal(v2, v[:short])),
           "shorter_call": {"positions": int(short), "conv3_tile_rows": int(net.conv3_tile_rows())},
           "plies_sampled": [int(st["ply"][idx].min()), int(st["ply"][idx].max())], "tolerance": 1e-5,
           "checker": "oracle/nn_numpy.py (float64 restatement of Net/OthelloNN.py:42-56)"}
    if net.precision == "f16x2":
        dpi, dv, npos = net.self_check()
        rep["commit_self_check"] = {"max_abs_d_pi": dpi, "max_abs_d_v": dv, "positions": npos, "limit": 8e-6,
                                    "note": "oz_net_commit: the calibration positions through the f16x2 kernels and through the exact-fp32 kernels"}
    return rep, (own, opp, rows, pi64, v64)


def parity_rows(net, sample):
    """another network object (the other precision) on the SAME call and the SAME rows as parity_sample"""
    import numpy as np
    own, opp, rows, pi64, v64 = sample
    pi, v = net.predict_batch(own, opp)
    return {"max_abs_err_pi": float(np.abs(pi.reshape(own.size, -1)[rows] - pi64).max()), "max_abs_err_v": float(np.abs(v[rows] - v64).max()),
            "precision": net.precision, "positions_in_the_call": int(own.size), "rows_checked_vs_float64": int(rows.size), "tolerance": 1e-5}


def live_traffic(args, layer, grid_leaves, timeout_s=150, precision=None):
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
    vals, child_leaves = {}, {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out_dir, "-o", "run", "--",
                   sys.executable, BENCH_PY, "--steps", "1", "--warmup", "0", "--stagger-sims", "8", "--no-compare", "--no-cpu-baseline",
                   "--precision", precision or args.precision, "--games", str(args.games), "--sims", str(args.sims), "--board", str(args.board),
                   "--channels", str(args.channels), "--driver", args.driver, "--batch-cap", str(args.batch_cap), "--dedup", args.dedup]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=env)
            files = glob.glob(os.path.join(out_dir, "**", "*_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}"
            try:      # what the child's launches really held (its own line): the caller scales the bytes to ITS leaves per launch with this, not with the cap
                cl = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                child_leaves[counter] = cl["leaves_evaluated_rank0"] / max(cl["roofline"]["launches"], 1)
            except Exception:                                    # noqa: BLE001
                child_leaves[counter] = float(grid_leaves)
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
    # bytes per LEAF of the child's launches (each counter by its own pass's leaves per launch) x the leaves per launch the caller asks about
    byts = (vals["FETCH_SIZE"] * 1024 * 2 / child_leaves["FETCH_SIZE"] + vals["WRITE_SIZE"] * 1024 / child_leaves["WRITE_SIZE"]) * float(grid_leaves)
    return byts, (f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two separate child runs of bench.py --steps 1 --stagger-sims 8 "
                  f"--no-compare), average over the {args.sims} launches of the timed step ({child_leaves['FETCH_SIZE']:.0f} leaves per launch there); FETCH_SIZE x2 (gfx950) "
                  f"= {vals['FETCH_SIZE'] * 2048 / 1e9:.3f} GB read + {vals['WRITE_SIZE'] * 1024 / 1e9:.3f} GB written per launch")


def config5_arena(channels, precision, plies=0, games=512, sims=800, sample=2, dedup_compare_plies=6):
    """BASELINE configs[4]: arena evaluation (agents.py:44-84, duel_between_agents) -- `games` parallel 8x8 games, `sims` simulations per
    move per agent, two REAL networks (seeds 0 / 1: best vs candidate), deterministic play (temperature 0, RNG_TIE stream), every game
    played TO THE END (plies = 0; a positive value bounds the games, for quick looks only); `sample` WHOLE games are replayed by the CPU
    oracle's arena fed with the GPU networks' own (pi, v) -> sample_mismatches must be 0.  The timed run evaluates every expansion by
    itself; the library default (cross-game de-duplication) is timed beside it on the first `dedup_compare_plies` plies."""
    import numpy as np
    import oracle
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.agents import arena_batch
    n = 8
    nets = [NNetWrapper((n, n), num_channels_1=channels, max_batch=games, seed=sd, precision=precision) for sd in (0, 1)]
    arena_batch(nets[0], nets[1], n, games, 8, 1.0, seed=11, first_game_id=0, q_mode=1, max_rounds=2)      # untimed: allocation, code load
    # headline of the leg: every expansion evaluated by itself (cross-game de-duplication off, as in the self-play headline); HIP events
    # around the dominant launch of both networks (conv3) ride along: two events per simulation step, the same instrumentation as the
    # self-play timed region
    for nt in nets:
        nt.profile_kernels(reset=True); nt.profile(1)
    t0 = time.perf_counter()
    r = arena_batch(nets[0], nets[1], n, games, sims, 1.0, seed=11, first_game_id=0, q_mode=1, max_rounds=plies, dedup=False)
    dt = time.perf_counter() - t0
    dom = [nt.profile_read() for nt in nets]
    dom_layer = nets[0].profiled_layer()
    conv3_rows = nets[0].conv3_tile_rows()
    for nt in nets:
        nt.profile(0)
    # the library default (on) -- arena games start from ONE opening and play deterministically, so most expansions of a step share a board --
    # on the opening plies, against the same plies of the run above
    t0 = time.perf_counter()
    rd = arena_batch(nets[0], nets[1], n, games, sims, 1.0, seed=11, first_game_id=0, q_mode=1, max_rounds=dedup_compare_plies, dedup=True)
    dtd = time.perf_counter() - t0
    # ... and every sharing the library offers: de-duplication + the two networks' persistent evaluation caches, WHOLE games, next to
    # the pure number -- not instead of it -- and checked against it move for move over all games
    from othellozero_amd import _lib as _ozlib
    for nt in nets:
        nt.set_option(_ozlib.NET_OPT_LATENCY_SPLITS, 1)        # batches of a few leaves: k-splits chosen for latency (oz_net_set_option; a per-network constant)
        nt.commit()
        nt.set_eval_cache(1 << 22)
    t0 = time.perf_counter()
    rc = arena_batch(nets[0], nets[1], n, games, sims, 1.0, seed=11, first_game_id=0, q_mode=1, max_rounds=plies, dedup=True, eval_cache=True)
    dtc = time.perf_counter() - t0
    cstats = [nt.eval_cache_stats() for nt in nets]
    for nt in nets:
        nt.set_eval_cache(0)
    # its checker: the SAME networks (same k-splits, so the same bits) with every expansion evaluated by itself, on the opening plies
    rp = arena_batch(nets[0], nets[1], n, games, sims, 1.0, seed=11, first_game_id=0, q_mode=1, max_rounds=dedup_compare_plies, dedup=False)
    kc = np.minimum(rp["n_moves"], rc["n_moves"])
    livec = np.arange(128)[None, :] < kc[:, None]
    same_cached_plies = bool(np.array_equal(rp["actions"][livec], rc["actions"][livec]) and np.array_equal(rp["players"][livec], rc["players"][livec]))
    for nt in nets:
        nt.set_eval_cache(0)
        nt.set_option(_ozlib.NET_OPT_LATENCY_SPLITS, 0)
        nt.commit()
    # (against the pure run ABOVE the networks differ in their k-splits, i.e. in the last bits of (pi, v): the games usually agree, and how many do is reported)
    games_equal_to_pure = int(sum(1 for g_ in range(games) if r["n_moves"][g_] == rc["n_moves"][g_]
                                  and np.array_equal(r["actions"][g_], rc["actions"][g_]) and r["winner"][g_] == rc["winner"][g_] and r["points"][g_] == rc["points"][g_]))
    k_cmp = np.minimum(rd["n_moves"], r["n_moves"])
    live = np.arange(128)[None, :] < k_cmp[:, None]                             # the plies both runs played
    same = bool(np.array_equal(r["actions"][live], rd["actions"][live]) and np.array_equal(r["players"][live], rd["players"][live]))
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
    bad, replayed_plies = 0, 0
    for gi in range(sample):
        o = oracle.arena(oracle.Mcts(n, 1.0, 1, evaluator=evaluator(0)), oracle.Mcts(n, 1.0, 1, evaluator=evaluator(1)), sims, 11, gi, max_plies=plies if plies > 0 else -1)
        k = o["n_moves"]
        replayed_plies += int(k)
        ok = (int(r["n_moves"][gi]) == k and np.array_equal(r["actions"][gi][:k], o["action"]) and np.array_equal(r["players"][gi][:k], o["player"])
              and (int(r["final_black"][gi]), int(r["final_white"][gi])) == (int(o["final_black"]), int(o["final_white"])))
        if plies <= 0:
            ok = ok and bool(o["finished"]) and int(r["winner"][gi]) == int(o["winner"]) and int(r["points"][gi]) == int(o["points"])
        bad += 0 if ok else 1
    whole = plies <= 0
    # ---- the regime in kernels: a network batch holds the leaves of ONE agent's movers (<= `games`), so every GEMM runs far below the shapes
    # of the self-play line -- roofline of the dominant launch from the timed run above, kernels[] from a bounded run of `kernel_plies` plies
    # with events around every launch of both networks and both searches (events between ~12 launches per 0.4 ms step are not free: the
    # timed run above carries only the two around conv3)
    dom_ms, dom_launches = dom[0][0] + dom[1][0], dom[0][1] + dom[1][1]
    leaves = int(r["leaves_evaluated"])
    ach = leaves * conv_flop_per_leaf(dom_layer, n, channels) / max(dom_ms * 1e-3, 1e-9) / 1e12
    roof = roofline(precision, dom_layer, ach, dom_ms, dom_launches, leaves, n, channels, conv3_rows=conv3_rows)
    roof["traffic"], roof["traffic_source"] = None, "not measured at this launch shape"
    roof["leaves_per_launch"] = leaves / max(dom_launches, 1)
    roof["grid_note"] = (f"{leaves / max(dom_launches, 1):.0f} leaves per launch on average; the forward picks the tile per call: 192 rows (2-phase loop) while 456 .. 512 "
                         "games are alive (512 leaves = 18432 rows = 192 blocks on 256 CUs, one k-slice each), 128 rows below (<= 256 blocks); conv3_tile_rows of the "
                         f"LAST call of the run: {conv3_rows}")
    kernel_plies = 4
    for nt in nets:
        nt.profile_kernels(reset=True); nt.profile(2)
    rk = arena_batch(nets[0], nets[1], n, games, sims, 1.0, seed=11, first_game_id=0, q_mode=1, max_rounds=kernel_plies, dedup=False, profile=True)
    net_k = {}
    for nt in nets:
        for name, (ms, cnt) in nt.profile_kernels().items():
            a0 = net_k.setdefault(name, [0.0, 0])
            a0[0] += ms; a0[1] += cnt
        nt.profile(0)
    steps_k = kernel_plies * sims
    stk = rk["stats_black"] + rk["stats_white"]
    ktab = kernel_table({k: tuple(v) for k, v in net_k.items()}, rk["tree_kernels"], steps_k, int(rk["leaves_evaluated"]), int(stk[0]), n, channels,
                        precision, dom_layer == 3)
    for row in ktab:                                            # a "step" of this table is ONE simulation step (one network batch), not 100 of them
        row["us_per_sim_step"] = row.pop("ms_per_step") * 1e3
        row["launches_per_sim_step"] = row.pop("launches_per_step")
    out = {"workload": f"BASELINE configs[4]: {games} parallel 8x8 arena games, {sims} sims/move per agent, two {channels}-filter OthelloNN "
                       f"(random init, seeds 0 / 1), temperature 0, deterministic; " + ("every game played to the end" if whole else f"first {plies} plies of every game"),
           "seconds": dt, "games": games, "games_finished": int((r["n_moves"] > 0).sum()) if whole else 0, "moves": moves,
           "moves_per_game": moves / games, "simulations": int(st[0]), "expansions": int(st[2]),
           "sims_per_s": float(st[0]) / dt, "value": float(st[2]) / dt, "unit": "node-expansions/s", "moves_per_s": moves / dt,
           "precision": precision, "leaves_evaluated": int(r["leaves_evaluated"]),
           "black_wins": int((r["winner"] == 1).sum()),
           "with_cross_game_dedup": {"plies": dedup_compare_plies, "seconds": dtd, "sims_per_s": float(rd["stats_black"][0] + rd["stats_white"][0]) / dtd,
                                     "leaves_evaluated": int(rd["leaves_evaluated"]), "identical_moves_on_those_plies": same,
                                     "note": "library default: a board several games reach in one step is evaluated once (timed on the opening plies only)"},
           "us_per_sim_step": dt / max(moves / games * sims, 1) * 1e6,
           "roofline": roof, "kernels": ktab,
           "kernels_note": f"HIP events around every launch of both networks and both searches over the first {kernel_plies} plies of a second run "
                           "(per simulation step = one network batch of one agent); roofline.avg_launch_ms is from the timed whole-game run",
           "with_dedup_and_eval_cache": {"seconds": dtc, "games_per_s": games / dtc if whole else None,
                                         "sims_per_s": float(rc["stats_black"][0] + rc["stats_white"][0]) / dtc,
                                         "leaves_evaluated": int(rc["leaves_evaluated"]), "expansions": int((rc["stats_black"] + rc["stats_white"])[2]),
                                         "cache_hit_rate": sum(c["hits"] for c in cstats) / max(sum(c["lookups"] for c in cstats), 1),
                                         "identical_moves_on_the_opening_plies_vs_every_expansion_evaluated": same_cached_plies, "plies_compared": dedup_compare_plies,
                                         "games_identical_to_the_pure_run_above": games_equal_to_pure,
                                         "note": "oz_arena_set_eval_cache + cross-game de-duplication + OZ_NET_OPT_LATENCY_SPLITS (batches of ~15 leaves): checked move for move on the opening plies against the same "
                                                 "networks evaluating every expansion (same k-splits = same bits); whole games compared with the pure run above, whose networks differ "
                                                 "in the last bits of (pi, v); reported BESIDE the pure number"},
           "sample_games_replayed_by_oracle": sample, "sample_plies_replayed": replayed_plies, "sample_mismatches": bad,
           "oracle_replay_s": round(time.perf_counter() - t1, 2),
           "note": "oz_arena_run_rounds: BLACK movers search in net A's trees, WHITE movers in net B's, one searched ply per game and round; "
                   "the oracle's arena (agents.py restated, oracle/oz_oracle.c orc_arena_plies) is fed the GPU networks' own (pi, v), so "
                   "actions, movers, boards, winner and points must agree bit for bit"}
    if whole:
        out["games_per_s"] = games / dt
    else:
        out["plies_per_game"] = plies
    # the same regime in the library's other precisions, on the opening plies only (a whole exact-fp32 arena would take minutes): microseconds per simulation
    # step with every expansion evaluated by itself -- what configs[4] costs in exact fp32 and in the fp32-class bf16x3 mode beside the f16x2 headline of the leg
    del nets
    out["other_precisions"] = {}
    for prec in [p for p in PRECISIONS if p != precision]:
        try:
            pn = [NNetWrapper((n, n), num_channels_1=channels, max_batch=games, seed=sd, precision=prec) for sd in (0, 1)]
            arena_batch(pn[0], pn[1], n, games, 8, 1.0, seed=11, first_game_id=0, q_mode=1, max_rounds=2)
            t0 = time.perf_counter()
            rp2 = arena_batch(pn[0], pn[1], n, games, sims, 1.0, seed=11, first_game_id=0, q_mode=1, max_rounds=3, dedup=False)
            dtp = time.perf_counter() - t0
            steps_p = float(rp2["n_moves"].sum()) / games * sims
            out["other_precisions"][prec] = {"us_per_sim_step": dtp / max(steps_p, 1) * 1e6, "plies": 3, "seconds": dtp,
                                             # (whole games at the opening plies' step time: the opening's full 512-leaf batches are the slowest steps of a game)
                                             "games_per_s_at_this_step_time": games / max(dtp / max(steps_p, 1) * (moves / games) * sims, 1e-12) if whole else None}
            del pn
        except Exception as e:                                   # noqa: BLE001 -- a side figure
            out["other_precisions"][prec] = {"error": repr(e)}
    return out



def _sig(x, digits=6):
    """a float with `digits` significant digits (the secondary scalars of the stdout line; the contract fields keep every digit)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def _finite(x):
    """a JSON-safe copy: non-finite floats become null (strict JSON has no NaN / Infinity; a single bad secondary figure must never cost the line)"""
    if isinstance(x, float):
        return x if x == x and x not in (float("inf"), float("-inf")) else None
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    return x


def compact_line(out, limit=6000):
    """THE stdout line: strict JSON, contract keys first, <= `limit` bytes whatever the rank count -- everything else of `out` lives in
    bench_detail.json (written next to bench.py) and is never needed to read the metric.  VERDICT r5 item 1: the 33 KB line of round 5 did
    not fit the driver's 8 KB tail and was recorded as unparsed."""
    r = out["roofline"]
    cfg = out["config"]
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": cfg["workload_short"], "games_per_gpu": cfg["games_per_gpu"], "sims_per_move": cfg["sims_per_move"], "board": cfg["board"],
                      "driver": cfg["driver"], "batch_cap": cfg["batch_cap"], "parallelism": cfg["parallelism"], "backend": cfg["backend"],
                      "leaf_dedup": cfg["leaf_dedup"].split(":")[0], "precision": cfg["precision"]}
    line["roofline"] = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms", "kernel",
                                              "algorithmic_bytes_per_launch", "flop_per_leaf", "leaves_per_launch")}
    line["roofline"]["kernel"] = r.get("kernel_short") or r.get("kernel")
    if "matrix_pipe_frac" in r:
        line["roofline"]["mfma_products_per_fp32_product"] = r["mfma_products_per_fp32_product"]
        line["roofline"]["matrix_pipe_frac"] = _sig(r["matrix_pipe_frac"])
    if isinstance(r.get("traffic_source"), str):
        line["roofline"]["traffic_measured_in_this_run"] = r["traffic_source"].startswith("measured in this run")
    c = out.get("cpu_baseline")
    if c:
        rp = c.get("reference_python") or {}
        line["cpu_baseline"] = {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "cpu_model": c.get("cpu_model"), "kind": c["kind"],
                                "sample": c["sample"][:220],
                                "one_game_per_thread": _sig((c.get("one_game_per_thread") or {}).get("value")), "one_thread": _sig(c.get("value_1_thread")),
                                "reference_python": {"value": rp.get("value"), "cores": rp.get("cores"), "kind": rp.get("kind"),
                                                     "where": "build container (the reference never travels)"} if "value" in rp else None}
    for k in ("games_per_s", "sims_per_s", "games_completed", "expansions", "simulations", "pooled_records", "gather_ms"):
        line[k] = _sig(out[k]) if isinstance(out.get(k), float) else out.get(k)
    line["per_rank_records"] = out["per_rank_records"]
    line["per_rank_ms_per_step"] = [_sig(x, 5) for x in out["per_rank"]["ms_per_step"]]
    line["leaves_evaluated_rank0"] = out["leaves_evaluated_rank0"]
    fe = out["flop_per_expansion"]
    line["flop_per_expansion"] = {"reference_network": fe["reference_network"], "executed": fe["executed"]}
    line["whole_path_frac"] = _sig(out["whole_path"]["frac_executed_flop"])
    line["whole_path_frac_reference_flop"] = _sig(out["whole_path"]["frac_reference_flop"])
    if out["roofline"]["peak"] != PEAK_F32_MATRIX_TFLOPS:      # a split mode on the 16-bit pipes: the same rate against the fp32 matrix roof it is there to beat
        line["whole_path_vs_fp32_matrix_peak"] = _sig(out["whole_path"]["frac_executed_flop"] * out["roofline"]["peak"] / PEAK_F32_MATRIX_TFLOPS)
    line["whole_path_note"] = "value x FLOP / roofline.peak; reference-FLOP figure may exceed 1 by construction (conv1 + conv2 are exact table lookups, no FLOP executed)"
    # the other precisions of the same workload, same run (value_<mode>, roofline_<mode>{frac, avg_launch_ms, peak, achieved}, dtype_<mode>)
    for p in PRECISIONS:
        if ("value_" + p) in out:
            line["value_" + p] = _sig(out["value_" + p])
            line["games_per_s_" + p] = _sig(out["games_per_s_" + p])
            line["roofline_" + p] = {k: _sig(v) for k, v in out["roofline_" + p].items()}
            line["dtype_" + p] = out["dtype_" + p]
        if ("parity_max_err_" + p) in out:
            line["parity_max_err_" + p] = _sig(out["parity_max_err_" + p], 3)
    if "parity_sample_max_err" in out:
        line["parity_sample_max_err"] = _sig(out["parity_sample_max_err"], 3)
        line["parity_tolerance"] = 1e-5
    cal = out.get("device_calibration") or {}
    if "f16" in cal:
        line["device_calibration"] = {"f16_sustained_tflops": _sig(cal["f16"]["sustained_tflops"], 4), "f32_sustained_tflops": _sig(cal["f32"]["sustained_tflops"], 4),
                                      "bf16_sustained_tflops": _sig((cal.get("bf16") or {}).get("sustained_tflops"), 4),
                                      "dominant_kernel_share_of_sustained": _sig(cal["dominant_kernel_share_of_sustained"], 4)}
    if "config4" in out:
        line["config4_value"] = _sig(out["config4"]["value"])
        line["config4_games_per_s"] = _sig(out["config4"]["games_per_s"])
        line["config4_roofline_frac"] = _sig(out["config4"]["roofline"]["frac"])
    if "config5" in out:
        c5 = out["config5"]
        line["config5_games_per_s"] = _sig(c5.get("games_per_s"))
        line["config5_sample_mismatches"] = c5.get("sample_mismatches")
        line["config5_precision"] = c5.get("precision")
        line["config5_us_per_sim_step"] = _sig(c5.get("us_per_sim_step"), 4)
        for p, leg in (c5.get("other_precisions") or {}).items():
            if "us_per_sim_step" in leg:
                line["config5_us_per_sim_step_" + p] = _sig(leg["us_per_sim_step"], 4)
    if "dropin_config0" in out:
        line["dropin_config0_s"] = _sig(out["dropin_config0"]["gpu_dropin"]["seconds"], 4)
    for k, short in (("cross_game_dedup", "value_with_cross_game_dedup"), ("eval_cache", "value_with_eval_cache"), ("other_driver", "value_other_driver"),
                     ("all_layers_as_gemm", "value_all_layers_as_gemm")):
        if k in out:
            line[short] = _sig(out[k]["value"])
    line["tree_side_hbm_frac"] = _sig(out["tree_side_hbm"]["frac"], 3)
    line["detail"] = out.get("detail_file")
    line = _finite(line)
    text = json.dumps(line, allow_nan=False)
    if len(text) > limit:                                        # cannot happen at N <= 8; never print a line the driver cannot keep whole
        for k in ("whole_path_note", "per_rank_ms_per_step", "flop_per_expansion", "device_calibration"):
            line.pop(k, None)
        line["cpu_baseline"] = {k: v for k, v in (line.get("cpu_baseline") or {}).items() if k != "sample"} or None
        text = json.dumps(line, allow_nan=False)
    return text


def run_secondary(ctx):
    """the secondary legs of the N = 1 line, after the timed region.  ctx: the timed region's objects (bench.py run_rank): args, out, wall, world,
    net, eng, make_engine, measure, advance, layer, n, G, cap_main, d (the timed region's counter deltas), dom_launches, flop_ref."""
    from othellozero_amd.NNet import NNetWrapper
    args, out, wall, world, net, eng = ctx["args"], ctx["out"], ctx["wall"], ctx["world"], ctx["net"], ctx["eng"]
    make_engine, measure, advance = ctx["make_engine"], ctx["measure"], ctx["advance"]
    layer, n, G, cap_main, d, dom_launches, flop_ref = ctx["layer"], ctx["n"], ctx["G"], ctx["cap_main"], ctx["d"], ctx["dom_launches"], ctx["flop_ref"]
    secondary = world == 1 and not args.no_compare
    cheap_pre = max(2, min(8, args.sims))                   # stagger of the secondary legs: 8 sims/move (a few tenths of a second)
    vsteps = max(args.steps // 2, 1)                        # the variation legs (de-duplication, cache, other driver, all-GEMM) run half the steps: rates, not headlines
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
        byts, how = live_traffic(args, layer, d["leaves_evaluated"] / max(dom_launches, 1))      # scaled to the leaves per launch of the timed region
        if byts is not None:
            out["roofline"]["traffic_from_committed_profile"] = out["roofline"]["traffic"]
            out["roofline"]["traffic"] = byts
            out["roofline"]["traffic_source"] = how
        else:
            out["roofline"]["live_traffic_error"] = how
        wall["live_traffic_s"] = round(time.perf_counter() - t_sec, 2)
    psample = None
    if secondary and not args.no_cpu_baseline:
        t_sec = time.perf_counter()
        ps, psample = parity_sample(net, eng, n, args.channels, cap_main or G)
        out["parity_sample_max_err"] = max(ps["max_abs_err_pi"], ps["max_abs_err_v"])
        out["parity_sample"] = ps
        out["parity_max_err_" + args.precision] = out["parity_sample_max_err"]
        wall["parity_sample_s"] = round(time.perf_counter() - t_sec, 2)
    ctx["eng"] = eng = None                                   # the timed engine's HBM is free for the legs below (N > 1 never gets here with legs)
    if secondary and args.dedup == "off":
        t_sec = time.perf_counter()
        # the same workload with the library default (cross-game de-duplication on): identical records, fewer evaluations
        eng2 = make_engine(True)
        eng2.stagger(cheap_pre)
        q, dt2 = measure(eng2, vsteps)
        out["cross_game_dedup"] = {
            "value": q["expansions"] / dt2, "unit": "node-expansions/s", "ms_per_step": dt2 / vsteps * 1e3, "steps": vsteps,
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
        qc, dtc = measure(engc, vsteps)
        c1 = net.eval_cache_stats()
        out["eval_cache"] = {
            "value": qc["expansions"] / dtc, "unit": "node-expansions/s", "ms_per_step": dtc / vsteps * 1e3, "steps": vsteps,
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
        qo, dto = measure(engo, vsteps, driver=other)
        out["other_driver"] = {
            "driver": other, "value": qo["expansions"] / dto, "unit": "node-expansions/s", "ms_per_step": dto / vsteps * 1e3, "steps": vsteps,
            "games_per_s": qo["games_completed"] / dto, "sims_per_s": qo["simulations"] / dto,
            "leaves_per_batch": qo["leaves_evaluated"] / (vsteps * args.sims),
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
    if secondary:
        # ---- the same workload, same run, same number of steps, in the library's OTHER precisions.  The line's top-level value is measured in
        # args.precision (default f32: what the reference computes in, Net/NNet.py:85); every other mode rides beside it (value_<mode>, roofline_<mode>)
        out["precisions"] = {}
        for prec in [p for p in PRECISIONS if p != args.precision]:
            t_sec = time.perf_counter()
            try:
                netp = NNetWrapper((n, n), num_channels_1=args.channels, max_batch=G, seed=0, precision=prec)
            except Exception as e:                                   # noqa: BLE001 -- a secondary leg never costs the line
                out["precisions"][prec] = {"error": repr(e)}
                continue
            ep = make_engine(args.dedup == "on", the_net=netp)
            ep.stagger(cheap_pre)
            qp, dtp = measure(ep, args.steps, netp)
            msp, lp = netp.profile_read()
            layerp = netp.profiled_layer()
            achp = qp["leaves_evaluated"] * conv_flop_per_leaf(layerp, n, args.channels) / max(msp * 1e-3, 1e-9) / 1e12
            leg = {"value": qp["expansions"] / dtp, "unit": "node-expansions/s", "ms_per_step": dtp / args.steps * 1e3, "steps": args.steps,
                   "games_per_s": qp["games_completed"] / dtp, "sims_per_s": qp["simulations"] / dtp, "dtype": DTYPE_LABEL[prec],
                   "dtype_detail": DTYPE_DETAIL[prec],
                   "roofline": roofline(prec, layerp, achp, msp, lp, qp["leaves_evaluated"], n, args.channels,
                                        conv3_rows=netp.conv3_tile_rows() if prec != "f32" else 256),
                   "note": f"same workload and run as the headline (slots staggered at {cheap_pre} sims/move), precision='{prec}'"}
            if psample is not None:
                pp_ = parity_rows(netp, psample)
                leg["parity_sample"] = pp_
                out["parity_max_err_" + prec] = max(pp_["max_abs_err_pi"], pp_["max_abs_err_v"])
            del ep, netp
            if not args.no_live_traffic and prec != "f16x2":            # (two more PMC child runs per mode: the exact-fp32 and bf16x3 legs get them)
                byts, how = live_traffic(args, layerp, qp["leaves_evaluated"] / max(lp, 1), precision=prec)
                rp = leg["roofline"]
                if byts is not None:
                    rp["traffic_from_committed_profile"] = rp["traffic"]
                    rp["traffic"] = byts
                    rp["traffic_source"] = how
                else:
                    rp["live_traffic_error"] = how
            out["precisions"][prec] = leg
            out["value_" + prec] = leg["value"]
            out["games_per_s_" + prec] = leg["games_per_s"]
            out["ms_per_step_" + prec] = leg["ms_per_step"]
            out["roofline_" + prec] = {"frac": leg["roofline"]["frac"], "avg_launch_ms": leg["roofline"]["avg_launch_ms"], "peak": leg["roofline"]["peak"],
                                       "achieved": leg["roofline"]["achieved"]}
            out["dtype_" + prec] = leg["dtype"]
            wall[prec + "_s"] = round(time.perf_counter() - t_sec, 2)
    if secondary and n == 8:
        # ---- BASELINE configs[3]: 6x6 boards, same network family, same engine -- with its own roofline (dominant launch: conv3 of the 6x6
        # network on the 256-row tile), kernels[], parity sample (both precisions, same rows) and exact-fp32 rate
        t_sec = time.perf_counter()
        cap6 = ctx["batch_cap"](6, G) if "batch_cap" in ctx else 0
        net6 = NNetWrapper((6, 6), num_channels_1=args.channels, max_batch=G, seed=0, precision=args.precision)
        e6 = make_engine(args.dedup == "on", the_net=net6, board=6)
        e6.stagger(cheap_pre)
        q6, dt6 = measure(e6, args.steps, net6)
        ms6, l6 = net6.profile_read()
        layer6 = net6.profiled_layer()
        ach6 = q6["leaves_evaluated"] * conv_flop_per_leaf(layer6, 6, args.channels) / max(ms6 * 1e-3, 1e-9) / 1e12
        roof6 = roofline(args.precision, layer6, ach6, ms6, l6, q6["leaves_evaluated"], 6, args.channels, conv3_rows=conv3_tile_rows(6, cap6 or G, args.channels))
        roof6["traffic"], roof6["traffic_source"] = None, "not measured for the 6x6 network"
        roof6["leaves_per_launch"] = q6["leaves_evaluated"] / max(l6, 1)
        out["config4"] = {
            "workload": f"{G} concurrent 6x6 self-play games, {args.sims} sims/move (BASELINE configs[3])",
            "value": q6["expansions"] / dt6, "unit": "node-expansions/s", "ms_per_step": dt6 / args.steps * 1e3,
            "games_per_s": q6["games_completed"] / dt6, "sims_per_s": q6["simulations"] / dt6,
            "flop_per_expansion_reference": FLOP_PER_EXPANSION[6], "precision": args.precision, "roofline": roof6}
        rounds6 = 2                                             # every kernel of a 6x6 step against its own roof, as for the headline
        e6.profile(True); net6.profile(2)
        e6.profile_read(reset=True); net6.profile_kernels(reset=True)
        a6 = e6.stats()
        advance(e6, rounds6, True)
        b6 = e6.stats()
        out["config4"]["kernels"] = kernel_table(net6.profile_kernels(), e6.profile_read(), rounds6, b6["leaves_evaluated"] - a6["leaves_evaluated"],
                                                 b6["simulations"] - a6["simulations"], 6, args.channels, args.precision, layer6 == 3, args.driver)
        e6.profile(False); net6.profile(0)
        ps6 = None
        if not args.no_cpu_baseline:
            rep6, ps6 = parity_sample(net6, e6, 6, args.channels, G, check=256)
            out["config4"]["parity_sample"] = rep6
        del e6, net6
        out["config4"]["precisions"] = {}
        for prec in [p for p in PRECISIONS if p != args.precision]:
            try:
                np_ = NNetWrapper((6, 6), num_channels_1=args.channels, max_batch=G, seed=0, precision=prec)
            except Exception as e:                                   # noqa: BLE001
                out["config4"]["precisions"][prec] = {"error": repr(e)}
                continue
            e6p = make_engine(args.dedup == "on", the_net=np_, board=6)
            e6p.stagger(cheap_pre)
            steps6p = max(args.steps // 2, 1)
            q6p, dt6p = measure(e6p, steps6p, np_)
            ms6p, l6p = np_.profile_read()
            lay6p = np_.profiled_layer()
            a6p = q6p["leaves_evaluated"] * conv_flop_per_leaf(lay6p, 6, args.channels) / max(ms6p * 1e-3, 1e-9) / 1e12
            r6p = roofline(prec, lay6p, a6p, ms6p, l6p, q6p["leaves_evaluated"], 6, args.channels, conv3_rows=np_.conv3_tile_rows() if prec != "f32" else 256)
            r6p["traffic"], r6p["traffic_source"] = None, "not measured for the 6x6 network"
            out["config4"]["precisions"][prec] = {"value": q6p["expansions"] / dt6p, "unit": "node-expansions/s", "ms_per_step": dt6p / steps6p * 1e3,
                                                  "steps": steps6p, "games_per_s": q6p["games_completed"] / dt6p, "dtype": DTYPE_LABEL[prec], "roofline": r6p,
                                                  "conv3_tile_rows": np_.conv3_tile_rows()}
            if ps6 is not None:
                out["config4"]["precisions"][prec]["parity_sample"] = parity_rows(np_, ps6)
            del e6p, np_
        wall["config4_s"] = round(time.perf_counter() - t_sec, 2)
    if secondary and n == 8 and not args.no_cpu_baseline:
        # ---- BASELINE configs[4]: arena evaluation with two real networks (bounded plies), sampled games replayed by the oracle
        t_sec = time.perf_counter()
        out["config5"] = config5_arena(args.channels, args.arena_precision, args.arena_plies)
        if "games_per_s" in out["config5"]:
            out["config5_games_per_s"] = out["config5"]["games_per_s"]
            out["config5_sample_mismatches"] = out["config5"]["sample_mismatches"]
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
