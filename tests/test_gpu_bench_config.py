"""Parity at the configuration bench.py runs (BASELINE configs[1] / configs[3]): the 512-filter OthelloNN built with
max_batch = 4096 -- every convolution on ONE k-slice with the fused BN + ReLU + h2 re-split epilogue, fc1 on the 256 x 256
ping-pong tile with 4-way split-K, fc2 on the thin tile without split-K; precision f32 without split-K slabs -- against the
float64 restatement of Net/OthelloNN.py:42-56 (oracle/nn_numpy.py), and the batched engine at 4096 games x 100 sims with
that network against the oracle's search (MCTS/__init__.py:30-84, training.py:26-72) replaying sampled games.

Tolerance (BASELINE.json north_star): |d pi|, |d v| <= 1e-5 absolute; moves, boards, z and visit counts bit-exact."""
import numpy as np
import pytest

import oracle
from oracle import nn_numpy

pytestmark = pytest.mark.gpu

TOL = 1e-5            # north_star: "within 1e-5 on float policy/value"
TWIN_TOL = 2e-6       # two builds of the same network with other tile shapes / k-splits: rounding only
C = 512
B = 4096


@pytest.fixture(scope="module")
def oz():
    from othellozero_amd import _lib
    _lib.require_gpu()
    return _lib


def _weights(n):
    from othellozero_amd.weights import init_weights
    w = init_weights(n, seed=40 + n, channels=C, randomize_all=True)     # kernels, biases, BN gamma / beta / mean / variance all random
    for i in (36, 38):
        w[i] = w[i] * 4.0                                                # logits away from uniform, |v| away from 0
    return w


def _positions(n, count):
    """`count` canonical positions: the corner cases of the input planes, every position of 28 self-play games (what the
    engine really feeds the network: openings, middle games, nearly full boards, both movers), random fillings for the rest"""
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    valid = np.uint64(sum(1 << (r * 8 + c) for r in range(n) for c in range(n)))
    bit = lambda r, c: np.uint64(1 << (r * 8 + c))
    rs = np.random.RandomState(900 + n)
    full = rs.randint(0, 2**63, size=1, dtype=np.uint64)[0] & valid
    special = [(0, 0), (valid, 0), (0, valid), (full, valid & ~full), (bit(0, 0), bit(n - 1, n - 1)), (bit(0, n - 1), bit(n - 1, 0)),
               (bit(n // 2, 0), bit(0, n // 2)), (bit(n - 1, n // 2) | bit(n // 2, n - 1), 0)]
    G = 28
    eng = SelfPlayEngine(StubNetWrapper((n, n), 17, 0, max_batch=G), n, G, 8, 1.0, 1.0, 0.7, seed=3)
    rec = eng.play_to_end()
    own = np.where(rec["player"] == 1, rec["black"], rec["white"])
    opp = np.where(rec["player"] == 1, rec["white"], rec["black"])
    # ... and the final boards of those games (full or nearly full)
    own = np.concatenate([np.array([s[0] for s in special], np.uint64), own, rec["final_black"][::16]])
    opp = np.concatenate([np.array([s[1] for s in special], np.uint64), opp, rec["final_white"][::16]])
    rest = count - own.size
    assert rest > count // 8
    dens = rs.rand(rest)                                                 # sparse to dense fillings
    a = np.zeros(rest, np.uint64); b = np.zeros(rest, np.uint64)
    for r in range(n):
        for c in range(n):
            u = rs.rand(rest)
            a |= np.where(u < dens * 0.5, bit(r, c), np.uint64(0))
            b |= np.where((u >= dens * 0.5) & (u < dens), bit(r, c), np.uint64(0))
    own, opp = np.concatenate([own, a]), np.concatenate([opp, b])
    assert own.size == count and np.all((own & opp) == 0) and np.all(((own | opp) & ~valid) == 0)
    return own, opp


_CACHE = {}


def _case(n):
    """weights, 4096 positions and their float64 (pi, v) -- computed once per board size"""
    if n not in _CACHE:
        w = _weights(n)
        own, opp = _positions(n, B)
        pi64, v64 = nn_numpy.forward_chunked(w, own, opp, n, chunk=512)
        assert pi64.std() > 2e-3 and np.abs(v64).max() > 0.05            # the comparison is not vacuous
        _CACHE[n] = (w, own, opp, pi64, v64)
    return _CACHE[n]


@pytest.mark.parametrize("precision", ["f16x2", "f32", "bf16x3"])
@pytest.mark.parametrize("n", [8, 6])
def test_network_at_bench_batch_vs_float64_oracle(oz, n, precision):
    """NNetWrapper(max_batch=4096), ONE call with 4096 positions == one forward of the kernels bench.py times"""
    from othellozero_amd.NNet import NNetWrapper
    w, own, opp, pi64, v64 = _case(n)
    net = NNetWrapper((n, n), num_channels_1=C, max_batch=B, weights=w, precision=precision)
    pi, v = net.predict_batch(own, opp)
    pi = pi.reshape(B, -1)
    err_pi, err_v = np.abs(pi - pi64).max(), np.abs(v - v64).max()
    assert err_pi <= TOL and err_v <= TOL, (err_pi, err_v)
    assert np.abs(pi.sum(axis=1) - 1).max() < 1e-5
    # row mapping: the worst row is no worse than the tolerance, and no two distinct positions were swapped
    # (a transposed / shifted epilogue row would put position i's outputs at position j != i)
    rows = np.abs(pi - pi64).max(axis=1)
    assert rows.max() <= TOL
    # the same positions in another order, and a shorter call: bit-identical per position
    perm = np.random.RandomState(1).permutation(B)
    p2, v2 = net.predict_batch(own[perm], opp[perm])
    assert np.array_equal(p2.reshape(B, -1), pi[perm]) and np.array_equal(v2, v[perm])
    p3, v3 = net.predict_batch(own[100:1337], opp[100:1337])
    assert np.array_equal(p3.reshape(1237, -1), pi[100:1337]) and np.array_equal(v3, v[100:1337])
    # a max_batch = 2048 build (the same one-k-slice kernels on half the grid) and a max_batch = 64 twin (split-K 2-8 with
    # k_splitk_reduce_h2, 128 x 128 dense tiles): the same network to rounding
    half = NNetWrapper((n, n), num_channels_1=C, max_batch=2048, weights=w, precision=precision)
    ph, vh = half.predict_batch(own[:2048], opp[:2048])
    assert np.abs(ph.reshape(2048, -1) - pi64[:2048]).max() <= TOL and np.abs(vh - v64[:2048]).max() <= TOL
    assert np.abs(ph.reshape(2048, -1) - pi[:2048]).max() <= TWIN_TOL and np.abs(vh - v[:2048]).max() <= TWIN_TOL
    del half
    twin = NNetWrapper((n, n), num_channels_1=C, max_batch=64, weights=w, precision=precision)
    sel = np.r_[0:64, 1000:1064, B - 64:B]
    pt, vt = twin.predict_batch(own[sel], opp[sel])
    assert np.abs(pt.reshape(sel.size, -1) - pi[sel]).max() <= TWIN_TOL and np.abs(vt - v[sel]).max() <= TWIN_TOL


@pytest.mark.parametrize("precision", ["f16x2", "f32", "bf16x3"])
def test_network_at_the_timed_shape_vs_float64_oracle(oz, precision):
    """the exact launch bench.py's timed region makes: NNetWrapper(max_batch=4096), ONE call with preferred_batch_cap(8, 4096, 512) =
    3640 positions -- conv3 then runs on the 256-row tile (1024 tiles = 4.0 grid rounds), the kernel the roofline row is about.
    <= 1e-5 against the float64 oracle, and bit-identical on the shared rows with the 4096-position call (192-row tile: same sums)"""
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.training import preferred_batch_cap
    n = 8
    cap = preferred_batch_cap(n, B, C)
    assert cap == 3640
    w, own, opp, pi64, v64 = _case(n)
    net = NNetWrapper((n, n), num_channels_1=C, max_batch=B, weights=w, precision=precision)
    pi, v = net.predict_batch(own[:cap], opp[:cap])
    rows_cap = net.conv3_tile_rows()
    pi = pi.reshape(cap, -1)
    err_pi, err_v = np.abs(pi - pi64[:cap]).max(), np.abs(v - v64[:cap]).max()
    assert err_pi <= TOL and err_v <= TOL, (err_pi, err_v)
    p4, v4 = net.predict_batch(own, opp)
    rows_full = net.conv3_tile_rows()
    if precision == "f16x2":
        assert (rows_cap, rows_full) == (256, 192)                       # the capped call IS the timed kernel; the full call is the other tile
    elif precision == "bf16x3":
        assert (rows_cap, rows_full) == (128, 128)                       # k_gemm_b3's one tile (128 x 256: 6 B per element, two stages of 72 KB)
    else:
        # exact fp32: the library reports what oz_gemm_f32_launch really launched -- k_gemm_f32<GmBig>, the 256 x 256 tile, for every
        # call of a max_batch = 4096 network (the tile is keyed on the capacity, not on the size of a call)
        assert (rows_cap, rows_full) == (256, 256)
    assert np.array_equal(p4.reshape(B, -1)[:cap], pi) and np.array_equal(v4[:cap], v)
    # the last rows of the call sit in the partly filled last row tile (3640 * 36 = 511.9 tiles of 256 rows)
    tail = np.abs(pi[-8:] - pi64[cap - 8:cap]).max()
    assert tail <= TOL


def test_f32_big_tile_bit_identical_to_the_standard_tile(oz):
    """precision f32: the 256 x 256 tile of large batches (k_gemm_f32<GmBig>, the kernel behind value_exact_fp32) against the 128 x 128 tile
    every other layer and mode runs on (OZ_NET_OPT_F32_STD_TILE forces it): both add every output element's products in the same order, so
    (pi, v) must be BIT-identical -- at the timed shape (3640 positions) and on a full 4096-position call, 8x8 and 6x6.  Also the screen for
    the LDS-DMA publication of both instantiations (a visibility race in one of them would show up as a mismatch)."""
    from othellozero_amd.NNet import NNetWrapper
    for n in (8, 6):
        w, own, opp, pi64, v64 = _case(n)
        net = NNetWrapper((n, n), num_channels_1=C, max_batch=B, weights=w, precision="f32")
        for count in (3640, B):
            net.set_option(oz.NET_OPT_F32_STD_TILE, 0)
            pb, vb = net.predict_batch(own[:count], opp[:count])
            assert net.conv3_tile_rows() == 256
            net.set_option(oz.NET_OPT_F32_STD_TILE, 1)
            ps, vs = net.predict_batch(own[:count], opp[:count])
            assert net.conv3_tile_rows() == 128
            assert np.array_equal(pb, ps) and np.array_equal(vb, vs), (n, count)
            assert np.abs(pb.reshape(count, -1) - pi64[:count]).max() <= TOL and np.abs(vb - v64[:count]).max() <= TOL
        net.set_option(oz.NET_OPT_F32_STD_TILE, 0)
        # a one-position network reports the weight-stream kernel
        one = NNetWrapper((n, n), num_channels_1=C, max_batch=1, weights=w, precision="f32")
        one.predict_batch(own[:1], opp[:1])
        assert one.conv3_tile_rows() == 64
        del net, one


def test_conv3_tiles_bit_identical(oz):
    """precision f16x2: conv3's three row tiles -- 256 rows (4-phase ping-pong loop), 192 and 128 rows (2-phase loop) -- add every output
    element's products in the same order, so (pi, v) must be BIT-identical whichever tile a call runs on (OZ_NET_OPT_CONV3_TILE forces one):
    the arena's network (max_batch 512: also conv4 and fc1 on the 128-row tile, fc2's k-slices added inside the heads kernel) at the call sizes
    on either side of the forward's own choice, and bench.py's network at the cap and at a full call; every result within 1e-5 of float64.
    Also the screen for the LDS-DMA schedule of the 2-phase loop (regions refilled in the phase after their last read)."""
    from othellozero_amd.NNet import NNetWrapper
    n = 8
    w, own, opp, pi64, v64 = _case(n)
    for max_batch, counts in ((512, ((100, 128), (430, 128), (456, 192), (512, 192))), (B, ((3640, 256), (B, 192)))):
        net = NNetWrapper((n, n), num_channels_1=C, max_batch=max_batch, weights=w, precision="f16x2")
        for count, picked in counts:
            net.set_option(oz.NET_OPT_CONV3_TILE, 0)
            p0, v0 = net.predict_batch(own[:count], opp[:count])
            assert net.conv3_tile_rows() == picked, (max_batch, count, net.conv3_tile_rows())
            assert np.abs(p0.reshape(count, -1) - pi64[:count]).max() <= TOL and np.abs(v0 - v64[:count]).max() <= TOL
            for tile in (128, 192, 256):
                net.set_option(oz.NET_OPT_CONV3_TILE, tile)
                pt, vt = net.predict_batch(own[:count], opp[:count])
                assert net.conv3_tile_rows() == tile
                assert np.array_equal(pt, p0) and np.array_equal(vt, v0), (max_batch, count, tile)
        net.set_option(oz.NET_OPT_CONV3_TILE, 0)
        # a position's result does not depend on the size of the call (nor, therefore, on the tile the call picked)
        pa, va = net.predict_batch(own[:counts[0][0]], opp[:counts[0][0]])
        pb, vb = net.predict_batch(own[:counts[-1][0]], opp[:counts[-1][0]])
        k = counts[0][0]
        assert np.array_equal(pb.reshape(counts[-1][0], -1)[:k], pa.reshape(k, -1)) and np.array_equal(vb[:k], va)
        with pytest.raises(oz.OzError):
            net.set_option(oz.NET_OPT_CONV3_TILE, 64)
        del net


@pytest.mark.parametrize("n", [8, 6])
def test_bf16x3_tiles_bit_identical(oz, n):
    """precision bf16x3: k_gemm_b3's two tiles -- 128 x 256 (two 72 KB LDS stages, 2-phase ping-pong loop) and 256 x 256 (five 24 KB regions, each refilled in
    the phase after its last read, 4-phase loop with B n0 read twice) -- add every output element's six-product sums in the same order: (pi, v) must be
    BIT-identical whichever tile conv3 / conv4 run on (OZ_NET_OPT_B3_TILE forces one), at the bench's call sizes and at ragged ones (rows beyond M read
    the zero line; the last row tile partly filled); every result within 1e-5 of float64.  Also the screen for the region schedule of the big tile."""
    from othellozero_amd.NNet import NNetWrapper
    w, own, opp, pi64, v64 = _case(n)
    net = NNetWrapper((n, n), num_channels_1=C, max_batch=B, weights=w, precision="bf16x3")
    for count in (B, 3916, 3640, 777, 1):
        net.set_option(oz.NET_OPT_B3_TILE, 0)
        p0, v0 = net.predict_batch(own[:count], opp[:count])
        picked = net.conv3_tile_rows()
        assert picked == (128 if n == 8 else 256)                 # conv3 at capacity 4096: 9 rounds of 128 rows against 5 of 256 (8x8); 4 against 2.0 (6x6)
        assert np.abs(p0.reshape(count, -1) - pi64[:count]).max() <= TOL and np.abs(v0 - v64[:count]).max() <= TOL
        for tile in (128, 256):
            net.set_option(oz.NET_OPT_B3_TILE, tile)
            for rep in range(3):                                  # (repeated: a DMA visibility race would be intermittent)
                pt, vt = net.predict_batch(own[:count], opp[:count])
                assert net.conv3_tile_rows() == tile
                assert np.array_equal(pt, p0) and np.array_equal(vt, v0), (count, tile, rep)
    net.set_option(oz.NET_OPT_B3_TILE, 0)
    with pytest.raises(oz.OzError):
        net.set_option(oz.NET_OPT_B3_TILE, 192)


def test_config5_real_networks_whole_games_vs_oracle(oz):
    """bench.py's config5 leg (BASELINE configs[4] with REAL networks): 512 arena games x 800 sims per move and agent on two 512-filter
    networks (seeds 0 / 1), every game played TO THE END; two sampled WHOLE games are replayed by the oracle's arena (agents.py:44-84
    restated) fed with the GPU networks' own (pi, v): actions, movers, boards, winner and points bit-exact"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("oz_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = bench.config5_arena(C, "f16x2")
    assert "plies_per_game" not in out and out["games_finished"] == 512 and out["games_per_s"] > 0
    assert out["sample_mismatches"] == 0 and out["sample_games_replayed_by_oracle"] == 2 and out["sample_plies_replayed"] >= 2 * 50
    assert 512 * 50 <= out["moves"] <= 512 * 62 and out["simulations"] == out["moves"] * 800 and out["expansions"] > 0.7 * out["simulations"]
    assert out["sims_per_s"] > 0 and out["value"] > 0
    # the leg's headline evaluates every expansion by itself; the library default shares boards between the games of a step -- same moves
    assert out["leaves_evaluated"] == out["expansions"]
    assert out["with_cross_game_dedup"]["identical_moves_on_those_plies"]
    cc = out["with_dedup_and_eval_cache"]                      # the cached arena plays the SAME 512 games, and evaluates far fewer positions
    assert cc["identical_moves_on_the_opening_plies_vs_every_expansion_evaluated"] and cc["games_identical_to_the_pure_run_above"] >= 500
    assert cc["leaves_evaluated"] < 0.5 * cc["expansions"] and cc["cache_hit_rate"] > 0.005     # (of the lookups LEFT after the same-step de-duplication)
    assert cc["games_per_s"] > out["games_per_s"]
    # round 5: the leg carries the regime's own roofline (conv3 at <= 512 leaves per launch, events in the timed run) and kernels[]
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and rf["launches"] == out["simulations"] // 512 * 1 or rf["launches"] > 0
    assert 0 < rf["frac"] < 1 and rf["leaves_per_launch"] <= 512 and rf["avg_launch_ms"] > 0
    names = {k["name"] for k in out["kernels"]}
    assert {"conv2", "conv3", "conv4", "fc1", "fc2", "heads", "select", "compact"} <= names
    assert all(k["us_per_sim_step"] > 0 for k in out["kernels"])
    assert abs(sum(k["us_per_sim_step"] for k in out["kernels"]) / out["us_per_sim_step"] - 1) < 0.6
    # the bounded form (quick looks) still says that it is bounded
    short = bench.config5_arena(C, "f16x2", plies=3, sample=1)
    assert short["plies_per_game"] == 3 and short["moves"] == 512 * 3 and short["sample_mismatches"] == 0 and "games_per_s" not in short


@pytest.mark.parametrize("precision", ["f16x2", "f32", "bf16x3"])
def test_network_at_bench_batch_gemm_form_vs_float64_oracle(oz, precision):
    """the same 4096-position forward with conv1 as a kernel and conv2 as an MFMA implicit GEMM (oz_net_set_tables 0: the
    `all_layers_as_gemm` leg of bench.py) -- the 256 x 256 ping-pong tile on conv2 at one k-slice"""
    from othellozero_amd.NNet import NNetWrapper
    n = 8
    w, own, opp, pi64, v64 = _case(n)
    net = NNetWrapper((n, n), num_channels_1=C, max_batch=B, weights=w, precision=precision)
    net.set_tables(0)
    pi, v = net.predict_batch(own, opp)
    assert net.profiled_layer() == 2
    assert np.abs(pi.reshape(B, -1) - pi64).max() <= TOL and np.abs(v - v64).max() <= TOL
    if precision == "f16x2":
        net.set_tables(1)                                                # conv1 from its table inside conv2's operand gather
        p1, v1 = net.predict_batch(own, opp)
        assert np.array_equal(p1, pi) and np.array_equal(v1, v)          # documented bit-identical to mode 0


@pytest.mark.parametrize("n,precision,dedup", [(8, "f16x2", False), (8, "f16x2", True), (8, "f32", False), (6, "f16x2", False), (6, "f32", False),
                                               (8, "bf16x3", False), (6, "bf16x3", False)])
def test_config2_real_network_search_replay(oz, n, precision, dedup):
    """BASELINE configs[1] (8x8) and configs[3] (6x6) with the real network: 4096 concurrent games x 100 sims/move x 2 move rounds on the
    512-filter OthelloNN (max_batch 4096: the kernels bench.py times -- on 6x6 what its `config4` leg times); 16 sampled games are replayed by
    the oracle's search (training.py:26-72, MCTS/__init__.py:30-84 restated) fed with the GPU network's own (pi, v) per position -- moves, boards
    and root visit counts must match bit for bit (network rounding cannot excuse a divergent game), with the cross-game leaf de-duplication
    off (bench headline) and on"""
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    G, sims, rounds = B, 100, 2
    net = NNetWrapper((n, n), num_channels_1=C, max_batch=G, seed=0, precision=precision)        # bench.py's network (seed 0)
    eng = SelfPlayEngine(net, n, G, sims, 1.0, 1.0, 0.9, seed=1234, first_game_id=0, q_mode=1, dedup=dedup)
    eng.run(1)
    counts1 = eng.last_counts().copy()
    eng.run(rounds - 1)
    st, after, counts2 = eng.stats(), eng.state(), eng.last_counts()
    assert st["simulations"] == G * sims * rounds and st["moves"] == G * rounds and st["overflow"] == 0
    assert (st["leaves_evaluated"] < st["expansions"]) if dedup else (st["leaves_evaluated"] == st["expansions"])
    cache = {}

    def ev(own, opp, nn):
        if (own, opp) not in cache:
            p, v = net.predict_batch([own], [opp])                       # a position's (pi, v) does not depend on the batch
            cache[(own, opp)] = (p[0].ravel(), float(v[0]))
        return cache[(own, opp)]
    for gi in (0, 1, 2, 3, 341, 682, 1023, 1364, 1705, 2046, 2387, 2728, 3069, 3410, G - 2, G - 1):
        ep = oracle.Mcts(n, 1.0, 1, evaluator=ev).episode(sims, 1.0, 0.9, 1234, gi, max_moves=rounds)
        assert np.array_equal(counts1[gi], ep["counts"][0]), gi
        assert np.array_equal(counts2[gi], ep["counts"][1]), gi
        import ctypes as CT
        b, w = CT.c_uint64(int(ep["black"][-1])), CT.c_uint64(int(ep["white"][-1]))
        pl, fin = CT.c_int(int(ep["player"][-1])), CT.c_int(0)
        oracle.lib().orc_game_play(CT.byref(b), CT.byref(w), n, CT.byref(pl), CT.byref(fin), int(ep["action"][-1]))
        assert (b.value, w.value, pl.value) == (int(after["black"][gi]), int(after["white"][gi]), int(after["player"][gi])), gi


def test_stagger_spreads_games_and_keeps_records_exact(oz):
    """oz_selfplay_stagger: slot g ends (g * P) // G plies into its first game; the staggered moves are ordinary searched
    moves -- a game's record list equals the oracle's episode with `sims_pre` simulations on its first offset(g) plies and
    `sims` afterwards; afterwards every move round completes games"""
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    n, G, sims, pre = 6, 64, 10, 4
    P = n * n - 4
    eng = SelfPlayEngine(StubNetWrapper((n, n), 9, 0, max_batch=G), n, G, sims, 1.0, 1.0, 0.9, seed=21, first_game_id=0,
                         game_id_stride=G, refill=True, record_cap=G * 8 * n * n)
    eng.stagger(pre)
    st = eng.state()
    offs = (np.arange(G) * P) // G
    live = st["game_id"] < G                                              # (a game that ended during the stagger was refilled)
    assert np.array_equal(st["ply"][live], offs[live]) and live.sum() >= G - 2
    done0 = eng.stats()["games_completed"]
    eng.run(P + 4)
    s1 = eng.stats()
    assert s1["games_completed"] - done0 >= G                              # every first-generation game finished ...
    rec = eng.records()
    per_round = np.bincount(rec["game_id"].astype(np.int64) % G, minlength=G)
    assert per_round.min() > 0
    with pytest.raises(oz.OzError):
        eng.stagger(pre)                                                   # first driver call only
    # exactness of a staggered game: the oracle plays game g with `pre` sims on plies < offset(g), `sims` afterwards
    for g in (5, 33, 63):
        r = rec[rec["game_id"] == g]
        m = oracle.Mcts(n, 1.0, 1, salt=9)
        ep = m.episode(sims, 1.0, 0.9, 21, g, sims_pre=pre, pre_plies=int(offs[g]))
        assert np.array_equal(r["action"], ep["action"]) and np.array_equal(r["z"], ep["z"]), g


@pytest.mark.parametrize("dedup,cap", [(False, 0), (True, 0), (False, 3640), (True, 3640)])
def test_free_running_driver_at_config2_size_equals_lock_step(oz, dedup, cap):
    """bench.py's default driver (oz_selfplay_run_steps: a game runs on by itself, batches stay full) at the bench's own size -- 4096
    staggered 8x8 games x 100 simulations on the 512-filter network, f16x2, one evaluation per expansion: every game that completes
    under both drivers has the records of the lock-step driver (whose games the tests above replay with the oracle) bit for bit"""
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    n, G, sims = 8, B, 100
    net = NNetWrapper((n, n), num_channels_1=C, max_batch=G, seed=0, precision="f16x2")

    def make():
        return SelfPlayEngine(net, n, G, sims, 1.0, 1.0, 0.9, seed=1234, first_game_id=0, game_id_stride=G, q_mode=1, refill=True,
                              record_cap=G * 16, dedup=dedup)            # bench headline (off) and the library default (on)
    lock = make()
    lock.stagger(8)
    lock.run(3)
    rl, sl = lock.records(), lock.stats()
    del lock
    free = make()
    if cap:                                                                 # bench.py's cap: conv3 on 1024 tiles of 256 x 256 = 4.0 grid rounds
        free.set_batch_cap(cap)
    free.stagger(8)
    free.run_steps(3 * sims + (80 if cap else 40))
    rf, sf = free.records(), free.stats()
    assert sl["overflow"] == 0 and sf["overflow"] == 0
    il, jf = set(int(x) for x in np.unique(rl["game_id"])), set(int(x) for x in np.unique(rf["game_id"]))
    both = sorted(il & jf)
    assert len(both) >= 150, (len(il), len(jf), len(both))                  # ~68 games complete per move round
    order = lambda r: r[np.lexsort((r["ply"], r["game_id"]))]
    a, b = order(rl[np.isin(rl["game_id"], both)]), order(rf[np.isin(rf["game_id"], both)])
    assert a.tobytes() == b.tobytes()


def test_evaluation_cache_at_config2_size_changes_no_record(oz):
    """the network's persistent evaluation cache at the bench's own size: 4096 staggered 8x8 games x 100 simulations on the 512-filter
    network (f16x2), free-running driver with the batch cap, de-duplication on: every game that completes with and without the cache has
    the same records bit for bit, and root visit counts agree; a large share of the leaves never reaches the network"""
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    n, G, sims = 8, B, 100
    net = NNetWrapper((n, n), num_channels_1=C, max_batch=G, seed=0, precision="f16x2")

    def run(cache):
        e = SelfPlayEngine(net, n, G, sims, 1.0, 1.0, 0.9, seed=1234, first_game_id=0, game_id_stride=G, q_mode=1, refill=True,
                           record_cap=G * 16, dedup=True, batch_cap=3640, eval_cache=cache)
        e.stagger(8)
        e.run_steps(3 * sims + 60)
        return e.records(), e.stats()
    r0, s0 = run(False)
    net.set_eval_cache(1 << 22)
    r1, s1 = run(True)
    assert s0["overflow"] == 0 and s1["overflow"] == 0
    both = sorted(set(int(x) for x in np.unique(r0["game_id"])) & set(int(x) for x in np.unique(r1["game_id"])))
    assert len(both) >= 150
    order = lambda r: r[np.lexsort((r["ply"], r["game_id"]))]
    a, b = order(r0[np.isin(r0["game_id"], both)]), order(r1[np.isin(r1["game_id"], both)])
    assert a.tobytes() == b.tobytes()
    st = net.eval_cache_stats()
    # (steady-state self-play: the games sit at every ply, only the opening plies recur across games and generations -- a few per cent of
    #  the lookups in a window this short, on top of the ~25 % the same-batch de-duplication already shares)
    assert st["hits"] > 20000 and st["hits"] + st["inserts"] <= st["lookups"] and s1["leaves_evaluated"] < 0.9 * s1["expansions"]
    assert s1["leaves_evaluated"] < s0["leaves_evaluated"]
    assert s1["expansions"] >= s0["expansions"]              # hits need no batch slot: the same number of batches serves at least as many leaves


def test_config3_last_rank_shard_at_full_size(oz):
    """BASELINE configs[2] (32768 games over 8 GPUs) as ONE rank sees it: the engine of rank 7 of 8 -- 4096 slots holding
    global game ids [28672, 32768), refills stepping by the job-wide 32768 -- at 100 sims/move: per-game RNG streams are keyed
    by the GLOBAL id, so sampled games (first and second generation) equal the oracle's episodes of those ids move for move.
    (The all-gather of the eight ranks' records is covered by the world_size-2 tests; RCCL itself needs the 8-GPU node.)"""
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    n, G, world, rank, sims = 6, 4096, 8, 7, 100
    eng = SelfPlayEngine(StubNetWrapper((n, n), 23, 0, max_batch=G), n, G, sims, 1.0, 1.0, 0.9, seed=1234, first_game_id=rank * G,
                         game_id_stride=world * G, q_mode=1, refill=True, record_cap=G * 3 * n * n)
    eng.run(2 * (n * n - 4) + 2)                                           # two generations of 6x6 games
    st = eng.stats()
    assert st["overflow"] == 0 and st["games_completed"] >= 2 * G - 8
    rec = eng.records()
    ids = np.unique(rec["game_id"])
    assert ids.min() == rank * G and np.all((ids % (world * G)) >= rank * G)   # only this rank's residue classes
    for gid in (rank * G, rank * G + 1234, (rank + 1) * G - 1, rank * G + world * G, (rank + 1) * G - 1 + world * G):
        r = rec[rec["game_id"] == gid]
        ep = oracle.Mcts(n, 1.0, 1, salt=23).episode(sims, 1.0, 0.9, 1234, gid)
        assert np.array_equal(r["action"], ep["action"]) and np.array_equal(r["z"], ep["z"]), gid
        assert np.array_equal(r["black"], ep["black"]) and np.array_equal(r["white"], ep["white"]), gid


@pytest.mark.parametrize("driver", ["lockstep", "free_capped"])
def test_soak_two_generations_at_config2_size_vs_oracle(oz, driver):
    """tools/soak_check.py in small: 4096 concurrent 8x8 games x 100 sims with the device stub network, staggered start, 130
    move rounds (every slot finishes its first game and most of a refilled one), 32 randomly sampled games -- first and second
    generation, i.e. through table reset and slot refill -- equal the oracle's episodes move for move"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    extra = ["--driver", "free", "--batch-cap", "3640", "--rounds", "150"] if driver == "free_capped" else ["--rounds", "130"]      # bench.py's driver and cap
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_check.py"), "--sample", "32"] + extra,
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["mismatching_games"] == 0 and out["sampled"] == 32 and out["games_completed"] > 2 * 4096 - 200
