"""GPU parity tests (pytest -m gpu, real MI355X): the HIP path through the C ABI vs the reference-generated
golden fixtures and vs the CPU oracle on the same seeded inputs.  Bit-exact for boards, moves, visit counts
and the float64/float32 search tables; <= 1e-5 for the network outputs (tolerance stated per test)."""
import ctypes as C
import random

import numpy as np
import pytest

import oracle
from conftest import load_golden
from oracle import nn_numpy

pytestmark = pytest.mark.gpu

QT_F32 = 1


@pytest.fixture(scope="module")
def oz():
    import othellozero_amd  # noqa: F401
    from othellozero_amd import _lib
    _lib.require_gpu()
    return _lib


# ------------------------------------------------------------------ device arithmetic
def test_device_arithmetic_is_ieee(oz):
    """sqrt / division / float32 mul-add-div on the device are bit-identical to the host (PUCT, backup)."""
    lib = oz.load()
    rs = np.random.RandomState(0)
    a = np.concatenate([np.arange(0, 1 << 20, dtype=np.float64), rs.rand(200000) * 1e4, rs.standard_normal(100000)])
    b = np.concatenate([1.0 + np.arange(0, 1 << 20, dtype=np.float64) % 4097, rs.rand(200000) + 1e-3,
                        rs.standard_normal(100000) + 3.0])
    a = np.abs(a)
    n = a.size
    sq, dv = np.zeros(n), np.zeros(n)
    fd, fq = np.zeros(n, np.float32), np.zeros(n, np.float32)
    oz.check(lib.oz_selftest_arith(oz.p_f64(a), oz.p_f64(b), n, oz.p_f64(sq), oz.p_f64(dv), oz.p_f32(fd), oz.p_f32(fq)))
    assert np.array_equal(sq, np.sqrt(a))
    assert np.array_equal(dv, a / b)
    x, y = a.astype(np.float32), b.astype(np.float32)
    assert np.array_equal(fd, x / y)
    assert np.array_equal(fq, (x * y + x) / y)


# ------------------------------------------------------------------ rules
def test_rules_vs_golden(oz, golden_rules):
    from othellozero_amd import Othello as O
    g = golden_rules
    for n in (4, 6, 8):
        sel = g["n"] == n
        b, w = g["black"][sel], g["white"][sel]
        assert np.array_equal(O.rules_legal_moves(b, w, n), g["legal_black"][sel])
        assert np.array_equal(O.rules_legal_moves(w, b, n), g["legal_white"][sel])
        fin, p0, p1, win = O.rules_status(b, w, n)
        assert np.array_equal(fin, g["finished"][sel]) and np.array_equal(win, g["winner"][sel])
        assert np.array_equal(p0, g["pts_black"][sel]) and np.array_equal(p1, g["pts_white"][sel])
        msel = g["n"][g["mv_pos"]] == n
        pos = g["mv_pos"][msel]
        isb = g["mv_player"][msel] == 1
        own = np.where(isb, g["black"][pos], g["white"][pos])
        opp = np.where(isb, g["white"][pos], g["black"][pos])
        o2, p2 = O.rules_apply_moves(own, opp, g["mv_sq"][msel], n)
        assert np.array_equal(np.where(isb, o2, p2), g["mv_black"][msel])
        assert np.array_equal(np.where(isb, p2, o2), g["mv_white"][msel])
        psel = g["pl_n"] == n
        b2, w2, pl2, f2 = O.rules_play(g["pl_black"][psel], g["pl_white"][psel], g["pl_player"][psel], g["pl_sq"][psel], n)
        assert np.array_equal(b2, g["pl_black2"][psel]) and np.array_equal(w2, g["pl_white2"][psel])
        assert np.array_equal(pl2, g["pl_player2"][psel]) and np.array_equal(f2, g["pl_finished2"][psel])


def test_rules_vs_oracle_random_boards(oz):
    """2e4 random fillings per size (dense runs -> flip-through), every legal move applied"""
    from othellozero_amd import Othello as O
    L = oracle.lib()
    rs = np.random.RandomState(5)
    for n in (4, 6, 8):
        cnt = 20000
        valid = sum(1 << (r * 8 + c) for r in range(n) for c in range(n))
        a = rs.randint(0, 1 << 62, size=cnt, dtype=np.int64).astype(np.uint64) * np.uint64(4) + rs.randint(0, 4, size=cnt).astype(np.uint64)
        b = rs.randint(0, 1 << 62, size=cnt, dtype=np.int64).astype(np.uint64) * np.uint64(4) + rs.randint(0, 4, size=cnt).astype(np.uint64)
        e = rs.randint(0, 1 << 62, size=cnt, dtype=np.int64).astype(np.uint64) * np.uint64(4) + rs.randint(0, 4, size=cnt).astype(np.uint64)
        own = a & ~b & (e | a) & np.uint64(valid)
        opp = b & ~a & np.uint64(valid)
        legal = O.rules_legal_moves(own, opp, n)
        idx = rs.choice(cnt, 2000, replace=False)
        for i in idx:
            assert int(legal[i]) == L.orc_legal_mask(int(own[i]), int(opp[i]), n, 0)
        mo, mp, ms = [], [], []
        for i in idx:
            for s in oracle.mask_to_squares(legal[i]):
                mo.append(own[i]); mp.append(opp[i]); ms.append(s)
        o2, p2 = O.rules_apply_moves(np.array(mo, np.uint64), np.array(mp, np.uint64), np.array(ms, np.uint8), n)
        for j in range(0, len(mo), 7):
            x, y = C.c_uint64(int(mo[j])), C.c_uint64(int(mp[j]))
            L.orc_apply_move(C.byref(x), C.byref(y), n, 0, int(ms[j]))
            assert (int(o2[j]), int(p2[j])) == (x.value, y.value)


def test_rules_edge_cases(oz):
    from othellozero_amd import Othello as O
    assert O.rules_legal_moves(np.zeros(0, np.uint64), np.zeros(0, np.uint64), 8).size == 0      # empty batch
    full = np.uint64(0xFFFFFFFFFFFFFFFF)
    fin, p0, p1, win = O.rules_status([full, 0, 0], [0, full, 0], 8)
    assert list(fin) == [1, 1, 1] and list(win) == [1, -1, 1] and list(p0) == [64, 0, 0]          # empty board: draw -> ch0
    # flip-through known answer: . O O X O X  flips 3
    b = (1 << 3) | (1 << 5); w = (1 << 1) | (1 << 2) | (1 << 4)
    o2, p2 = O.rules_apply_moves([b], [w], [0], 8)
    assert int(o2[0]) == b | w | 1 and int(p2[0]) == 0


def test_othello_game_dropin(oz, golden_rules):
    """OthelloGame instance API replayed over golden play() transitions, board mutated in place"""
    from othellozero_amd.Othello import BoardView, OthelloGame, OthelloPlayer
    g = golden_rules
    game, view = None, None
    played = 0
    for j in range(len(g["pl_n"])):
        n = int(g["pl_n"][j])
        if game is None or game.has_finished():
            game = OthelloGame(n)
            view = game.board(BoardView.TWO_CHANNELS)
            if oracle.pack_board(view) != (int(g["pl_black"][j]), int(g["pl_white"][j])):
                game = None
                continue
        if oracle.pack_board(view) != (int(g["pl_black"][j]), int(g["pl_white"][j])) or game.board_size != n:
            game = None
            continue
        sq = int(g["pl_sq"][j])
        assert game.current_player.value == int(g["pl_player"][j])
        acts = [tuple(int(x) for x in a) for a in game.get_valid_actions()]
        assert (sq >> 3, sq & 7) in acts and acts == sorted(acts)
        game.play(sq >> 3, sq & 7)
        assert oracle.pack_board(view) == (int(g["pl_black2"][j]), int(g["pl_white2"][j]))   # same array object
        assert game.current_player.value == int(g["pl_player2"][j]) and game.has_finished() == bool(g["pl_finished2"][j])
        played += 1
        if played > 400:
            break
    assert played > 300
    with pytest.raises(AssertionError):
        fin = OthelloGame(4, initial_board=np.ones((4, 4, 2), dtype=bool) & np.array([True, False]))
        fin.play(0, 0)


def test_symmetry_expansion(oz):
    from othellozero_amd import _lib
    from othellozero_amd.training import expand_examples, training_example_symmetries
    g = load_golden("symmetries.npz")
    for n in (4, 6, 8):
        perm = np.zeros((8, n * n), np.int32)
        _lib.check(_lib.load().oz_symmetry_table(n, _lib.p_i32(perm)))
        assert np.array_equal(perm, g[f"perm_{n}"])
    rs = np.random.RandomState(1)
    n = 8
    rec = np.zeros(50, dtype=_lib.RECORD_DTYPE)
    r64 = lambda: rs.randint(0, 1 << 62, 50).astype(np.uint64) * np.uint64(4) + rs.randint(0, 4, 50).astype(np.uint64)
    rec["black"] = r64(); rec["white"] = r64() & ~rec["black"]
    rec["final_black"] = r64(); rec["final_white"] = ~rec["final_black"]
    rec["action"] = rs.randint(0, 64, 50); rec["z"] = rs.choice([-1, 1], 50)
    for alias in (False, True):
        boards, pol, z = expand_examples(rec, n, alias_final=alias)
        for i in range(50):
            src = _lib.unpack_board(int(rec["final_black" if alias else "black"][i]), int(rec["final_white" if alias else "white"][i]), n)
            onehot = np.zeros((n, n)); a = int(rec["action"][i]); onehot[a >> 3, a & 7] = 1
            for t, (sb, sp) in enumerate(training_example_symmetries(src, onehot)):
                assert np.array_equal(boards[8 * i + t].astype(bool), sb)
                assert int(pol[8 * i + t]) == int(np.argmax(sp)) and z[8 * i + t] == rec["z"][i]


# ------------------------------------------------------------------ search
class PyStubNet:
    """host-side duck-typed net (like the reference's tests would use): oracle's stub formula"""
    def __init__(self, n, salt, keep, f64):
        from othellozero_amd.NNet import NeuralNets
        self.network_type = NeuralNets.ONN
        self.n, self.salt, self.keep, self.f64 = n, salt, keep, f64
        self.calls = 0

    def predict(self, board):
        self.calls += 1
        own, opp = oracle.pack_board(board)
        return oracle.stub_predict(own, opp, self.n, self.salt, self.keep)


def _check_tables(dump, g, prefix):
    boards = g[prefix + "boards"]
    assert len(dump) == len(boards)
    for i, nd in enumerate(dump):
        assert (nd["k0"], nd["k1"]) == (int(boards[i][0]), int(boards[i][1])), (prefix, i)
        assert nd["Ns"] == int(g[prefix + "Ns"][i]) and nd["legal"] == int(g[prefix + "legal"][i])
        assert np.array_equal(nd["P"], g[prefix + "P"][i]), (prefix, i)
        assert np.array_equal(nd["N"], g[prefix + "N"][i]), (prefix, i)
        assert np.array_equal(nd["Q"], g[prefix + "Q"][i]), (prefix, i)
        if g[prefix + "edges_init"][i]:
            for sq in oracle.mask_to_squares(nd["legal"]):
                if nd["N"][sq]:
                    assert (nd["qtag"][sq] == 1) == (g[prefix + "qtype"][i][sq] == QT_F32), (prefix, i, sq)


@pytest.mark.parametrize("device_net", [False, True])
def test_mcts_traces_vs_golden(oz, golden_mcts, device_net):
    """OthelloMCTS.simulate sim by sim: full (Ns, Nsa, Qsa, Psa) tables, return values and their dynamic types
    after k simulations, for a host-side duck-typed net and for the device stub net."""
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.Othello import OthelloPlayer
    from othellozero_amd.othelo_mcts import OthelloMCTS
    g = golden_mcts
    for name in g["names"]:
        name = str(name)
        n, player, salt, keep, qmode, nsims = (int(x) for x in g[f"{name}/meta"])
        c = float(g[f"{name}/c"][0])
        root = oz.unpack_board(int(g[f"{name}/root"][0]), int(g[f"{name}/root"][1]), n)
        net = StubNetWrapper((n, n), salt, keep) if device_net else PyStubNet(n, salt, keep, qmode == 1)
        m = OthelloMCTS(n, net, c, q_mode=qmode, node_cap=1024)
        pl = OthelloPlayer(player)
        done, rets, rts = 0, [], []
        for cp in g[f"{name}/cps"]:
            while done < int(cp):
                r = m.simulate(root, pl)
                rets.append(float(r))
                rts.append(0 if isinstance(r, int) else (1 if isinstance(r, np.float32) else 2))
                done += 1
            _check_tables(m.dump(), g, f"{name}/cp{int(cp)}/")
        assert np.array_equal(np.array(rets), g[f"{name}/ret"]), name
        assert np.array_equal(np.array(rts, np.uint8), g[f"{name}/ret_type"]), name
        state = root if player == 1 else root[:, :, ::-1]
        assert np.array_equal(m.get_policy_action_probabilities(state, 1), g[f"{name}/pi_T1"]), name
        if not device_net:
            assert net.calls == len(g[f"{name}/cp{int(g[f'{name}/cps'][-1])}/boards"])     # one predict per expansion


def test_mcts_keyerror_and_unknown_state(oz):
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.Othello import OthelloGame, OthelloPlayer
    from othellozero_amd.othelo_mcts import OthelloMCTS
    m = OthelloMCTS(6, StubNetWrapper((6, 6), 1), 1)
    root = OthelloGame.initial_board(6)
    assert m.N(root) == 0 and m.N(root, (1, 2)) == 0               # unknown state -> 0 (MCTS/__init__.py:173-174)
    m.simulate(root, OthelloPlayer.BLACK)                          # only expands the root
    with pytest.raises(KeyError):
        m.N(root, (1, 2))                                          # _Nsa[hash] is still {}
    with pytest.raises(KeyError):
        m.get_policy_action_probabilities(root, 1)
    m.simulate(root, OthelloPlayer.BLACK)
    assert m.N(root) == 1 and sum(m.N(root, a) for a in m.get_state_actions(root)) == 1


# ------------------------------------------------------------------ drivers
def _patch_rng(monkeypatch, seed, game, ply_of):
    L = oracle.lib()
    monkeypatch.setattr(random, "random", lambda: (L.orc_rng(seed, game, ply_of(), 0) >> 11) * (1.0 / 9007199254740992.0))
    monkeypatch.setattr(random, "choice", lambda seq: seq[L.orc_rng(seed, game, ply_of(), 2) % len(seq)])
    monkeypatch.setattr(np.random, "choice", lambda k: L.orc_rng(seed, game, ply_of(), 1) % k)


@pytest.mark.parametrize("name", ["ep8_25", "ep8_25_f64", "ep8_T0", "ep6_50_f64_c2", "ep6_sparse", "ep4_60"])
def test_execute_episode_dropin_vs_golden(oz, golden_episodes, monkeypatch, name):
    """training.execute_episode with the reference's own random calls patched exactly as the fixture generator
    patched them for the reference: same moves, same returned examples (incl. the aliasing quirk), same z."""
    from othellozero_amd import training
    from othellozero_amd.Othello import OthelloGame
    g = golden_episodes
    n, sims, seed, game, salt, keep, qmode, k = (int(x) for x in g[f"{name}/meta"])
    c, T, eg = (float(x) for x in g[f"{name}/params"])
    ply = [0]
    orig_play = OthelloGame.play

    def counting_play(self, row, col):
        orig_play(self, row, col)
        ply[0] += 1
    monkeypatch.setattr(OthelloGame, "play", counting_play)
    _patch_rng(monkeypatch, seed, game, lambda: ply[0])
    net = PyStubNet(n, salt, keep, qmode == 1)
    T_arg = int(T) if T == int(T) else T
    ex = training.execute_episode(n, net, int(c) if c == int(c) else c, sims, T_arg, eg, q_mode=qmode)
    assert len(ex) == 8 * k
    eb, ep, ez = g[f"{name}/ex_board"], g[f"{name}/ex_policy"], g[f"{name}/ex_z"]
    for i, (b, p, z) in enumerate(ex):
        assert b.dtype == np.bool_ and p.dtype == np.float64 and isinstance(z, int)
        assert oracle.pack_board(b) == (int(eb[i][0]), int(eb[i][1])), (name, i)
        assert int(np.argmax(p)) == int(ep[i]) and p.sum() == 1.0 and z == int(ez[i]), (name, i)
    assert net.calls == int(g[f"{name}/n_expansions"][0])


def test_selfplay_engine_vs_golden_episodes(oz, golden_episodes):
    """the batched engine (device stub net, counter RNG streams) replays the reference's episodes move for move"""
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    g = golden_episodes
    for name in g["names"]:
        name = str(name)
        n, sims, seed, game, salt, keep, qmode, k = (int(x) for x in g[f"{name}/meta"])
        c, T, eg = (float(x) for x in g[f"{name}/params"])
        eng = SelfPlayEngine(StubNetWrapper((n, n), salt, keep, max_batch=1), n, 1, sims, c, T, eg, seed=seed,
                             first_game_id=game, q_mode=qmode)
        counts = []
        for _ in range(k):
            eng.run(1)
            counts.append(eng.last_counts()[0].copy())
        rec = eng.records()
        assert rec.size == k and eng.stats()["live_games"] == 0, name
        assert np.array_equal(rec["action"], g[f"{name}/action"]) and np.array_equal(rec["player"], g[f"{name}/player"]), name
        assert np.array_equal(rec["black"], g[f"{name}/black"]) and np.array_equal(rec["white"], g[f"{name}/white"]), name
        assert np.array_equal(np.array(counts), g[f"{name}/counts"]), name
        assert eng.stats()["expansions"] == int(g[f"{name}/n_expansions"][0]), name
        assert np.array_equal(np.repeat(rec["z"], 8), g[f"{name}/ex_z"]), name


@pytest.mark.parametrize("n,sims,T,qmode,keep", [(8, 40, 1.0, 1, 0), (6, 60, 0.0, 0, 0), (6, 30, 1.0, 0, 7), (4, 50, 1.0, 1, 3)])
def test_selfplay_engine_vs_oracle_many_games(oz, n, sims, T, qmode, keep):
    """64 concurrent games in lock step == 64 sequential oracle episodes (moves, snapshots, z, counters)"""
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    G, seed, first = 64, 777, 1000
    eng = SelfPlayEngine(StubNetWrapper((n, n), 9, keep, max_batch=G), n, G, sims, 1.25, T, 0.8, seed=seed,
                         first_game_id=first, q_mode=qmode)
    rec = eng.play_to_end()
    st = eng.stats()
    tot = dict(visits=0, expansions=0, terminal=0, fallback=0)
    off = 0
    for gi in range(G):
        m = oracle.Mcts(n, 1.25, qmode, salt=9, keep_mask=keep)
        ep = m.episode(sims, T, 0.8, seed, first + gi)
        k = ep["n_moves"]
        r = rec[off:off + k]; off += k
        assert np.all(r["game_id"] == first + gi) and np.array_equal(r["ply"], np.arange(k))
        assert np.array_equal(r["action"], ep["action"]) and np.array_equal(r["player"], ep["player"]), gi
        assert np.array_equal(r["black"], ep["black"]) and np.array_equal(r["white"], ep["white"]), gi
        assert np.array_equal(r["z"], ep["z"]) and np.array_equal(r["greedy"], ep["greedy"]), gi
        assert np.all(r["final_black"] == ep["final_black"]) and np.all(r["final_white"] == ep["final_white"])
        for key in tot:
            tot[key] += ep["stats"][key]
    assert off == rec.size
    assert (st["node_visits"], st["expansions"], st["terminal_hits"], st["fallbacks"]) == \
        (tot["visits"], tot["expansions"], tot["terminal"], tot["fallback"])
    assert st["games_completed"] == G and st["moves"] == rec.size
    if keep:
        assert st["fallbacks"] > 0


def test_selfplay_refill_and_capacity_error(oz):
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    net = StubNetWrapper((4, 4), 3, 0, max_batch=8)
    eng = SelfPlayEngine(net, 4, 8, 20, refill=True, seed=1, first_game_id=0, game_id_stride=8)
    eng.run(40)
    st = eng.stats()
    assert st["games_completed"] >= 16 and st["live_games"] == 8
    rec = eng.records()
    ids = np.unique(rec["game_id"])
    assert ids.size == st["games_completed"] and ids.max() >= 8          # refilled slots carry new global ids
    for gid in ids[:12]:                                                 # a refilled game == a fresh oracle episode
        ep = oracle.Mcts(4, 1.0, 1, salt=3).episode(20, 1.0, 0.9, 1, int(gid))
        r = rec[rec["game_id"] == gid]
        assert np.array_equal(r["action"], ep["action"]) and np.array_equal(r["z"], ep["z"])
    small = SelfPlayEngine(net, 4, 8, 20, node_cap=16)
    with pytest.raises(oz.OzError) as ei:
        small.run(12)
    assert ei.value.code == oz.OZ_ERR_CAPACITY


def test_arena_vs_golden_and_oracle(oz, golden_arena):
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.agents import arena_batch
    g = golden_arena
    for name in g["names"]:
        name = str(name)
        n, sims, seed, game, sa, sb, qmode, k = (int(x) for x in g[f"{name}/meta"])
        c = float(g[f"{name}/c"][0])
        r = arena_batch(StubNetWrapper((n, n), sa, 0, max_batch=1), StubNetWrapper((n, n), sb, 0, max_batch=1), n, 1, sims, c,
                        seed=seed, first_game_id=game, q_mode=qmode)
        assert int(r["n_moves"][0]) == k, name
        assert np.array_equal(r["actions"][0][:k], g[f"{name}/action"]) and np.array_equal(r["players"][0][:k], g[f"{name}/player"])
        assert (int(r["final_black"][0]), int(r["final_white"][0])) == tuple(int(x) for x in g[f"{name}/final"])
        assert (int(r["winner"][0]), int(r["points"][0])) == tuple(int(x) for x in g[f"{name}/result"]), name
    # 32 concurrent deterministic games vs the oracle
    G, n, sims = 32, 6, 120
    r = arena_batch(StubNetWrapper((n, n), 41, 0, max_batch=G), StubNetWrapper((n, n), 42, 0, max_batch=G), n, G, sims, 1.0,
                    seed=7, first_game_id=500, q_mode=0)
    for gi in range(G):
        o = oracle.arena(oracle.Mcts(n, 1.0, 0, salt=41), oracle.Mcts(n, 1.0, 0, salt=42), sims, 7, 500 + gi)
        k = o["n_moves"]
        assert int(r["n_moves"][gi]) == k and np.array_equal(r["actions"][gi][:k], o["action"]), gi
        assert (int(r["winner"][gi]), int(r["points"][gi])) == (o["winner"], o["points"]), gi


def test_arena_against_random_agent_vs_golden_and_oracle(oz):
    """batched arena with RandomOthelloAgent on one colour (oz_arena_create with a NULL network) == the reference's traces
    (arena_random.npz) and == the oracle on 24 concurrent games per colour; the loop-level evaluation helper on top"""
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.agents import arena_batch
    from othellozero_amd.loop import evaluate_against_random_batch
    g = load_golden("arena_random.npz")
    for name in g["names"]:
        name = str(name)
        n, sims, seed, game, salt, colour, qmode, k = (int(x) for x in g[f"{name}/meta"])
        net = StubNetWrapper((n, n), salt, 0, max_batch=1)
        r = arena_batch(net if colour == 1 else None, None if colour == 1 else net, n, 1, sims, float(g[f"{name}/c"][0]),
                        seed=seed, first_game_id=game, q_mode=qmode)
        assert int(r["n_moves"][0]) == k, name
        assert np.array_equal(r["actions"][0][:k], g[f"{name}/action"]) and np.array_equal(r["players"][0][:k], g[f"{name}/player"]), name
        assert (int(r["final_black"][0]), int(r["final_white"][0])) == tuple(int(x) for x in g[f"{name}/final"]), name
        assert (int(r["winner"][0]), int(r["points"][0])) == tuple(int(x) for x in g[f"{name}/result"][:2]), name
    G, n, sims = 24, 6, 40
    net = StubNetWrapper((n, n), 51, 0, max_batch=G)
    for colour in (1, -1):
        r = arena_batch(net if colour == 1 else None, None if colour == 1 else net, n, G, sims, 1.0, seed=9, first_game_id=700, q_mode=1)
        for gi in range(G):
            m = oracle.Mcts(n, 1.0, 1, salt=51)
            o = oracle.arena(m if colour == 1 else None, None if colour == 1 else m, sims, 9, 700 + gi)
            k = o["n_moves"]
            assert int(r["n_moves"][gi]) == k and np.array_equal(r["actions"][gi][:k], o["action"]), (colour, gi)
            assert (int(r["winner"][gi]), int(r["points"][gi])) == (o["winner"], o["points"]), (colour, gi)
    ev = evaluate_against_random_batch(n, net, 11, sims, 1.0, seed=3)
    assert ev["black_games"] + ev["white_games"] == 11 and ev["wins"] == ev["black_wins"] + ev["white_wins"] <= 11
    assert ev["black_wins"] <= 6 and ev["white_wins"] <= 5


def test_agents_dropin_duel(oz, golden_arena, monkeypatch):
    """duel_between_agents with two NeuralNetworkOthelloAgent (host-side nets) == the reference's trace"""
    from othellozero_amd.Othello import OthelloGame
    from othellozero_amd.agents import NeuralNetworkOthelloAgent, duel_between_agents
    g = golden_arena
    name = "ar6_200_f64"
    n, sims, seed, game, sa, sb, qmode, k = (int(x) for x in g[f"{name}/meta"])
    ply = [0]
    log = []
    orig_play = OthelloGame.play

    def counting_play(self, row, col):
        log.append((self.current_player.value, int(row) * 8 + int(col)))
        orig_play(self, row, col)
        ply[0] += 1
    monkeypatch.setattr(OthelloGame, "play", counting_play)
    _patch_rng(monkeypatch, seed, game, lambda: ply[0])
    game_obj = OthelloGame(n)
    a1 = NeuralNetworkOthelloAgent(game_obj, PyStubNet(n, sa, 0, True), sims, 1, q_mode=qmode)
    a2 = NeuralNetworkOthelloAgent(game_obj, PyStubNet(n, sb, 0, True), sims, 1, q_mode=qmode)
    winner, points = duel_between_agents(game_obj, a1, a2)
    assert [x[1] for x in log] == list(g[f"{name}/action"]) and [x[0] for x in log] == list(g[f"{name}/player"])
    assert (1 if winner is a1 else -1, points) == tuple(int(x) for x in g[f"{name}/result"])


# ------------------------------------------------------------------ network
def _boards(n, count, seed):
    rs = np.random.RandomState(seed)
    own, opp = [], []
    for _ in range(count):
        a = rs.rand(n, n) < 0.4
        b = (rs.rand(n, n) < 0.4) & ~a
        o, p = oracle.pack_board(np.stack([a, b], axis=2))
        own.append(o); opp.append(p)
    return np.array(own, np.uint64), np.array(opp, np.uint64)


@pytest.mark.parametrize("n,channels,batch,precision", [
    (8, 128, 37, "f32"), (6, 128, 70, "f32"), (8, 512, 9, "f32"), (6, 512, 5, "f32"),
    (8, 256, 37, "f16x2"), (6, 256, 70, "f16x2"), (8, 512, 9, "f16x2"), (6, 512, 5, "f16x2")])
def test_network_vs_float64_oracle(oz, n, channels, batch, precision):
    """pi and v within 1e-5 (absolute) of the float64 restatement of OthelloNN; every parameter kind random
    (kernels, biases, BN gamma / beta / moving mean / moving variance); logits lifted away from uniform."""
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.weights import init_weights
    w = init_weights(n, seed=11, channels=channels, randomize_all=True)
    for i in (36, 38):
        w[i] = w[i] * 4.0
    net = NNetWrapper((n, n), num_channels_1=channels, max_batch=64, weights=w, precision=precision)
    own, opp = _boards(n, batch, seed=n + channels)
    pi, v = net.predict_batch(own, opp)                        # batch > max_batch exercises the chunking too
    pi64, v64 = nn_numpy.forward(w, own, opp, n)
    assert pi.dtype == np.float32 and v.dtype == np.float32 and pi.shape == (batch, n, n)
    assert pi64.std() > 2e-3 and np.abs(v64).max() > 0.05       # the test is not vacuous
    assert np.abs(pi.reshape(batch, -1) - pi64).max() <= 1e-5
    assert np.abs(v - v64).max() <= 1e-5
    assert np.abs(pi.reshape(batch, -1).sum(axis=1) - 1).max() < 1e-5
    # a position's result does not depend on its place in the batch, nor on the batch size (bit-exact)
    perm = np.random.RandomState(0).permutation(batch)
    pi2, v2 = net.predict_batch(own[perm], opp[perm])
    assert np.array_equal(pi2, pi[perm]) and np.array_equal(v2, v[perm])
    p1, v1 = net.predict(oz.unpack_board(int(own[3]), int(opp[3]), n))
    assert np.array_equal(p1, pi[3]) and v1 == v[3] and isinstance(v1, np.float32)
    # get/set weights round trip and copy()
    cp = net.copy()
    assert all(np.array_equal(a, b) for a, b in zip(cp.get_weights(), w))
    pc, vc = cp.predict_batch(own[:4], opp[:4])
    assert np.array_equal(pc, pi[:4]) and np.array_equal(vc, v[:4])


@pytest.mark.parametrize("n,channels,batch,network", [(8, 256, 37, "ONN"), (6, 256, 200, "ONN"), (8, 512, 130, "ONN"), (6, 512, 9, "BNN"), (8, 768, 21, "ONN")])
def test_bf16x3_network_vs_float64_oracle(oz, n, channels, batch, network):
    """precision bf16x3 (every fp32 value as three bf16 planes, six MFMA products; oz_net_b3.h) on networks of max_batch >= 128 -- the k_gemm_b3 path --
    at channel counts other than the bench's (256 / 768: the thread-per-pixel gather with its own b3 output, one / three column tiles), both boards,
    both network types, every parameter kind random: pi, v within 1e-5 of the float64 oracle AND within 4e-6 of the exact-fp32 kernels (fp32-class:
    the two differ by rounding only); a position's bits do not depend on its place in the batch, on the size of the call or on the tables mode's GEMM
    twin being built in another object; a max_batch = 64 network of the same precision runs the exact-fp32 latency kernels and says so"""
    from othellozero_amd.NNet import NNetWrapper, NeuralNets
    from othellozero_amd.weights import init_weights
    cin = 2 if network == "ONN" else 1
    w = init_weights(n, seed=13, channels=channels, randomize_all=True, in_channels=cin)
    for i in (36, 38):
        w[i] = w[i] * 4.0
    kind = NeuralNets.ONN if cin == 2 else NeuralNets.BNN
    net = NNetWrapper((n, n), num_channels_1=channels, max_batch=160, weights=w, precision="bf16x3", network=kind)
    assert net.arithmetic() == "bf16x3"
    own, opp = _boards(n, batch, seed=n + channels)
    pi, v = net.predict_batch(own, opp)                        # batch > max_batch exercises the chunking too
    assert net.conv3_tile_rows() == 128
    pi64, v64 = nn_numpy.forward(w, own, opp, n)
    assert np.abs(pi.reshape(batch, -1) - pi64).max() <= 1e-5 and np.abs(v - v64).max() <= 1e-5
    assert np.abs(pi.reshape(batch, -1).sum(axis=1) - 1).max() < 1e-5
    exact = NNetWrapper((n, n), num_channels_1=channels, max_batch=160, weights=w, precision="f32", network=kind)
    pe, ve = exact.predict_batch(own, opp)
    # two differently ROUNDED fp32 evaluations of the same network (each ~1e-6 from float64 on this all-parameters-random, heads x 4 network)
    assert np.abs(pi - pe).max() <= 4e-6 and np.abs(v - ve).max() <= 4e-6
    assert abs(np.abs(v - v64).max() - np.abs(ve - v64).max()) <= 2e-6        # ... and neither is further from float64 than the other by more than rounding
    perm = np.random.RandomState(0).permutation(batch)
    pi2, v2 = net.predict_batch(own[perm], opp[perm])
    assert np.array_equal(pi2, pi[perm]) and np.array_equal(v2, v[perm])
    p1, v1 = net.predict_batch(own[3:4], opp[3:4])
    assert np.array_equal(p1[0], pi[3]) and v1[0] == v[3]
    small = NNetWrapper((n, n), num_channels_1=channels, max_batch=64, weights=w, precision="bf16x3", network=kind)
    assert small.arithmetic() == "f32"
    ps, vs = small.predict_batch(own[:8], opp[:8])
    assert np.abs(ps - pi[:8]).max() <= 4e-6 and np.abs(vs - v[:8]).max() <= 4e-6


def test_bf16x3_split_is_exact(oz):
    """what precision bf16x3 rests on, on the device: x == (b1 + b2) + b3 BIT FOR BIT for every finite fp32 x with |x| >= 2^-100 (bf16 has fp32's
    exponent range, the three planes carry 8 + 8 + 8 significand bits; round to nearest makes the residuals small enough), |b2| <= 2^-8 |b1| and
    |b3| <= 2^-16 |b1| (the order of the six kept cross terms), zero stays zero with its sign; below 2^-100 a residual can fall under the smallest
    normal fp32 / bf16's subnormal step and the ABSOLUTE error stays below 2^-120 (1e-36: nothing a network's activations or weights resolve)"""
    rs = np.random.RandomState(7)
    bits = rs.randint(0, 2**32, size=1 << 20, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32).copy()
    x = x[np.isfinite(x)]
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0**-23, 1.0 - 2.0**-24, 3.4028235e38, -3.4028235e38, 2.0**-109, 2.0**-126, 65504.0, 1.0 / 3.0,
                     np.float32(2.0**-8) * np.float32(1.0 + 2.0**-23), 255.99998], np.float32)
    x = np.concatenate([edge, x]).astype(np.float32)
    planes, total = np.zeros(3 * x.size, np.float32), np.zeros(x.size, np.float32)
    oz.check(oz.load().oz_selftest_b3_split(oz.p_f32(x), x.size, oz.p_f32(planes), oz.p_f32(total)))
    planes = planes.reshape(-1, 3)
    big = np.abs(x) >= np.float32(2.0**-100)
    # the three planes add up to x EXACTLY (float64 on the host holds the 3 x 8 bits without rounding) over fp32's whole finite range ...
    assert np.array_equal(planes[big].astype(np.float64).sum(axis=1), x[big].astype(np.float64)), int((planes[big].astype(np.float64).sum(axis=1) != x[big]).sum())
    assert np.isfinite(planes).all()                                # ... including its top 0.4 %, where bf16(x) would round to infinity (b1 = bf16's largest value there)
    # ... and so does the device's own fp32 evaluation (b1 + b2) + b3, wherever b1 + b2 cannot overflow on the way
    inner = big & (np.abs(x) < np.float32(2.0**126))
    assert np.array_equal(total[inner].view(np.uint32), x[inner].view(np.uint32)), int((total[inner].view(np.uint32) != x[inner].view(np.uint32)).sum())
    small = ~big
    assert np.abs(total[small].astype(np.float64) - x[small].astype(np.float64)).max() <= 2.0**-120
    assert np.array_equal(planes[:2, 0].copy().view(np.uint32), x[:2].view(np.uint32)) and not planes[:2, 1:].any()      # +0.0 and -0.0: the first plane keeps the sign
    # every plane is a bf16 value (16 low bits of the fp32 pattern clear) and the planes fall off by 2^-8 each (in magnitude, up to rounding to even)
    assert not (planes.view(np.uint32) & 0xFFFF).any()
    b1, b2, b3 = (np.abs(planes[big, k].astype(np.float64)) for k in range(3))
    assert (b2 <= b1 / 255.0).all() and (b3 <= b1 * 2.0**-16).all()        # (1 / 255 exactly at the clamped top of the range: b1 = 255 * 2^120, b2 = 2^120; 2^-8 elsewhere)
    top = np.abs(x[big]) < np.float32(3.0e38)
    assert (b2[top] <= b1[top] * 2.0**-8).all()


def test_bf16x3_needs_channels_multiple_of_256(oz):
    from othellozero_amd.NNet import NNetWrapper
    with pytest.raises(oz.OzError):
        NNetWrapper((8, 8), num_channels_1=128, precision="bf16x3")


def test_f16x2_range_guards_fail_loudly(oz):
    """precision f16x2 must refuse (not silently mis-compute) positions whose activations leave the window the 2 x fp16 split carries:
    HIGH side (an activation above 65504) and LOW side (a pixel row whose largest scaled activation is non-zero and below the threshold).
    oz_net_commit puts every tested network inside the window, so both guards are provoked through their options: calibration maxima moved
    to 2^18 (above the fp16 range), the row threshold raised above every activation."""
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.weights import init_weights
    w = init_weights(8, seed=4, channels=256, randomize_all=True)
    own, opp = _boards(8, 4, seed=1)
    for max_batch in (4, 64):                                  # latency path (split-K reduce epilogues) and the tile epilogues
        net = NNetWrapper((8, 8), num_channels_1=256, max_batch=max_batch, weights=w, precision="f16x2")
        pi, v = net.predict_batch(own, opp)                    # defaults: inside the window
        pi64, v64 = nn_numpy.forward(w, own, opp, 8)
        assert np.abs(pi.reshape(4, -1) - pi64).max() <= 1e-5 and np.abs(v - v64).max() <= 1e-5
        net.set_option(oz.NET_OPT_ACT_TARGET_LOG2, 18)
        net.commit()                                           # (the self-check only runs at the default target)
        with pytest.raises(oz.OzError) as ei:
            net.predict_batch(own, opp)
        assert ei.value.code == oz.OZ_ERR_STATE and "fp16 range" in str(ei.value)
        net.set_option(oz.NET_OPT_ACT_TARGET_LOG2, -2)
        net.set_option(oz.NET_OPT_LOW_GUARD_LOG2, 12)          # every row's maximum is "low" now
        with pytest.raises(oz.OzError) as ei:                  # ... already on the calibration positions of the commit's self-check
            net.commit()
        assert ei.value.code == oz.OZ_ERR_STATE and "fell below" in str(ei.value)
        assert net.self_check_guard() & 4 and net.self_check()[2] > 0      # the refusal left its measurement behind ...
        net.set_option(oz.NET_OPT_SELF_CHECK, 2)               # "measure only" (ADVICE r4): the same commit succeeds, says what it saw,
        net.commit()
        assert net.self_check_guard() & 4 and net.self_check()[2] > 0 and net.self_check()[0] >= 0
        assert oz.load().oz_net_check(net._h) == oz.OZ_OK       # ... and does not leave the device flag sticky
        net.set_option(oz.NET_OPT_SELF_CHECK, 0)               # without it the commit succeeds and the guard fires where positions are evaluated
        net.commit()
        assert net.self_check_guard() == 0
        with pytest.raises(oz.OzError) as ei:
            net.predict_batch(own, opp)
        assert ei.value.code == oz.OZ_ERR_STATE and "fell below" in str(ei.value)
        for mode in (0, 1):                                    # the conv1 kernel / conv1-table-in-the-GEMM forms guard act1 themselves
            net.set_tables(mode)
            with pytest.raises(oz.OzError) as ei:
                net.predict_batch(own, opp)
            assert ei.value.code == oz.OZ_ERR_STATE
        net.set_tables(-1)
        net.set_option(oz.NET_OPT_LOW_GUARD_LOG2, -17)
        net.set_option(oz.NET_OPT_SELF_CHECK, 1)
        net.commit()                                           # back to the defaults: the same bits as before
        pi2, v2 = net.predict_batch(own, opp)
        assert np.array_equal(pi2, pi) and np.array_equal(v2, v)
    with pytest.raises(oz.OzError):
        NNetWrapper((8, 8), num_channels_1=128, precision="f16x2")       # needs channels % 256 == 0
    # a refusal at commit on the path trained weights take (NNetWrapper.train -> set_weights(on_refusal="f32")): the wrapper warns and the
    # network carries on in exact fp32 on the GPU instead of ending a long run
    net = NNetWrapper((8, 8), num_channels_1=256, max_batch=4, weights=w, precision="f16x2")
    net.set_option(oz.NET_OPT_LOW_GUARD_LOG2, 12)
    with pytest.raises(oz.OzError):
        net.set_weights(w)                                       # default: the caller decides
    with pytest.warns(UserWarning, match="THESE weights run in precision f32"):
        net.set_weights(w, on_refusal="f32")
    assert net.precision == "f32" and net.requested_precision == "f16x2" and net.f16x2_refusals == 1 and oz.load().oz_net_get_precision(net._h) == 0
    pi, v = net.predict_batch(own, opp)
    assert np.abs(pi.reshape(4, -1) - pi64).max() <= 1e-5 and np.abs(v - v64).max() <= 1e-5
    # the fallback holds for THAT set of weights only: the next set_weights asks for f16x2 again -- accepted once the guard is back at its default
    net.set_option(oz.NET_OPT_LOW_GUARD_LOG2, -17)
    net.set_weights(w, on_refusal="f32")
    assert net.precision == "f16x2" and net.f16x2_refusals == 1 and oz.load().oz_net_get_precision(net._h) == 1
    pi, v = net.predict_batch(own, opp)
    assert np.abs(pi.reshape(4, -1) - pi64).max() <= 1e-5 and np.abs(v - v64).max() <= 1e-5


@pytest.mark.parametrize("n,C_", [(8, 256), (6, 512)])
def test_network_vs_the_vendor_libraries(oz, n, C_):
    """a FOURTH opinion on the OthelloNN arithmetic (the NN oracle cannot be pinned by TensorFlow, which is absent everywhere): the same graph
    (Net/OthelloNN.py:42-56) through PyTorch-ROCm on this card, i.e. MIOpen convolutions, MIOpen / native batch norm and rocBLAS GEMMs in
    float32 -- code that shares nothing with this library, the NumPy oracle, the C oracle or torch-CPU's oneDNN path.  All five agree
    within 1e-5 on (pi, v): vendor fp32 vs the float64 oracle, this library in both precisions vs the vendor result."""
    import torch
    import torch.nn.functional as F
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.weights import init_weights
    w = init_weights(n, seed=17, channels=C_, randomize_all=True)
    for i in (36, 38):
        w[i] = w[i] * 4.0
    B = 40
    own, opp = _boards(n, B, seed=9)
    pi64, v64 = nn_numpy.forward(w, own, opp, n)
    dev = torch.device("cuda", 0)
    tw = [torch.from_numpy(np.asarray(a, dtype=np.float32)).to(dev) for a in w]
    with torch.no_grad():
        x = torch.from_numpy(nn_numpy.planes(own, opp, n, dtype=np.float32)).to(dev).permute(0, 3, 1, 2).contiguous()      # NCHW
        for layer, pad in enumerate((1, 1, 0, 0)):
            k, bias, g, b, mu, var = tw[6 * layer:6 * layer + 6]
            x = F.conv2d(x, k.permute(3, 2, 0, 1).contiguous(), bias, padding=pad)
            x = F.batch_norm(x, mu, var, g, b, training=False, eps=1e-3).relu()
        x = x.permute(0, 2, 3, 1).reshape(B, -1)                                                                              # keras Flatten of NHWC
        for layer in (4, 5):
            k, bias, g, b, mu, var = tw[6 * layer:6 * layer + 6]
            x = F.batch_norm(x @ k + bias, mu, var, g, b, training=False, eps=1e-3).relu()
        pit = torch.softmax(x @ tw[36] + tw[37], dim=1).cpu().numpy().astype(np.float64)
        vt = torch.tanh(x @ tw[38] + tw[39])[:, 0].cpu().numpy().astype(np.float64)
    assert np.abs(pit - pi64).max() <= 1e-5 and np.abs(vt - v64).max() <= 1e-5
    for precision in ("f32", "f16x2"):
        net = NNetWrapper((n, n), num_channels_1=C_, max_batch=64, weights=w, precision=precision)
        pi, v = net.predict_batch(own, opp)
        assert np.abs(pi.reshape(B, -1) - pit).max() <= 1e-5 and np.abs(v - vt).max() <= 1e-5, precision


def test_precision_can_be_switched_on_a_live_network(oz):
    """oz_net_set_precision + oz_net_commit on ONE network object, f32 -> f16x2 -> f32 -> f16x2: every commit rebuilds that precision's
    images (scales, tables, the self-check's fp32 copies) and the outputs are bit for bit those of a fresh network of the same precision"""
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.weights import init_weights
    n, C_ = 6, 256
    w = init_weights(n, seed=8, channels=C_, randomize_all=True)
    own, opp = _boards(n, 20, seed=2)
    fresh = {p: NNetWrapper((n, n), num_channels_1=C_, max_batch=32, weights=w, precision=p).predict_batch(own, opp) for p in ("f32", "f16x2")}
    net = NNetWrapper((n, n), num_channels_1=C_, max_batch=32, weights=w, precision="f32")
    lib = oz.load()
    assert net.self_check()[2] == 0                            # no self-check in exact fp32
    for p in ("f32", "f16x2", "f32", "f16x2"):
        oz.check(lib.oz_net_set_precision(net._h, {"f32": 0, "f16x2": 1}[p]))
        net.precision = p
        net.commit()
        pi, v = net.predict_batch(own, opp)
        assert np.array_equal(pi, fresh[p][0]) and np.array_equal(v, fresh[p][1]), p
    dpi, dv, npos = net.self_check()
    assert 0 <= dpi <= 8e-6 and 0 <= dv <= 8e-6 and npos == 64


def _rescaled(w, case, C_):
    """badly scaled but EQUIVALENT (or at least well-posed) parameterisations of the network `w`"""
    w = [a.copy() for a in w]
    rs = np.random.RandomState(99)
    if case == "small_conv2_kernel_small_bn3_variance":        # as asked: conv2 kernel x 2^-12, conv3's BN variance x 2^-24
        w[6] *= 2.0 ** -12
        w[17] *= 2.0 ** -24
    elif case == "tiny_activations":                           # every conv2 output x 2^-12 (BN gamma, beta), undone by conv3's kernel: same function
        w[8] *= 2.0 ** -12; w[9] *= 2.0 ** -12
        w[12] *= 2.0 ** 12
    elif case == "tiny_activations_everywhere":                # conv1 .. conv4 outputs x 2^-10 each, each undone by the next kernel: same function
        for l in range(4):
            w[6 * l + 2] *= 2.0 ** -10; w[6 * l + 3] *= 2.0 ** -10
            w[6 * (l + 1)] *= 2.0 ** 10
    elif case == "weights_span_2^20_by_input_channel":         # conv3 kernel rows x 2^e, e in [-10, 10], undone in conv2's BN: same function
        e = rs.randint(-10, 11, size=C_).astype(np.float64)
        w[12] = (w[12] * (2.0 ** e)[None, None, :, None]).astype(np.float32)
        w[8] = (w[8] * 2.0 ** -e).astype(np.float32); w[9] = (w[9] * 2.0 ** -e).astype(np.float32)
    elif case == "weights_span_2^20_by_output_channel":        # conv3 kernel columns x 2^e, e in [0, 20], undone in conv3's BN statistics: same function
        e = rs.randint(0, 21, size=C_).astype(np.float64)
        w[12] = (w[12] * (2.0 ** e)[None, None, None, :]).astype(np.float32)
        w[13] = (w[13] * 2.0 ** e).astype(np.float32); w[16] = (w[16] * 2.0 ** e).astype(np.float32)
        w[17] = ((w[17].astype(np.float64) + 1e-3) * 4.0 ** e - 1e-3).astype(np.float32)
    elif case == "weights_span_2^20_unstructured":             # every element of conv3's kernel x its own 2^e, e in [-10, 10]
        e = rs.randint(-10, 11, size=w[12].shape).astype(np.float64)
        w[12] = (w[12] * 2.0 ** e).astype(np.float32)
        w[17] *= 2.0 ** 16                                     # keep conv3's output at O(1): the 2^10-weights dominate the sums
    elif case == "dense_layers_rescaled":                      # fc1 outputs x 2^-14 (BN), undone by fc2's kernel rows; conv4 outputs x 2^9, undone by fc1
        w[26] *= 2.0 ** -14; w[27] *= 2.0 ** -14; w[30] *= 2.0 ** 14
        w[20] *= 2.0 ** 9; w[21] *= 2.0 ** 9; w[24] *= 2.0 ** -9
    else:
        raise KeyError(case)
    return w


@pytest.mark.parametrize("case", ["small_conv2_kernel_small_bn3_variance", "tiny_activations", "tiny_activations_everywhere",
                                  "weights_span_2^20_by_input_channel", "weights_span_2^20_by_output_channel",
                                  "weights_span_2^20_unstructured", "dense_layers_rescaled"])
@pytest.mark.parametrize("n,C_,max_batch", [(8, 256, 64), (6, 512, 4)])
def test_f16x2_scaling_holds_the_1e5_class_on_badly_scaled_networks(oz, case, n, C_, max_batch):
    """the low side of the f16x2 accuracy claim: networks whose activations or weights sit far from O(1) -- small conv
    kernels compensated by the next BN, whole layers of tiny activations, kernels spanning 2^20 by row, by column and element-wise --
    must be within 1e-5 of the float64 oracle in precision f16x2 (or raise OZ_ERR_STATE; with the commit-time scaling they all pass),
    and pass in f32."""
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.weights import init_weights
    base = init_weights(n, seed=21, channels=C_, randomize_all=True)
    for i in (36, 38):
        base[i] = base[i] * 4.0
    w = _rescaled(base, case, C_)
    B = 48
    own, opp = _boards(n, B, seed=5)
    own[0], opp[0] = own[0] & np.uint64(0), opp[0] & np.uint64(0)         # the empty board: every pixel is the all-empty pattern
    pi64, v64 = nn_numpy.forward(w, own, opp, n)
    assert np.isfinite(pi64).all() and np.isfinite(v64).all()
    ref = NNetWrapper((n, n), num_channels_1=C_, max_batch=max_batch, weights=w, precision="f32")
    p32, v32 = ref.predict_batch(own, opp)
    assert np.abs(p32.reshape(B, -1) - pi64).max() <= 1e-5 and np.abs(v32 - v64).max() <= 1e-5
    try:
        net = NNetWrapper((n, n), num_channels_1=C_, max_batch=max_batch, weights=w, precision="f16x2")
        pi, v = net.predict_batch(own, opp)
    except oz.OzError as e:                                    # a loud refusal is within the contract; a silent miss is not
        assert e.code == oz.OZ_ERR_STATE
        # one case may be refused: it is not a range problem but a CONDITIONING one (conv3's BN divides by sqrt(1e-3) behind a large
        # constant; the split's 22 bits against fp32's 24 measured 1.1e-5 against float64 here) -- the commit-time self-check against
        # the exact-fp32 kernels catches it.  Everything else must be carried by the scaling.
        assert case == "small_conv2_kernel_small_bn3_variance" and "self-check" in str(e), (case, str(e))
        return
    assert np.abs(pi.reshape(B, -1) - pi64).max() <= 1e-5, case
    assert np.abs(v - v64).max() <= 1e-5, case
    dpi, dv, npos = net.self_check()
    assert 0 <= dpi <= 8e-6 and 0 <= dv <= 8e-6 and npos in (64, 512)
    if case == "tiny_activations":
        # the exponents followed the tensor: conv2's outputs are exactly 2^-12 of the base network's, so every channel's exponent is 12 higher
        base_net = NNetWrapper((n, n), num_channels_1=C_, max_batch=max_batch, weights=base, precision="f16x2")
        assert np.array_equal(net.scaling(1), base_net.scaling(1) + 12)
        assert np.array_equal(net.scaling(0), base_net.scaling(0)) and np.array_equal(net.scaling(2), base_net.scaling(2))
        pb, vb = base_net.predict_batch(own, opp)              # ... and the function is the same one, bit for bit (all scales are exact)
        assert np.array_equal(pb, pi) and np.array_equal(vb, v)
    if case == "weights_span_2^20_by_input_channel":
        base_net = NNetWrapper((n, n), num_channels_1=C_, max_batch=max_batch, weights=base, precision="f16x2")
        pb, vb = base_net.predict_batch(own, opp)
        # the same function again; not bit for bit here: a channel that never fires on the calibration positions takes the MEDIAN exponent
        # of its tensor, which moves by another amount than that channel's own 2^e -- its (small) values round at the 2^-25 floor differently
        assert np.abs(pb - pi).max() <= 1e-6 and np.abs(vb - v).max() <= 1e-6


@pytest.mark.parametrize("precision,C_", [("f32", 128), ("f16x2", 256)])
def test_search_with_real_network_vs_oracle(oz, precision, C_):
    """end to end: batched engine + OthelloNN on the GPU == the oracle's search fed with the GPU network's
    own (pi, v) per position (so float rounding in the net cannot excuse a divergent game)."""
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    from othellozero_amd.weights import init_weights
    n, G, sims = 6, 16, 30
    w = init_weights(n, seed=2, channels=C_, randomize_all=True)
    for i in (36, 38):
        w[i] = w[i] * 4.0
    net = NNetWrapper((n, n), num_channels_1=C_, max_batch=G, weights=w, precision=precision)
    eng = SelfPlayEngine(net, n, G, sims, 1.0, 1.0, 0.9, seed=31, first_game_id=0, q_mode=1)
    rec = eng.play_to_end()
    cache = {}

    def ev(own, opp, nn):
        if (own, opp) not in cache:
            p, v = net.predict_batch([own], [opp])
            cache[(own, opp)] = (p[0].ravel(), float(v[0]))
        return cache[(own, opp)]
    for gi in range(G):
        ep = oracle.Mcts(n, 1.0, 1, evaluator=ev).episode(sims, 1.0, 0.9, 31, gi)
        r = rec[rec["game_id"] == gi]
        assert np.array_equal(r["action"], ep["action"]) and np.array_equal(r["z"], ep["z"]), gi
        assert np.array_equal(r["black"], ep["black"]) and np.array_equal(r["white"], ep["white"]), gi


# ------------------------------------------------------------------ BASELINE config sizes (size-independent properties)
def test_config2_size_properties(oz):
    """4096 concurrent 8x8 games at 100 sims/move (BASELINE config 2) for a few move rounds with the device stub
    net: every recorded transition is legal under the oracle's rules, disc counts grow by one per ply plus flips,
    counters are consistent, and a second run reproduces the first bit for bit."""
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.Othello import rules_legal_moves
    from othellozero_amd.training import SelfPlayEngine
    G, n, sims, rounds = 4096, 8, 100, 3
    states = []
    for rep in range(2):
        eng = SelfPlayEngine(StubNetWrapper((n, n), 77, 0, max_batch=G), n, G, sims, seed=1234)
        before = eng.state()
        eng.run(rounds)
        st, after = eng.stats(), eng.state()
        assert st["simulations"] == G * sims * rounds and st["moves"] == G * rounds and st["live_games"] == G
        assert st["expansions"] + st["terminal_hits"] <= st["simulations"] and st["overflow"] == 0
        assert st["expansions"] > 0.8 * st["simulations"]
        assert np.all(after["ply"] == rounds)
        discs = np.array([bin(int(b)).count("1") + bin(int(w)).count("1") for b, w in zip(after["black"], after["white"])])
        assert np.all(discs == 4 + rounds) and np.all((after["black"] & after["white"]) == 0)
        counts = eng.last_counts()
        assert np.all(counts.sum(axis=1) >= sims - 1)            # Ns of the root: this move's sims + reused visits
        states.append((after["black"].copy(), after["white"].copy(), counts.copy(), st["expansions"]))
        assert np.all(before["ply"] == 0)
    assert np.array_equal(states[0][0], states[1][0]) and np.array_equal(states[0][1], states[1][1])
    assert np.array_equal(states[0][2], states[1][2]) and states[0][3] == states[1][3]
    # spot-check 64 slots against the oracle
    for gi in range(0, G, 64):
        ep = oracle.Mcts(n, 1.0, 1, salt=77).episode(sims, 1.0, 0.9, 1234, gi, max_moves=rounds)
        b, w = C.c_uint64(int(ep["black"][-1])), C.c_uint64(int(ep["white"][-1]))
        pl, fin = C.c_int(int(ep["player"][-1])), C.c_int(0)
        oracle.lib().orc_game_play(C.byref(b), C.byref(w), n, C.byref(pl), C.byref(fin), int(ep["action"][-1]))
        assert (b.value, w.value) == (int(states[0][0][gi]), int(states[0][1][gi])), gi


def test_config4_size_properties_6x6(oz):
    """BASELINE config 4: 4096 concurrent 6x6 games, 100 sims/move, device stub net: counters, legality of the
    position after 3 move rounds (vs the oracle for a sample), bit reproducibility."""
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    G, n, sims, rounds = 4096, 6, 100, 3
    res = []
    for rep in range(2):
        eng = SelfPlayEngine(StubNetWrapper((n, n), 55, 0, max_batch=G), n, G, sims, seed=4321, q_mode=0)
        eng.run(rounds)
        st, after = eng.stats(), eng.state()
        assert st["simulations"] == G * sims * rounds and st["moves"] == G * rounds and st["overflow"] == 0
        valid = sum(1 << (r * 8 + c) for r in range(n) for c in range(n))
        assert np.all(((after["black"] | after["white"]) & ~np.uint64(valid)) == 0)
        res.append((after["black"].copy(), after["white"].copy(), eng.last_counts().copy()))
    assert all(np.array_equal(a, b) for a, b in zip(res[0], res[1]))
    for gi in range(0, G, 128):
        ep = oracle.Mcts(n, 1.0, 0, salt=55).episode(sims, 1.0, 0.9, 4321, gi, max_moves=rounds)
        b, w = C.c_uint64(int(ep["black"][-1])), C.c_uint64(int(ep["white"][-1]))
        pl, fin = C.c_int(int(ep["player"][-1])), C.c_int(0)
        oracle.lib().orc_game_play(C.byref(b), C.byref(w), n, C.byref(pl), C.byref(fin), int(ep["action"][-1]))
        assert (b.value, w.value) == (int(res[0][0][gi]), int(res[0][1][gi])), gi
        assert np.array_equal(res[0][2][gi], ep["counts"][-1]), gi


def test_config5_arena_800_sims_512_games(oz):
    """BASELINE config 5: 512 parallel deterministic arena games at 800 sims/move (net A = BLACK vs net B = WHITE,
    temperature 0, tie stream): move lists, winners and points bit-exact against the oracle for a sample of games,
    result invariants for all of them, and the whole batch reproducible."""
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.agents import arena_batch
    G, n, sims = 512, 8, 800
    r = arena_batch(StubNetWrapper((n, n), 301, 0, max_batch=G), StubNetWrapper((n, n), 302, 0, max_batch=G), n, G, sims, 1.0,
                    seed=11, first_game_id=0, q_mode=1)
    assert np.all(r["n_moves"] > 40) and np.all(r["n_moves"] <= 60)
    discs = np.array([bin(int(b)).count("1") + bin(int(w)).count("1") for b, w in zip(r["final_black"], r["final_white"])])
    assert np.all(discs == 4 + r["n_moves"]) and np.all(np.abs(r["winner"]) == 1)
    for gi in (0, 1, 255, 511):
        o = oracle.arena(oracle.Mcts(n, 1.0, 1, salt=301), oracle.Mcts(n, 1.0, 1, salt=302), sims, 11, gi)
        k = o["n_moves"]
        assert int(r["n_moves"][gi]) == k and np.array_equal(r["actions"][gi][:k], o["action"]), gi
        assert np.array_equal(r["players"][gi][:k], o["player"]), gi
        assert (int(r["winner"][gi]), int(r["points"][gi])) == (o["winner"], o["points"]), gi
        assert (int(r["final_black"][gi]), int(r["final_white"][gi])) == (o["final_black"], o["final_white"]), gi


# ------------------------------------------------------------------ BaseNN
class PyStubNetBNN(PyStubNet):
    def __init__(self, n, salt, keep, f64):
        super().__init__(n, salt, keep, f64)
        from othellozero_amd.NNet import NeuralNets
        self.network_type = NeuralNets.BNN

    def predict(self, board):
        b = np.asarray(board)
        assert b.ndim == 2
        return super().predict(np.stack([b == 1, b == -1], axis=2))


@pytest.mark.parametrize("name", ["ep6_bnn", "ep8_bnn_f64"])
def test_execute_episode_one_channel_view_vs_golden(oz, monkeypatch, name):
    """drop-in execute_episode with a BNN-typed net: one-channel boards reach predict, examples are one-channel
    per-move snapshots -- identical to the reference's trace"""
    from othellozero_amd import training
    from othellozero_amd.Othello import OthelloGame
    g = load_golden("episodes_bnn.npz")
    n, sims, seed, game, salt, keep, qmode, k = (int(x) for x in g[f"{name}/meta"])
    c, T, eg = (float(x) for x in g[f"{name}/params"])
    ply = [0]
    orig_play = OthelloGame.play

    def counting_play(self, row, col):
        orig_play(self, row, col)
        ply[0] += 1
    monkeypatch.setattr(OthelloGame, "play", counting_play)
    _patch_rng(monkeypatch, seed, game, lambda: ply[0])
    ex = training.execute_episode(n, PyStubNetBNN(n, salt, keep, qmode == 1), int(c), sims, int(T), eg, q_mode=qmode)
    assert len(ex) == 8 * k
    for i, (b, p, z) in enumerate(ex):
        assert b.ndim == 2 and str(b.dtype) == str(g[f"{name}/dtype"][0])
        assert oracle.pack_board(np.stack([b == 1, b == -1], axis=2)) == tuple(int(x) for x in g[f"{name}/ex_board"][i]), (name, i)
        assert int(np.argmax(p)) == int(g[f"{name}/ex_policy"][i]) and z == int(g[f"{name}/ex_z"][i])


@pytest.mark.parametrize("precision,channels", [("f32", 128), ("f16x2", 256)])
def test_basenn_network_vs_float64_oracle(oz, precision, channels):
    """NNetWrapper(network=BNN): conv1 on ONE plane (+1 mover / -1 opponent), rest of the trunk shared; <= 1e-5"""
    from othellozero_amd.NNet import NNetWrapper, NeuralNets
    from othellozero_amd.weights import init_weights
    n, batch = 8, 21
    w = init_weights(n, seed=13, channels=channels, randomize_all=True, in_channels=1)
    for i in (36, 38):
        w[i] = w[i] * 4.0
    net = NNetWrapper((n, n), num_channels_1=channels, max_batch=32, weights=w, network=NeuralNets.BNN, precision=precision)
    own, opp = _boards(n, batch, seed=77)
    pi, v = net.predict_batch(own, opp)
    pi64, v64 = nn_numpy.forward(w, own, opp, n)
    assert pi64.std() > 2e-3
    assert np.abs(pi.reshape(batch, -1) - pi64).max() <= 1e-5 and np.abs(v - v64).max() <= 1e-5
    brd = oz.unpack_board(int(own[2]), int(opp[2]), n)
    p1, v1 = net.predict(brd[:, :, 0].astype(np.int64) - brd[:, :, 1].astype(np.int64))       # one-channel input
    assert np.array_equal(p1, pi[2]) and v1 == v[2]
    assert net.get_weights()[0].shape == (3, 3, 1, channels)


@pytest.mark.gpu
def test_pingpong_conv_loop_bit_identical_to_simple_loop():
    """the 4-phase ping-pong main loop (default) and the one-barrier-per-k-tile loop (oz_net_set_option OZ_NET_OPT_SIMPLE_LOOP) accumulate every
    output in the same order: (pi, v) must be bit-identical over batch sizes / boards / repetitions -- a LDS-DMA
    visibility race in the ping-pong schedule would show up as a mismatch (tools/pp_race_check.py)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "pp_race_check.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 differ" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f32", "f16x2"])
@pytest.mark.parametrize("n,network", [(8, "ONN"), (6, "ONN"), (8, "BNN")])
def test_small_network_latency_path_vs_float64(oz, n, network, precision):
    """networks with max_batch <= 32 (the drop-in single-position path) split every GEMM's k loop 16 ways: same 1e-5
    tolerance against float64, invariance to the size of the call, agreement with a throughput-path twin to rounding"""
    from othellozero_amd.NNet import NNetWrapper, NeuralNets
    from othellozero_amd.weights import init_weights
    cin = 2 if network == "ONN" else 1
    w = init_weights(n, seed=21, channels=512, randomize_all=True, in_channels=cin)
    for i in (36, 38):
        w[i] = w[i] * 4.0
    kind = NeuralNets.ONN if network == "ONN" else NeuralNets.BNN
    small = NNetWrapper((n, n), num_channels_1=512, max_batch=4, weights=w, precision=precision, network=kind)
    own, opp = _boards(n, 7, seed=3 * n)
    pi, v = small.predict_batch(own, opp)                       # chunks of 4, 3
    pi64, v64 = nn_numpy.forward(w, own, opp, n)
    assert np.abs(pi.reshape(7, -1) - pi64).max() <= 1e-5 and np.abs(v - v64).max() <= 1e-5
    p1, v1 = small.predict_batch(own[5:6], opp[5:6])            # a call of another size: bit-identical
    assert np.array_equal(p1[0], pi[5]) and v1[0] == v[5]
    big = NNetWrapper((n, n), num_channels_1=512, max_batch=64, weights=w, precision=precision, network=kind)
    pb, vb = big.predict_batch(own, opp)
    assert np.abs(pb - pi).max() <= 2e-6 and np.abs(vb - v).max() <= 2e-6


@pytest.mark.gpu
def test_game_sharding_does_not_change_the_pooled_records(oz):
    """BASELINE configs[2] / [3] in miniature: a job of 24 games played by ONE engine, and the same job sharded over 2, 3, 4 and 8
    'ranks' (engines owning contiguous blocks of global game ids, as bench.py / distributed.shard_games assign them) give the same
    multiset of move records -- with refills, too (second generation ids continue with the job-wide stride)"""
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    from othellozero_amd.distributed import shard_games
    n, G, sims = 6, 24, 12
    def run(num, first, stride, rounds):
        eng = SelfPlayEngine(StubNetWrapper((n, n), 61, 0, max_batch=num), n, num, sims, 1.0, 1.0, 0.9, seed=5, first_game_id=first,
                             game_id_stride=stride, refill=stride > 0, record_cap=num * 4 * n * n)
        eng.run(rounds)
        return eng.records()
    whole = run(G, 0, 0, n * n)
    for ways in (2, 3, 4, 8):                                  # SURVEY section 4, tier 5: the job sharded 1 / 2 / 4 / 8 ways (and 3: ragged shards)
        parts = [run(cnt, first, 0, n * n) for first, cnt in (shard_games(G, r, ways) for r in range(ways))]
        pooled = np.concatenate(parts)
        pooled = pooled[np.lexsort((pooled["ply"], pooled["game_id"]))]
        assert len(whole) == len(pooled) and whole.tobytes() == pooled.tobytes(), ways
    # with refill: 2 generations; a finished slot restarts as game id + job-wide stride
    whole2 = run(G, 0, G, 70)
    parts2 = [run(cnt, first, G, 70) for first, cnt in (shard_games(G, r, 3) for r in range(3))]
    pooled2 = np.concatenate(parts2)
    done_w = {int(g) for g in np.unique(whole2["game_id"])}
    done_p = {int(g) for g in np.unique(pooled2["game_id"])}
    common = sorted(done_w & done_p)
    assert len(common) >= G                                    # at least the whole first generation
    for gid in common[:G + 4]:
        a = whole2[whole2["game_id"] == gid]
        b = pooled2[pooled2["game_id"] == gid]
        b = b[np.argsort(b["ply"])]
        assert a.tobytes() == b.tobytes(), gid


@pytest.mark.gpu
@pytest.mark.parametrize("n,sims,T,keep,cap", [(6, 20, 1.0, 0, 0), (8, 12, 1.0, 0, 0), (6, 9, 0.0, 0x3, 0), (8, 30, 1.0, 0, 0),
                                              (6, 20, 1.0, 0, 29), (8, 12, 1.0, 0x3, 40), (8, 30, 1.0, 0, 11)])
def test_free_running_selfplay_equals_lock_step(oz, n, sims, T, keep, cap):
    """oz_selfplay_run_steps (every game runs on by itself; full leaf batches) produces exactly the move records of the
    lock-step oz_selfplay_run -- first generation and refilled games -- and needs fewer network batches; with a batch cap
    (leaves that find no slot wait for the next batch, rotating slot order) the records are still those"""
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    G = 48
    def make():
        return SelfPlayEngine(StubNetWrapper((n, n), 71, keep, max_batch=G), n, G, sims, 1.0, T, 0.9, seed=13, first_game_id=100,
                              game_id_stride=G, refill=True, record_cap=G * 6 * n * n)
    lock = make()
    rounds = 2 * (n * n - 4) + 4                               # two generations
    lock.run(rounds)
    rl = lock.records()
    free = make()
    if cap:
        free.set_batch_cap(cap)
    steps = 0
    while True:
        free.run_steps(50)
        steps += 50
        st = free.stats()
        if st["games_completed"] >= 2 * G or steps > rounds * sims * (2 if not cap else 2 * G // cap + 2):
            break
    rf = free.records()
    both = sorted(set(int(x) for x in np.unique(rl["game_id"])) & set(int(x) for x in np.unique(rf["game_id"])))
    assert len(both) >= G + G // 2, (len(both), steps)
    for gid in both:
        a, b = rl[rl["game_id"] == gid], rf[rf["game_id"] == gid]
        assert a.tobytes() == b.tobytes(), gid
    sl, sf = lock.stats(), free.stats()
    assert sf["overflow"] == 0 and sl["overflow"] == 0
    # the free-running driver wastes no batch slot on network-free simulations
    ev = free.eval_time()
    if cap:
        assert sf["leaves_evaluated"] <= cap * ev["launches"]                # no batch above the cap
    else:
        assert sf["expansions"] / max(ev["launches"], 1) > 0.9 * G or sf["live_games"] < G


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["stub", "onn"])
def test_cross_game_leaf_dedup_changes_nothing_but_the_work(oz, kind):
    """k_compact evaluates a board that several games reach in the same step once (default) -- records, visit counts and
    per-game statistics are those of one evaluation per game (oz_selfplay_config.dedup = OZ_DEDUP_OFF), and the opening plies cost far fewer evaluations"""
    from othellozero_amd.NNet import NNetWrapper, StubNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    n, G, sims = 6, 96, 12
    def run(dedup):
        net = (StubNetWrapper((n, n), 5, 0, max_batch=G) if kind == "stub" else
               NNetWrapper((n, n), num_channels_1=256, max_batch=G, seed=4, precision="f16x2"))
        eng = SelfPlayEngine(net, n, G, sims, 1.0, 1.0, 0.9, seed=77, first_game_id=0, game_id_stride=G, refill=False,
                             record_cap=G * n * n, dedup=dedup)
        eng.run(n * n)
        return eng.records(), eng.stats(), eng.last_counts()
    r1, s1, c1 = run(True)
    r0, s0, c0 = run(False)
    assert r1.tobytes() == r0.tobytes() and np.array_equal(c1, c0)
    for k in ("simulations", "node_visits", "expansions", "terminal_hits", "moves", "games_completed"):
        assert s1[k] == s0[k], k
    assert s0["leaves_evaluated"] == s0["expansions"]
    assert s1["leaves_evaluated"] < s1["expansions"]             # all games share the opening positions
    assert s1["overflow"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f16x2", "f32"])
@pytest.mark.parametrize("n,cin,C,B", [(8, 2, 512, 130), (6, 2, 256, 70), (8, 1, 256, 40), (6, 1, 512, 33)])
def test_conv_pattern_tables_vs_float64(oz, n, cin, C, B, precision):
    """precision f16x2 evaluates conv1 + conv2 as lookups in tables indexed by the 3^9 neighbourhood patterns of the
    discrete input planes (k_lut_ids / k_conv2_lut, tables built at commit): within 1e-5 of the float64 oracle on random
    boards AND on the corner cases of the pattern index (empty board, one colour everywhere, full board, single discs on
    edges and corners), for both networks and both board sizes; a position's result does not depend on the batch"""
    from othellozero_amd.NNet import NNetWrapper, NeuralNets
    from othellozero_amd.weights import init_weights
    w = init_weights(n, seed=31 + n + cin, channels=C, randomize_all=True, in_channels=cin)
    net = NNetWrapper((n, n), num_channels_1=C, max_batch=B, precision=precision, weights=w,
                      network=NeuralNets.ONN if cin == 2 else NeuralNets.BNN)
    rs = np.random.RandomState(1000 * n + cin)
    valid = np.uint64(sum(1 << (r * 8 + c) for r in range(n) for c in range(n)))
    own = rs.randint(0, 2**63, size=B, dtype=np.uint64) & valid
    opp = rs.randint(0, 2**63, size=B, dtype=np.uint64) & valid & ~own
    bit = lambda r, c: np.uint64(1 << (r * 8 + c))
    special = [(0, 0), (valid, 0), (0, valid), (own[0], valid & ~own[0]),
               (bit(0, 0), bit(n - 1, n - 1)), (bit(0, n - 1), bit(n - 1, 0)), (bit(n // 2, 0), bit(0, n // 2)),
               (bit(n - 1, n // 2) | bit(n // 2, n - 1), 0)]
    for i, (o, p) in enumerate(special):
        own[i + 1], opp[i + 1] = np.uint64(o), np.uint64(p)
    pi, v = net.predict_batch(own, opp)
    pi64, v64 = nn_numpy.forward(w, own, opp, n)
    assert np.abs(pi.reshape(B, -1) - pi64).max() <= 1e-5 and np.abs(v - v64).max() <= 1e-5
    order = rs.permutation(B)[: B // 2]                          # another batch: other slots, other neighbours, other size
    p2, v2 = net.predict_batch(own[order], opp[order])
    assert np.array_equal(p2, pi[order]) and np.array_equal(v2, v[order])
    net.set_tables(0)                                            # conv1 kernel + conv2 GEMM: the same network, another summation order
    p0, v0 = net.predict_batch(own, opp)
    assert np.abs(p0 - pi).max() <= 2e-6 and np.abs(v0 - v).max() <= 2e-6
    assert net.profiled_layer() == 2
    net.set_tables(-1)
    p3, v3 = net.predict_batch(own, opp)
    assert np.array_equal(p3, pi) and np.array_equal(v3, v) and net.profiled_layer() == 3


@pytest.mark.gpu
@pytest.mark.parametrize("kind,driver", [("stub", "lockstep"), ("onn", "lockstep"), ("stub", "free"), ("onn", "free_capped")])
def test_persistent_evaluation_cache_changes_nothing_but_the_work(oz, kind, driver):
    """oz_selfplay_config.eval_cache: leaves whose board the network has evaluated before -- in an earlier batch, another game, a game that
    ended long ago -- take (pi, v) from the network's HBM cache (the reference's _predict_cache, othelo_mcts.py:82-88, generalised).
    Records, visit counts and per-game statistics are those of the engine without it; far fewer positions reach the network; new
    weights (oz_net_commit) empty the cache"""
    from othellozero_amd.NNet import NNetWrapper, StubNetWrapper
    from othellozero_amd.training import SelfPlayEngine
    from othellozero_amd.weights import init_weights
    n, G, sims = 6, 96, 12

    def network():
        return (StubNetWrapper((n, n), 5, 0, max_batch=G) if kind == "stub" else
                NNetWrapper((n, n), num_channels_1=256, max_batch=G, seed=4, precision="f16x2"))

    def run(net, cache, dedup=True):
        eng = SelfPlayEngine(net, n, G, sims, 1.0, 1.0, 0.9, seed=77, first_game_id=0, game_id_stride=G, refill=True,
                             record_cap=G * 4 * n * n, dedup=dedup, eval_cache=cache, batch_cap=64 if driver == "free_capped" else 0)
        if driver == "lockstep":
            eng.run(2 * n * n)
        else:
            eng.run_steps(2 * n * n * sims)
        return eng.records(), eng.stats(), eng.last_counts()
    net = network()
    r0, s0, c0 = run(net, False)
    net.set_eval_cache(1 << 16)
    r1, s1, c1 = run(net, True)
    st = net.eval_cache_stats()
    def same_games(ra, rb):
        """records of the games both runs completed, bit for bit (under a batch cap a hit frees a slot for another leaf: the cached run gets
        further in the same number of batches, so it completes at least the games of the other)"""
        ids_a, ids_b = set(int(x) for x in np.unique(ra["game_id"])), set(int(x) for x in np.unique(rb["game_id"]))
        both = sorted(ids_a & ids_b)
        assert len(both) >= 0.7 * max(len(ids_a), len(ids_b)) and len(both) >= G
        return ra[np.isin(ra["game_id"], both)].tobytes() == rb[np.isin(rb["game_id"], both)].tobytes()
    if driver == "free_capped":
        assert same_games(r0, r1) and s1["games_completed"] >= s0["games_completed"]
    else:
        assert r1.tobytes() == r0.tobytes() and np.array_equal(c1, c0)
        for k in ("simulations", "node_visits", "expansions", "terminal_hits", "moves", "games_completed"):
            assert s1[k] == s0[k], k
        assert s1["leaves_evaluated"] < s0["leaves_evaluated"]
    assert st["hits"] > 0 and 0.9 * s1["leaves_evaluated"] < st["inserts"] <= s1["leaves_evaluated"]
    # a second engine on the same network starts with a warm cache: (almost) every position of the same games is a hit; still the same records
    r2, s2, _ = run(net, True, dedup=False)
    assert same_games(r0, r2) and s2["leaves_evaluated"] < 0.25 * s0["leaves_evaluated"]
    if kind == "onn":
        # new weights: the cache is emptied by the commit -- the engine must reproduce a fresh network's games, not serve stale (pi, v)
        w = init_weights(n, seed=9, channels=256)
        net.set_weights(w)
        assert net.eval_cache_stats()["entries"] >= 1 << 16
        r3, _, _ = run(net, True)
        fresh = NNetWrapper((n, n), num_channels_1=256, max_batch=G, weights=w, precision="f16x2")
        r4, _, _ = run(fresh, False)
        assert same_games(r4, r3) and not same_games(r0, r3)
    # a tiny cache (one bucket row per 1024 positions): constant replacement, still exact
    net2 = network()
    net2.set_eval_cache(1)
    r5, _, _ = run(net2, True)
    assert same_games(r0, r5)


@pytest.mark.gpu
def test_c_abi_exchange_step_over_rccl_on_one_rank(oz):
    """oz_comm_* / oz_selfplay_gather_records: the path's one collective behind the C ABI, RCCL bound at run time.  One GPU here, so a
    one-rank communicator: id, communicator, counts all-gather, padded record all-gather -- the pooled records equal the engine's own
    (the 8-rank run is the driver's; bench.py --gpus N repeats this check against the torch.distributed pool after its line)"""
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.distributed import Comm
    from othellozero_amd.training import SelfPlayEngine
    n, G = 6, 32
    eng = SelfPlayEngine(StubNetWrapper((n, n), 3, 0, max_batch=G), n, G, 8, 1.0, 1.0, 0.9, seed=5, refill=True, record_cap=G * 4 * n * n)
    comm = Comm(0, 1)
    rec, per = comm.gather_records(eng)                       # nothing completed yet: an empty gather must work
    assert rec.size == 0 and per.tolist() == [0]
    eng.run(n * n)
    own = eng.records()
    rec, per = comm.gather_records(eng)
    assert per.tolist() == [own.size] and own.size > G * 20
    assert rec[np.lexsort((rec["ply"], rec["game_id"]))].tobytes() == own.tobytes()
    eng.run(8)
    more = eng.stats()["records"]
    tail, per = comm.gather_records(eng, first_record=own.size)   # only what completed since
    assert per.tolist() == [more - own.size] and tail.size == more - own.size
    # the counts-only form (out = NULL, room 0) and the room check, which is decided from the gathered (count, room) pairs -- on every rank alike
    lib = oz.load()
    written, cnt = C.c_int64(), np.zeros(1, np.int64)
    oz.check(lib.oz_selfplay_gather_records(eng._h, comm._h, 0, None, 0, C.byref(written), oz.p_i64(cnt)))
    assert written.value == more and cnt.tolist() == [more]
    small = np.zeros(more - 1, dtype=oz.RECORD_DTYPE)
    rc = lib.oz_selfplay_gather_records(eng._h, comm._h, 0, small.ctypes.data_as(C.c_void_p), small.size, C.byref(written), oz.p_i64(cnt))
    assert rc == oz.OZ_ERR_ARG and "room" in lib.oz_last_error().decode() and written.value == 0 and cnt.tolist() == [more]
    exact, per = comm.gather_records(eng, max_records=more)       # exactly enough room
    assert exact.size == more
    # a rank whose own arguments are bad still JOINS the pair all-gather (room = -2, "cannot take part") and fails afterwards, so that its
    # peers fail with it instead of blocking in the collective; the communicator stays usable
    rc = lib.oz_selfplay_gather_records(eng._h, comm._h, -1, small.ctypes.data_as(C.c_void_p), small.size, C.byref(written), oz.p_i64(cnt))
    assert rc == oz.OZ_ERR_ARG and "first_record" in lib.oz_last_error().decode() and written.value == 0
    rc = lib.oz_selfplay_gather_records(eng._h, comm._h, 0, None, 5, C.byref(written), oz.p_i64(cnt))
    assert rc == oz.OZ_ERR_ARG and "null output buffer" in lib.oz_last_error().decode()
    again, per = comm.gather_records(eng, max_records=more)
    assert again.tobytes() == exact.tobytes()
    comm.close()


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal(tmp_path):
    """bench.py's N > 1 control flow (sharded game ids, barrier, max-over-ranks time, all-gather of records, one JSON line
    on rank 0) with two ranks on this one GPU over gloo -- the driver runs the real thing over RCCL on 8 GPUs"""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "36", "--warmup", "1",
           "--board", "6", "--games", "64", "--sims", "6", "--backend", "gloo", "--same-device",
           "--driver", "lockstep"]                          # 6x6 games end within the run; the other multi-rank test runs the default driver
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 36 and out["scaling"] == "weak" and out["value"] > 0
    assert len(out["per_rank_records"]) == 2 and sum(out["per_rank_records"]) == out["pooled_records"] and out["gather_ms"] > 0
    # records leave the engines when a game ends: both ranks' finished games (>= 28 moves each) are in the pooled tensor
    assert out["games_completed"] >= 2 * 48 and out["pooled_records"] >= 28 * out["games_completed"]
    assert "cpu_baseline" not in out and "cross_game_dedup" not in out
    assert out["roofline"]["bound"] == "mfma" and 0 < out["roofline"]["frac"] < 1


@pytest.mark.gpu
def test_bench_four_ranks_rehearsal(tmp_path):
    """the multi-rank control flow of `python bench.py --gpus N` at the widest this box allows: FOUR ranks on the one card (the GPU boxes refuse
    more than six processes on a card and the test runner is one of them -- a six-rank attempt was killed by the box's process guard -- so the
    8-rank job itself stays the driver's; the 8-way split of the exchange step is covered on the CPU by
    test_gloo_world_size_8_ragged_and_empty_ranks, the 8-way sharding of the games by test_game_sharding_does_not_change_the_pooled_records).
    bench.py launches its own ranks (file-store rendezvous, gloo); the line must carry one entry per rank in per_rank_records, their sum must be
    the pooled count, every rank must have contributed, and the aggregate must be the four ranks' work"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--same-device", "--backend", "gloo", "--steps", "30", "--warmup", "1",
                        "--games", "96", "--sims", "8", "--board", "6", "--channels", "256"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["rccl_ranks"] == 0 and out["scaling"] == "weak" and out["value"] > 0
    assert len(out["per_rank_records"]) == 4 and all(x > 0 for x in out["per_rank_records"])
    assert sum(out["per_rank_records"]) == out["pooled_records"] and out["gather_ms"] > 0
    # 4 x 96 games x 30 steps of 8 simulations: every rank's slots complete games inside the window
    assert out["games_completed"] >= 4 * 24 and out["simulations"] >= 4 * 96 * 30 * 8 * 0.9
    assert "cpu_baseline" not in out and "cross_game_dedup" not in out


@pytest.mark.gpu
def test_bench_single_rank_contract(tmp_path):
    """bench.py prints ONE strict-JSON line of < 6000 bytes with the contract's fields first (the driver keeps an 8 KB tail: round 5's 33 KB
    line was recorded as unparsed); the top-level value is measured in exact fp32 (the reference's arithmetic), the other precisions ride
    beside it as scalars; everything else is in bench_detail.json"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--games", "256", "--sims", "16",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip().endswith(lines[0])          # the line is the LAST thing on stdout
    assert len(lines[0]) < 6000, len(lines[0])

    def no_constants(name):
        raise ValueError(f"non-strict JSON constant {name}")
    out = json.loads(lines[0], parse_constant=no_constants)
    keys = list(out)
    assert keys[:13] == ["metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                         "dtype", "data"] and keys[13:15] == ["config", "roofline"]
    assert out["metric"] == "mcts_node_expansions_per_sec" and out["n_gpus"] == 1 and out["steps"] == 3 and out["vs_baseline"] is None
    assert "workload" in out["config"] and "model" not in out["config"] and out["config"]["precision"] == "bf16x3"
    # on top: fp32 values carried exactly as three bf16 planes, six bf16 MFMA products per fp32 product (fp32-class, 2.67x the fp32 matrix roof) ...
    assert out["dtype"] == "f32 (3xbf16 split)"
    rf = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms", "kernel"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["peak"] == 2500.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["launches"] == 3 * 16
    assert 0 < rf["frac"] < 1 / 6 and "k_gemm_b3" in rf["kernel"] and rf["mfma_products_per_fp32_product"] == 6 and 0 < rf["matrix_pipe_frac"] < 1
    # ... and the reference's literal arithmetic beside it: exact fp32 on the fp32 matrix cores, same workload, same run, same steps
    assert out["value_f32"] > 0 and out["dtype_f32"] == "f32" and out["roofline_f32"]["peak"] == 157.3 and 0 < out["roofline_f32"]["frac"] < 1
    # roofline.traffic is measured in the run (two rocprofv3 --pmc child passes); the committed profile is only the fallback
    assert rf["traffic_measured_in_this_run"] is True and rf["traffic"] > 0
    # every expansion is evaluated in the timed region
    assert out["leaves_evaluated_rank0"] == out["expansions"]
    assert out["games_completed"] > 0 and out["games_per_s"] > 0 and out["pooled_records"] >= 40 * out["games_completed"]
    assert out["flop_per_expansion"]["executed"] < out["flop_per_expansion"]["reference_network"]
    assert 0 < out["whole_path_frac"] < 1 and out["whole_path_frac_reference_flop"] > out["whole_path_frac"]
    # the other precisions of the same workload ride beside it: f16x2 (fp32-equivalent on the fp16 matrix cores)
    assert out["value_f16x2"] > 0 and out["dtype_f16x2"].startswith("f32 (2xf16") and out["roofline_f16x2"]["peak"] == 2500.0
    assert 0 < out["roofline_f16x2"]["frac"] < 0.334
    assert out["config4_value"] > 0 and out["value_other_driver"] > 0 and out["value_with_cross_game_dedup"] > 0 and out["value_all_layers_as_gemm"] > 0
    cal = out["device_calibration"]
    assert 500 < cal["f16_sustained_tflops"] < 2600 and 50 < cal["f32_sustained_tflops"] < 165 and 0 < cal["dominant_kernel_share_of_sustained"] < 1
    assert out["per_rank_records"] == [out["pooled_records"]] and len(out["per_rank_ms_per_step"]) == 1

    # ---- the detail file: kernels[], the nested legs, notes
    det = json.load(open(out["detail"]))
    assert det["value"] == out["value"] and det["roofline"]["frac"] == rf["frac"]
    assert "measured in this run" in (det["roofline"]["traffic_source"] or ""), det["roofline"].get("live_traffic_error")
    assert det["cross_game_dedup"]["leaves_evaluated"] < det["cross_game_dedup"]["expansions"]
    assert det["slot_ply_spread_rank0"][1] - det["slot_ply_spread_rank0"][0] >= 40
    p16 = det["precisions"]["f16x2"]
    assert p16["value"] == pytest.approx(out["value_f16x2"], rel=1e-5) and p16["steps"] == 3 and p16["roofline"]["peak"] == 2500.0
    p32 = det["precisions"]["f32"]
    assert p32["value"] == pytest.approx(out["value_f32"], rel=1e-5) and p32["steps"] == 3 and p32["roofline"]["peak"] == 157.3 and "k_gemm_f32" in p32["roofline"]["kernel"]
    assert "measured in this run" in (p32["roofline"]["traffic_source"] or ""), p32["roofline"].get("live_traffic_error")
    # the 6x6 config carries its own roofline (conv3 of the 6x6 network, HIP events in ITS timed steps), kernels[] and the other precisions' rates
    c4 = det["config4"]
    assert c4["roofline"]["bound"] == "mfma" and c4["roofline"]["launches"] == 3 * 16 and 0 < c4["roofline"]["frac"] < 1 and c4["roofline"]["flop_per_leaf"] == 2 * 16 * 4608 * 512
    assert c4["roofline"]["peak"] == 2500.0 and {"conv3", "conv4", "fc1", "select"} <= {k["name"] for k in c4["kernels"]}
    assert c4["precisions"]["f16x2"]["value"] > 0 and c4["precisions"]["f32"]["roofline"]["peak"] == 157.3
    pr = det["per_rank"]
    assert len(pr["ms_per_step"]) == 1 and pr["ms_per_step_min"] == pr["ms_per_step_max"] <= det["ms_per_step"] and pr["expansions"] == [det["expansions"]]
    assert det["config"]["driver"] == "free" and det["other_driver"]["driver"] == "lockstep" and det["other_driver"]["value"] > 0
    names = [k["name"] for k in det["kernels"]]
    for k in ("conv2", "conv3", "conv4", "fc1", "fc2", "heads", "select", "compact", "expand_backup"):      # (free-running driver: moves ride in "select")
        assert k in names, k
    assert all(k["ms_per_step"] > 0 and 0 < k["frac"] < 1.5 and k["bound"] for k in det["kernels"])
    assert abs(sum(k["ms_per_step"] for k in det["kernels"]) / det["ms_per_step"] - 1) < 0.5


@pytest.mark.gpu
def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO outer launcher starts its two ranks itself (child processes with RANK / LOCAL_RANK /
    WORLD_SIZE set; the parent never touches the GPU) and relays rank 0's line with n_gpus == 2 -- the role of
    workers.py:168-184,298-303.  Rehearsed on this one GPU over gloo; the driver runs the real thing over RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "1", "--board", "6",
                        "--games", "64", "--sims", "6", "--backend", "gloo", "--same-device"], capture_output=True, text=True, timeout=600,
                       cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 0 and out["config"]["backend"] == "gloo"
    assert out["games_completed"] > 0 and out["pooled_records"] >= 28 * out["games_completed"] and out["value"] > 0
    # the exchange step is diagnosable from the line: what every rank contributed, that nothing was lost, and what the gather cost
    assert len(out["per_rank_records"]) == 2 and min(out["per_rank_records"]) > 0
    assert sum(out["per_rank_records"]) == out["pooled_records"] and out["gather_ms"] > 0


@pytest.mark.gpu
def test_bench_rank_names_its_phase_when_a_peer_never_arrives(tmp_path):
    """a rank of a 2-rank world whose peer never starts: the rendezvous times out (OZ_BENCH_COLLECTIVE_TIMEOUT), the rank prints WHICH
    rank failed in WHICH phase and exits non-zero without a JSON line -- the first real multi-rank RCCL run must be diagnosable"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", OZ_BENCH_INIT_FILE=str(tmp_path / "rendezvous"),
               OZ_BENCH_COLLECTIVE_TIMEOUT="5", OZ_BENCH_TIMEOUT="60")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--board", "6", "--games", "64", "--sims", "6",
                        "--backend", "gloo", "--same-device"], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "rank 0" in r.stderr and "phase 'init'" in r.stderr, r.stderr[-1500:]


@pytest.mark.gpu
def test_worker_side_driver_functions_behave_like_the_reference(oz):
    """training.duel_between_neural_networks / evaluate_neural_network (training.py:75-118; workers.py:14-15 imports them next
    to execute_episode): the reference hands the (agent, points) tuple of duel_between_agents on as if it were the agent --
    the first raises KeyError, the second never counts a win (tests/golden/drivers_misc.json, written by running the
    reference).  The mirror does the same by default; fixed=True gives the evident intent."""
    import json
    import os
    from othellozero_amd import training
    from othellozero_amd.agents import RandomOthelloAgent
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "drivers_misc.json")))
    n, sims = 4, 6
    a, b = PyStubNet(n, 5, 0, True), PyStubNet(n, 6, 0, True)
    assert want["duel_between_neural_networks"] == {"raises": "KeyError"}
    with pytest.raises(KeyError):
        training.duel_between_neural_networks(n, a, b, 1, sims)
    assert training.duel_between_neural_networks(n, a, b, 1, sims, fixed=True) in (0, 1)
    random.seed(2); np.random.seed(2)
    assert training.evaluate_neural_network(n, want["evaluate_neural_network"]["iterations"], a, sims, 1, RandomOthelloAgent, ()) \
        == want["evaluate_neural_network"]["returns"] == 0
    random.seed(2); np.random.seed(2)
    wins = training.evaluate_neural_network(n, 5, a, sims, 1, RandomOthelloAgent, (), fixed=True)
    assert 0 <= wins <= 5


@pytest.mark.gpu
def test_mcts_template_hooks_and_geometry_helpers(oz, golden_rules):
    """the MCTS template hooks OthelloMCTS overrides in the reference (othelo_mcts.py:28-49,69-88) and OthelloGame's ray
    helpers (Othello/__init__.py:186-198) answer like the search kernels do: next state / terminal / reward against the
    golden transitions, the masked policy against the expanded node's P row"""
    from othellozero_amd.Othello import BoardView, OthelloGame, OthelloPlayer
    from othellozero_amd.othelo_mcts import OthelloMCTS
    n = 6
    net = PyStubNet(n, 3, 0, True)
    m = OthelloMCTS(n, net, 1.0, q_mode=1)
    g = OthelloGame(n)
    state = g.board(BoardView.TWO_CHANNELS)
    acts = m.get_state_actions(state)
    assert acts == [tuple(int(x) for x in a) for a in OthelloGame.get_player_valid_actions(state, OthelloPlayer.BLACK)]
    assert not m.is_terminal_state(state) and m.get_state_reward(state) in (1, -1)
    nxt = m.get_next_state(state, acts[0])
    assert nxt.shape == state.shape and nxt.sum() == state.sum() + 1 and not np.shares_memory(nxt, state)
    # the opponent can move after the first ply: channels swapped, so the mover's discs (channel 0) are the 1 disc WHITE kept
    assert nxt[:, :, 0].sum() == 1 and nxt[:, :, 1].sum() == 4
    p = m.moves_scaled_by_valid_moves(state)
    mask = m._mask_valid_moves(state)
    assert p.shape == (n, n) and np.all((p > 0) <= (mask > 0)) and mask.sum() == len(acts)
    assert net.calls == 1 and m.get_state_value(state) == net.predict(state)[1] and net.calls == 2     # the second came from this line, not the cache miss
    m.simulate(state, OthelloPlayer.BLACK)
    root = m.dump()[0]
    assert np.allclose(root["P"].reshape(8, 8)[:n, :n], p / p.sum(), rtol=0, atol=0)                   # the expanded node's P row = normalised masked policy
    rays = [list(r) for r in OthelloGame.get_all_directions_squares(n, 2, 3)]
    assert len(rays) == 8 and rays[0] == [(3, 4), (4, 5)] and rays[5] == [(1, 3), (0, 3)]
    assert g.is_square_free(0, 0) and not g.is_square_free(n // 2, n // 2)


@pytest.mark.gpu
def test_thread_workers_share_one_network_like_the_reference(oz):
    """workers.py:33-37,82-90 + main.py:353-354: N ThreadWorkers call execute_episode concurrently, each with its own OthelloMCTS, all
    sharing ONE NNetWrapper.  The drop-in must be thread-safe: with e_greedy = 1 an episode is a pure function of the network (the coin is
    drawn but cannot change the move), so every thread must return exactly the examples of a serial run -- and the ctypes calls release
    the GIL, so the threads really are inside the library at the same time."""
    import threading
    from othellozero_amd import training
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.weights import init_weights
    n, C_, sims, workers = 6, 128, 10, 6
    net = NNetWrapper((n, n), num_channels_1=C_, max_batch=1, weights=init_weights(n, seed=11, channels=C_, randomize_all=True))
    want = training.execute_episode(n, net, 1.0, sims, 1.0, 1.0, snapshot_boards=True)
    assert len(want) % 8 == 0 and len(want) >= 8 * 20
    got, errors = [None] * workers, []

    def work(i):
        try:
            got[i] = training.execute_episode(n, net, 1.0, sims, 1.0, 1.0, snapshot_boards=True)
        except Exception as e:              # noqa: BLE001 -- reported below, in the main thread
            errors.append((i, repr(e)))
    threads = [threading.Thread(target=work, args=(i,)) for i in range(workers)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a worker thread is stuck inside the library"
    assert not errors, errors
    for i in range(workers):
        assert len(got[i]) == len(want)
        for (b0, p0, z0), (b1, p1, z1) in zip(want, got[i]):
            assert z0 == z1 and np.array_equal(b0, b1) and np.array_equal(p0, p1)


@pytest.mark.gpu
def test_reference_conventions_inputs_untouched_and_error_types(oz, tmp_path):
    """SURVEY.md 8(b) conventions: the search never mutates the state it is given (np.copy at othelo_mcts.py:23,44), OthelloGame.play
    mutates in place and refuses a finished game with AssertionError('Game has ended') (Othello/__init__.py:143), load_checkpoint asserts
    the .h5 extension (Net/NNet.py:95), board() refuses a non-BoardView (TypeError), N() of an unknown state is 0 and of an expanded but
    never selected state raises KeyError (MCTS/__init__.py:73-84,172-175)."""
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.Othello import BoardView, OthelloGame, OthelloPlayer
    from othellozero_amd.othelo_mcts import OthelloMCTS
    n = 6
    net = PyStubNet(n, 3, 0, True)
    g = OthelloGame(n)
    state = g.board(BoardView.TWO_CHANNELS)
    assert state is g.board(BoardView.TWO_CHANNELS)                       # the live array, no copy (Othello/__init__.py:77-78)
    before = state.copy()
    m = OthelloMCTS(n, net, 1.0, q_mode=1)
    assert m.N(state) == 0                                                # unknown state
    m.simulate(state, OthelloPlayer.BLACK)                                # expands the root only
    with pytest.raises(KeyError):
        m.N(state, (0, 0))
    for _ in range(12):
        m.simulate(state, OthelloPlayer.BLACK)
    m.get_policy_action_probabilities(state, 1)
    assert np.array_equal(state, before), "the search wrote into the caller's state"
    r, c = m.get_state_actions(state)[0]
    g.play(r, c)
    assert not np.array_equal(state, before) and state is g.board(BoardView.TWO_CHANNELS)      # play() mutates the live board in place
    with pytest.raises(TypeError):
        g.board("two")
    # a finished game refuses another move
    while not g.has_finished():
        first = next(iter(OthelloGame.get_player_valid_actions(g.board(BoardView.TWO_CHANNELS), g.current_player)))      # a generator, as in the reference
        g.play(*tuple(int(x) for x in first))
    with pytest.raises(AssertionError, match="Game has ended"):
        g.play(0, 0)
    with pytest.raises(AssertionError, match="Board size must be even"):
        OthelloGame(5)
    real = NNetWrapper((n, n), num_channels_1=128, max_batch=1, seed=1)
    with pytest.raises(AssertionError, match=".h5"):
        real.load_checkpoint(str(tmp_path / "weights.bin"))


@pytest.mark.gpu
def test_c_abi_empty_ragged_and_invalid_inputs(oz):
    """the boundary's own edge cases: an empty batch is a no-op, a batch beyond max_batch is refused with an error code and a message (no
    exception crosses the C ABI, nothing is written), the Python mirror chunks a ragged batch over max_batch-sized calls with the same
    bits as single calls, null pointers are refused"""
    from othellozero_amd.NNet import NNetWrapper
    lib = oz.load()
    n = 6
    net = NNetWrapper((n, n), num_channels_1=128, max_batch=8, seed=4)
    rs = np.random.RandomState(9)
    valid = np.uint64(sum(1 << (r * 8 + c) for r in range(n) for c in range(n)))
    own = rs.randint(0, 2**63, size=21, dtype=np.uint64) & valid
    opp = rs.randint(0, 2**63, size=21, dtype=np.uint64) & valid & ~own
    pi0, v0 = net.predict_batch(own[:0], opp[:0])
    assert pi0.shape == (0, n, n) and v0.shape == (0,)
    pi, v = net.predict_batch(own, opp)                                   # 21 positions over calls of 8, 8, 5
    for i in (0, 7, 8, 20):
        p1, v1 = net.predict_batch(own[i:i + 1], opp[i:i + 1])
        assert np.array_equal(p1[0], pi[i]) and v1[0] == v[i]
    out_pi, out_v = np.full((9, n * n), 7.0, np.float32), np.full(9, 7.0, np.float32)
    rc = lib.oz_net_predict(net._h, oz.p_u64(own[:9].copy()), oz.p_u64(opp[:9].copy()), 9, oz.p_f32(out_pi), oz.p_f32(out_v))
    assert rc == oz.OZ_ERR_ARG and b"max_batch" in lib.oz_last_error()
    assert np.all(out_pi == 7.0) and np.all(out_v == 7.0)
    assert lib.oz_net_predict(net._h, None, None, 1, oz.p_f32(out_pi), oz.p_f32(out_v)) == oz.OZ_ERR_ARG
    assert lib.oz_net_predict(net._h, oz.p_u64(own[:1].copy()), oz.p_u64(opp[:1].copy()), 0, oz.p_f32(out_pi), oz.p_f32(out_v)) == oz.OZ_OK
    assert np.all(out_pi == 7.0)
    # the reference's own board layout at the boundary (Net/NNet.py:80-84): (n, n, 2) NHWC bytes, channel 0 = the mover
    boards = np.stack([oz.unpack_board(int(o), int(p), n) for o, p in zip(own[:8], opp[:8])]).astype(np.uint8)
    bp, bv = np.zeros((8, n * n), np.float32), np.zeros(8, np.float32)
    assert lib.oz_net_predict_boards(net._h, oz.p_u8(np.ascontiguousarray(boards)), 8, oz.p_f32(bp), oz.p_f32(bv)) == oz.OZ_OK
    assert np.array_equal(bp.reshape(8, n, n), pi[:8]) and np.array_equal(bv, v[:8])
    both = boards[:1].copy(); both[0, 2, 2, :] = 1
    assert lib.oz_net_predict_boards(net._h, oz.p_u8(both), 1, oz.p_f32(bp), oz.p_f32(bv)) == oz.OZ_ERR_ARG


@pytest.mark.gpu
def test_arena_shards_equal_the_whole_arena(oz):
    """arena games shard over ranks like self-play games.  A game depends on its GLOBAL id only (the RNG streams of the
    random agent and of the max-visit tie-break are keyed by it), so two shards played separately equal one arena of all the games, and
    distributed.arena_sharded on a one-rank world equals agents.arena_batch."""
    from othellozero_amd.agents import arena_batch
    from othellozero_amd.distributed import arena_sharded
    n, sims, G = 6, 10, 10
    from othellozero_amd.NNet import StubNetWrapper
    a = StubNetWrapper((n, n), 21, 0, max_batch=G)
    whole = arena_batch(a, None, n, G, sims, 1.0, seed=5)                        # stub network (BLACK) against the random agent (WHITE)
    lo = arena_batch(a, None, n, 6, sims, 1.0, seed=5, first_game_id=0)
    hi = arena_batch(a, None, n, 4, sims, 1.0, seed=5, first_game_id=6)
    for k in ("winner", "points", "n_moves", "final_black", "final_white"):
        assert np.array_equal(whole[k], np.concatenate([lo[k], hi[k]])), k
    assert len(set(whole["final_black"].tolist())) > 1                           # the random agent makes the games differ
    pooled = arena_sharded(a, None, n, G, sims, 1.0, seed=5)
    assert np.array_equal(pooled["game_id"], np.arange(G))
    for k in ("winner", "points", "n_moves"):
        assert np.array_equal(pooled[k], whole[k]), k


@pytest.mark.gpu
def test_library_side_random_initialisation_is_keras_default_and_deterministic(oz):
    """oz_net_init_random (the C host's way to a fresh OthelloNN): glorot_uniform limits per layer, zero biases, identity BatchNormalization
    (Net/OthelloNN.py:42-56 with Keras defaults), the same weights for the same seed, and a network the float64 oracle agrees with"""
    from othellozero_amd.NNet import NNetWrapper
    n, C_ = 6, 128
    net = NNetWrapper((n, n), num_channels_1=C_, max_batch=8, seed=1)
    net.init_random(42)
    w = net.get_weights()
    other = NNetWrapper((n, n), num_channels_1=C_, max_batch=8, seed=2)
    other.init_random(42)
    assert all(np.array_equal(a, b) for a, b in zip(w, other.get_weights()))
    other.init_random(43)
    assert not np.array_equal(w[6], other.get_weights()[6])
    fans = [(9 * 2, 9 * C_), (9 * C_, 9 * C_), (9 * C_, 9 * C_), (9 * C_, 9 * C_), ((n - 4) ** 2 * C_, 1024), (1024, 512)]
    for layer, (fi, fo) in enumerate(fans):
        k, bias, gamma, beta, mean, var = w[6 * layer: 6 * layer + 6]
        lim = np.sqrt(6.0 / (fi + fo))
        assert np.abs(k).max() <= lim * (1 + 1e-6) and np.abs(k).max() > 0.98 * lim and abs(float(k.mean())) < 0.02 * lim
        assert abs(float(k.std()) - lim / np.sqrt(3.0)) < 0.03 * lim                 # uniform on (-lim, lim)
        assert not bias.any() and np.all(gamma == 1) and not beta.any() and not mean.any() and np.all(var == 1)
    for k, bias, fo in ((w[36], w[37], n * n), (w[38], w[39], 1)):
        assert np.abs(k).max() <= np.sqrt(6.0 / (512 + fo)) * (1 + 1e-6) and not bias.any()
    rs = np.random.RandomState(3)
    valid = np.uint64(sum(1 << (r * 8 + c) for r in range(n) for c in range(n)))
    own = rs.randint(0, 2**63, size=8, dtype=np.uint64) & valid
    opp = rs.randint(0, 2**63, size=8, dtype=np.uint64) & valid & ~own
    pi, v = net.predict_batch(own, opp)
    pi64, v64 = nn_numpy.forward(w, own, opp, n)
    assert np.abs(pi.reshape(8, -1) - pi64).max() <= 1e-5 and np.abs(v - v64).max() <= 1e-5       # tolerance 1e-5 (north_star)


@pytest.mark.gpu
def test_c_abi_policy_equals_the_mirror(oz, monkeypatch):
    """oz_mcts_policy = OthelloMCTS.get_policy_action_probabilities (othelo_mcts.py:51-67) behind the C ABI: bit for bit the float64 (n, n) array
    the Python mirror computes with the reference's own expressions, for temperatures 1, 0.5, 2, 1/3, 0.7, 1.5, 3, and for temperature 0 with the tie
    draw standing in for random.choice(bests)"""
    from othellozero_amd.Othello import BoardView, OthelloGame, OthelloPlayer
    from othellozero_amd.othelo_mcts import OthelloMCTS
    lib = oz.load()
    n = 6
    m = OthelloMCTS(n, PyStubNet(n, 11, 0, True), 1.0, q_mode=1)
    g = OthelloGame(n)
    state = g.board(BoardView.TWO_CHANNELS)
    pol, rc = np.zeros((1, n * n), np.float64), np.zeros(1, np.int32)
    for sims in (1, 2, 7, 40):
        while m.N(state) < sims - 1:
            m.simulate(state, OthelloPlayer.BLACK)
        if sims == 1:
            m.simulate(state, OthelloPlayer.BLACK)                       # expanded, never selected from: KeyError in the reference, rc 2 here
            with pytest.raises(KeyError):
                m.get_policy_action_probabilities(state, 1)
            oz.check(lib.oz_mcts_policy(m._h, 1.0, None, pol.ctypes.data_as(C.POINTER(C.c_double)), oz.p_i32(rc)))
            assert rc[0] == 2 and not pol.any()
            continue
        for T in (1, 0.5, 2, 1 / 3, 0.7, 1.5, 3):               # exact powers / roots and temperatures where pow() has to round
            want = m.get_policy_action_probabilities(state, T)
            oz.check(lib.oz_mcts_policy(m._h, float(T), None, pol.ctypes.data_as(C.POINTER(C.c_double)), oz.p_i32(rc)))
            assert rc[0] == 0 and np.array_equal(pol.reshape(n, n), want), (sims, T)
        for k in (0, 1, 5):
            monkeypatch.setattr(random, "choice", lambda seq, k=k: seq[k % len(seq)])
            want = m.get_policy_action_probabilities(state, 0)
            draws = np.array([k], np.uint64)
            oz.check(lib.oz_mcts_policy(m._h, 0.0, oz.p_u64(draws), pol.ctypes.data_as(C.POINTER(C.c_double)), oz.p_i32(rc)))
            assert np.array_equal(pol.reshape(n, n), want), (sims, k)
            monkeypatch.undo()


@pytest.mark.gpu
def test_policy_temperatures_vs_reference_golden(oz, monkeypatch):
    """M10 at temperatures whose N ** (1 / T) is not an exact power: the root policy of four searches through the
    Python mirror AND through oz_mcts_policy, and every move's pi of one whole episode at T = 0.5 (drop-in execute_episode, the
    reference's draws patched as the generator patched them) -- bit for bit against tests/golden/policy_temps.npz, which
    tests/golden/gen_golden.py wrote by running the reference (othelo_mcts.py:51-67)."""
    from othellozero_amd import training
    from othellozero_amd.Othello import OthelloGame, OthelloPlayer
    from othellozero_amd.othelo_mcts import OthelloMCTS
    lib = oz.load()
    g = load_golden("policy_temps.npz")
    temps = [float(t) for t in g["temps"]]
    for name in g["names"]:
        name = str(name)
        n, salt, keep, qmode, sims = (int(x) for x in g[f"{name}/meta"])
        root = oz.unpack_board(int(g[f"{name}/root"][0]), int(g[f"{name}/root"][1]), n)
        m = OthelloMCTS(n, PyStubNet(n, salt, keep, qmode == 1), float(g[f"{name}/c"][0]), q_mode=qmode, node_cap=1024)
        for _ in range(sims):
            m.simulate(root, OthelloPlayer.BLACK)
        pol, rc = np.zeros((1, n * n), np.float64), np.zeros(1, np.int32)
        for i, T in enumerate(temps):
            assert np.array_equal(m.get_policy_action_probabilities(root, T), g[f"{name}/pi"][i]), (name, T)
            oz.check(lib.oz_mcts_policy(m._h, T, None, pol.ctypes.data_as(C.POINTER(C.c_double)), oz.p_i32(rc)))
            assert rc[0] == 0 and np.array_equal(pol.reshape(n, n), g[f"{name}/pi"][i]), (name, T)
    name = "ep6_T05"
    n, sims, seed, game, salt, keep, qmode, k = (int(x) for x in g[f"{name}/meta"])
    c, T, eg = (float(x) for x in g[f"{name}/params"])
    ply, pis, pis_abi = [0], [], []
    orig_play, orig_policy = OthelloGame.play, OthelloMCTS.get_policy_action_probabilities

    def counting_play(self, row, col):
        orig_play(self, row, col)
        ply[0] += 1

    def recording_policy(self, state, temperature):
        pi = orig_policy(self, state, temperature)
        pis.append(np.array(pi))
        pol, rc = np.zeros((1, n * n), np.float64), np.zeros(1, np.int32)
        oz.check(lib.oz_mcts_policy(self._h, float(temperature), None, pol.ctypes.data_as(C.POINTER(C.c_double)), oz.p_i32(rc)))
        pis_abi.append(pol.reshape(n, n).copy())
        return pi
    monkeypatch.setattr(OthelloGame, "play", counting_play)
    monkeypatch.setattr(OthelloMCTS, "get_policy_action_probabilities", recording_policy)
    _patch_rng(monkeypatch, seed, game, lambda: ply[0])
    ex = training.execute_episode(n, PyStubNet(n, salt, keep, qmode == 1), int(c), sims, T, eg, q_mode=qmode)
    assert len(ex) == 8 * k and len(pis) == k
    assert [z for _, _, z in ex] == [int(z) for z in g[f"{name}/ex_z"]]
    for i in range(k):
        assert np.array_equal(pis[i], g[f"{name}/pi"][i]) and np.array_equal(pis_abi[i], g[f"{name}/pi"][i]), i
