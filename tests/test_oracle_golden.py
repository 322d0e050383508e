"""Pins the CPU oracle (oracle/oz_oracle.c) against fixtures produced by the
reference's own Python (tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

import oracle
from conftest import load_golden

QT_INT, QT_F32, QT_F64 = 0, 1, 2


def test_initial_boards(golden_rules):
    L = oracle.lib()
    import ctypes as C
    for n in (4, 6, 8):
        b, w = C.c_uint64(), C.c_uint64()
        L.orc_initial_board(n, C.byref(b), C.byref(w))
        assert (b.value, w.value) == tuple(int(x) for x in golden_rules[f"initial_{n}"])
    # [verified] 8x8 opening moves
    b8, w8 = (int(x) for x in golden_rules["initial_8"])
    assert oracle.mask_to_squares(L.orc_legal_mask(b8, w8, 8, 0)) == [2 * 8 + 3, 3 * 8 + 2, 4 * 8 + 5, 5 * 8 + 4]


def test_rules_positions(golden_rules):
    g = golden_rules
    L = oracle.lib()
    import ctypes as C
    for i in range(len(g["n"])):
        n, b, w = int(g["n"][i]), int(g["black"][i]), int(g["white"][i])
        assert L.orc_legal_mask(b, w, n, 0) == int(g["legal_black"][i]), i
        assert L.orc_legal_mask(b, w, n, 1) == int(g["legal_white"][i]), i
        assert L.orc_finished(b, w, n) == int(g["finished"][i]), i
        pts = C.c_int()
        assert L.orc_winner(b, w, n, C.byref(pts)) == int(g["winner"][i]), i
        assert pts.value == int(g["winner_pts"][i]), i


def test_rules_moves(golden_rules):
    g = golden_rules
    L = oracle.lib()
    import ctypes as C
    for j in range(len(g["mv_pos"])):
        i = int(g["mv_pos"][j])
        n = int(g["n"][i])
        b, w = C.c_uint64(int(g["black"][i])), C.c_uint64(int(g["white"][i]))
        pch = 0 if int(g["mv_player"][j]) == 1 else 1
        L.orc_apply_move(C.byref(b), C.byref(w), n, pch, int(g["mv_sq"][j]))
        assert (b.value, w.value) == (int(g["mv_black"][j]), int(g["mv_white"][j])), j


def test_flip_through_quirk():
    """R3: . O O X O X  with X to move at col 0 flips 3 discs (incl. the O between the two X)."""
    L = oracle.lib()
    b = (1 << 3) | (1 << 5)
    w = (1 << 1) | (1 << 2) | (1 << 4)
    assert L.orc_flip_mask(b, w, 8, 0, 0) == w


def test_game_play_transitions(golden_rules):
    g = golden_rules
    L = oracle.lib()
    import ctypes as C
    for j in range(len(g["pl_n"])):
        b, w = C.c_uint64(int(g["pl_black"][j])), C.c_uint64(int(g["pl_white"][j]))
        pl, fin = C.c_int(int(g["pl_player"][j])), C.c_int(0)
        L.orc_game_play(C.byref(b), C.byref(w), int(g["pl_n"][j]), C.byref(pl), C.byref(fin), int(g["pl_sq"][j]))
        assert (b.value, w.value, pl.value, fin.value) == (
            int(g["pl_black2"][j]), int(g["pl_white2"][j]), int(g["pl_player2"][j]), int(g["pl_finished2"][j])), j


def test_demo_one_channel(golden_rules):
    g = golden_rules
    i = int(g["demo6_index"][0])
    brd = oracle.unpack_board(int(g["black"][i]), int(g["white"][i]), 6)
    one = brd[:, :, 0].astype(np.int8) - brd[:, :, 1].astype(np.int8)
    assert np.array_equal(one, g["demo6_one_channel"])


def test_pairwise_sum_matches_numpy():
    g = load_golden("pairwise.npz")
    for x, y, n in zip(g["x"], g["y"], g["length"]):
        assert oracle.pairwise_sum(x[:n]) == y
    # and against the NumPy installed here, on fresh inputs
    rs = np.random.RandomState(3)
    for n in (16, 36, 64):
        for _ in range(200):
            x = rs.random_sample(n) * (rs.random_sample(n) < 0.4)
            assert oracle.pairwise_sum(x) == np.sum(x.reshape(int(n ** 0.5), -1))


def test_symmetry_tables():
    g = load_golden("symmetries.npz")
    for n in (4, 6, 8):
        assert np.array_equal(oracle.symmetry_perms(n), g[f"perm_{n}"])
    # SURVEY T3 known answer, first output of the 4x4 grid
    assert list(oracle.symmetry_perms(4)[0]) == [15, 11, 7, 3, 14, 10, 6, 2, 13, 9, 5, 1, 12, 8, 4, 0]


def _check_tables(m, g, prefix, qmode):
    dump = m.dump()
    boards = g[prefix + "boards"]
    assert len(dump) == len(boards)
    for i, nd in enumerate(dump):
        assert (nd["k0"], nd["k1"]) == (int(boards[i][0]), int(boards[i][1])), (prefix, i)
        assert nd["Ns"] == int(g[prefix + "Ns"][i])
        assert nd["edges_init"] == int(g[prefix + "edges_init"][i])
        assert nd["legal"] == int(g[prefix + "legal"][i])
        assert np.array_equal(nd["P"], g[prefix + "P"][i]), (prefix, i)          # bit-exact float64
        if nd["edges_init"]:
            assert np.array_equal(nd["N"], g[prefix + "N"][i])
            assert np.array_equal(nd["Q"], g[prefix + "Q"][i]), (prefix, i)      # bit-exact
            gq = g[prefix + "qtype"][i]
            for sq in oracle.mask_to_squares(nd["legal"]):
                want_f32 = gq[sq] == QT_F32
                assert (nd["qtag"][sq] == 1) == want_f32, (prefix, i, sq)


def test_mcts_traces(golden_mcts):
    g = golden_mcts
    for name in g["names"]:
        name = str(name)
        n, player, salt, keep, qmode, nsims = (int(x) for x in g[f"{name}/meta"])
        c = float(g[f"{name}/c"][0])
        rb, rw = (int(x) for x in g[f"{name}/root"])
        m = oracle.Mcts(n, c, qmode, salt=salt, keep_mask=keep)
        done = 0
        rets, rts = [], []
        for cp in g[f"{name}/cps"]:
            while done < int(cp):
                v, vt = m.simulate(rb, rw, player)
                rets.append(v); rts.append(vt)
                done += 1
            _check_tables(m, g, f"{name}/cp{int(cp)}/", qmode)
        assert np.array_equal(np.array(rets), g[f"{name}/ret"]), name
        assert np.array_equal(np.array(rts, dtype=np.uint8), g[f"{name}/ret_type"]), name
        k0, k1 = (rb, rw) if player == 1 else (rw, rb)
        pol, _ = m.policy(k0, k1, 1.0)
        assert np.array_equal(pol, g[f"{name}/pi_T1"]), name


def test_mcts_fallback_branch_is_exercised(golden_mcts):
    """the sparse stub nets must reach "All valid moves were masked" (MCTS/__init__.py:52-55)"""
    g = golden_mcts
    hit = 0
    for name in ("init6_sparse", "init4_sparse"):
        n, player, salt, keep, qmode, nsims = (int(x) for x in g[f"{name}/meta"])
        rb, rw = (int(x) for x in g[f"{name}/root"])
        m = oracle.Mcts(n, float(g[f"{name}/c"][0]), qmode, salt=salt, keep_mask=keep)
        for _ in range(nsims):
            m.simulate(rb, rw, player)
        hit += m.stats()["fallback"]
    assert hit > 0


def test_episodes(golden_episodes):
    g = golden_episodes
    for name in g["names"]:
        name = str(name)
        n, sims, seed, game, salt, keep, qmode, k = (int(x) for x in g[f"{name}/meta"])
        c, T, eg = (float(x) for x in g[f"{name}/params"])
        m = oracle.Mcts(n, c, qmode, salt=salt, keep_mask=keep)
        ep = m.episode(sims, T, eg, seed, game)
        assert ep["n_moves"] == k, name
        assert np.array_equal(ep["action"], g[f"{name}/action"]), name
        assert np.array_equal(ep["player"], g[f"{name}/player"]), name
        assert np.array_equal(ep["black"], g[f"{name}/black"]), name
        assert np.array_equal(ep["white"], g[f"{name}/white"]), name
        assert np.array_equal(ep["counts"], g[f"{name}/counts"]), name
        assert ep["stats"]["expansions"] == int(g[f"{name}/n_expansions"][0]), name
        # as-returned examples: 8 symmetries per move, board aliased to the FINAL position (T2), one-hot policy, z
        perms = oracle.symmetry_perms(n)
        fin = oracle.unpack_board(ep["final_black"], ep["final_white"], n).reshape(n * n, 2)
        eb, epol, ez = g[f"{name}/ex_board"], g[f"{name}/ex_policy"], g[f"{name}/ex_z"]
        for i in range(k):
            a = int(ep["action"][i]); a = (a >> 3) * n + (a & 7)
            for t in range(8):
                sym = fin[perms[t]].reshape(n, n, 2)
                assert oracle.pack_board(sym) == (int(eb[8 * i + t][0]), int(eb[8 * i + t][1])), (name, i, t)
                assert int(np.nonzero(perms[t] == a)[0][0]) == int(epol[8 * i + t]), (name, i, t)
                assert int(ep["z"][i]) == int(ez[8 * i + t])


def test_arena(golden_arena):
    g = golden_arena
    for name in g["names"]:
        name = str(name)
        n, sims, seed, game, sa, sb, qmode, k = (int(x) for x in g[f"{name}/meta"])
        c = float(g[f"{name}/c"][0])
        ma = oracle.Mcts(n, c, qmode, salt=sa)
        mb = oracle.Mcts(n, c, qmode, salt=sb)
        r = oracle.arena(ma, mb, sims, seed, game)
        assert r["n_moves"] == k, name
        assert np.array_equal(r["action"], g[f"{name}/action"]), name
        assert np.array_equal(r["player"], g[f"{name}/player"]), name
        assert (r["final_black"], r["final_white"]) == tuple(int(x) for x in g[f"{name}/final"]), name
        assert (r["winner"], r["points"]) == tuple(int(x) for x in g[f"{name}/result"]), name


def test_arena_against_random_agent():
    """NeuralNetworkOthelloAgent vs RandomOthelloAgent (the evaluation games of main.py:163-233), both colours:
    the oracle's random mover reproduces the reference's traces (fixture arena_random.npz)"""
    g = load_golden("arena_random.npz")
    for name in g["names"]:
        name = str(name)
        n, sims, seed, game, salt, colour, qmode, k = (int(x) for x in g[f"{name}/meta"])
        m = oracle.Mcts(n, float(g[f"{name}/c"][0]), qmode, salt=salt)
        r = oracle.arena(m if colour == 1 else None, None if colour == 1 else m, sims, seed, game)
        assert r["n_moves"] == k, name
        assert np.array_equal(r["action"], g[f"{name}/action"]) and np.array_equal(r["player"], g[f"{name}/player"]), name
        assert (r["final_black"], r["final_white"]) == tuple(int(x) for x in g[f"{name}/final"]), name
        assert (r["winner"], r["points"]) == tuple(int(x) for x in g[f"{name}/result"][:2]), name


def test_stub_net_python_callback_equals_builtin():
    """the evaluator-callback plumbing gives the same search as the builtin stub"""
    def ev(own, opp, n):
        pi, v = oracle.stub_predict(own, opp, n, salt=5)
        return pi, v
    a = oracle.Mcts(6, 1.0, oracle.QMODE_NEP50, salt=5)
    b = oracle.Mcts(6, 1.0, oracle.QMODE_NEP50, evaluator=ev)
    import ctypes as C
    L = oracle.lib()
    bl, wh = C.c_uint64(), C.c_uint64()
    L.orc_initial_board(6, C.byref(bl), C.byref(wh))
    for _ in range(60):
        assert a.simulate(bl.value, wh.value, 1) == b.simulate(bl.value, wh.value, 1)
    da, db = a.dump(), b.dump()
    assert len(da) == len(db)
    for x, y in zip(da, db):
        assert x["k0"] == y["k0"] and np.array_equal(x["Q"], y["Q"]) and np.array_equal(x["P"], y["P"])


def test_episodes_one_channel_view():
    """BaseNN path (training.py:34-37,61-65): same search, examples are per-move one-channel snapshots (no aliasing)"""
    g = load_golden("episodes_bnn.npz")
    for name in g["names"]:
        name = str(name)
        n, sims, seed, game, salt, keep, qmode, k = (int(x) for x in g[f"{name}/meta"])
        c, T, eg = (float(x) for x in g[f"{name}/params"])
        ep = oracle.Mcts(n, c, qmode, salt=salt, keep_mask=keep).episode(sims, T, eg, seed, game)
        assert ep["n_moves"] == k and np.array_equal(ep["action"], g[f"{name}/action"]), name
        perms = oracle.symmetry_perms(n)
        for i in range(k):
            snap = oracle.unpack_board(ep["black"][i], ep["white"][i], n).reshape(n * n, 2)
            a = int(ep["action"][i]); a = (a >> 3) * n + (a & 7)
            for t in range(8):
                assert oracle.pack_board(snap[perms[t]].reshape(n, n, 2)) == tuple(int(x) for x in g[f"{name}/ex_board"][8 * i + t])
                assert int(np.nonzero(perms[t] == a)[0][0]) == int(g[f"{name}/ex_policy"][8 * i + t])
                assert int(ep["z"][i]) == int(g[f"{name}/ex_z"][8 * i + t])


def test_policy_at_awkward_temperatures():
    """M10 where `N ** (1 / T)` is not an exact power or root (T = 1/3, 0.7, 1.5, 3 next to 0.5 and 2): the reference evaluates
    Python int ** float = libm pow on float64, then np.sum (othelo_mcts.py:64-67).  Root policies of four searches and the pi of every
    move of one whole episode at T = 0.5, bit for bit against tests/golden/policy_temps.npz (generated by running the reference)."""
    g = load_golden("policy_temps.npz")
    temps = [float(t) for t in g["temps"]]
    assert any(abs(t - 1 / 3) < 1e-15 for t in temps) and 0.7 in temps and 1.5 in temps
    for name in g["names"]:
        name = str(name)
        n, salt, keep, qmode, sims = (int(x) for x in g[f"{name}/meta"])
        rb, rw = (int(x) for x in g[f"{name}/root"])
        m = oracle.Mcts(n, float(g[f"{name}/c"][0]), qmode, salt=salt, keep_mask=keep)
        for _ in range(sims):
            m.simulate(rb, rw, 1)
        rc, cnt, legal = m.counts(rb, rw)
        assert rc == 0 and np.array_equal(cnt, g[f"{name}/counts"]), name
        for i, T in enumerate(temps):
            pol, _ = m.policy(rb, rw, T)
            assert np.array_equal(pol, g[f"{name}/pi"][i]), (name, T)
            assert abs(pol.sum() - 1.0) < 1e-12
    name = "ep6_T05"
    n, sims, seed, game, salt, keep, qmode, k = (int(x) for x in g[f"{name}/meta"])
    c, T, eg = (float(x) for x in g[f"{name}/params"])
    assert T == 0.5
    ep = oracle.Mcts(n, c, qmode, salt=salt, keep_mask=keep).episode(sims, T, eg, seed, game)
    assert ep["n_moves"] == k and np.array_equal(ep["action"], g[f"{name}/action"]) and np.array_equal(ep["counts"], g[f"{name}/counts"])
    assert np.array_equal(np.repeat(ep["z"], 8), g[f"{name}/ex_z"])
    L = oracle.lib()
    for i in range(k):
        b, w, pl = int(g[f"{name}/black"][i]), int(g[f"{name}/white"][i]), int(g[f"{name}/player"][i])
        own, opp = (b, w) if pl == 1 else (w, b)
        pi = oracle.policy_from_counts(n, ep["counts"][i], L.orc_legal_mask(own, opp, n, 0), T)
        assert np.array_equal(pi, g[f"{name}/pi"][i]), i
