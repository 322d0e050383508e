#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Run in the build container only (the reference lives at /root/reference and never
travels):   python tests/golden/gen_golden.py

What runs here is the reference's own Python: Othello/__init__.py,
MCTS/__init__.py, othelo_mcts.py, training.py, agents.py.  `Net.NNet` (TensorFlow /
Keras, not installed) is replaced by a stub module that only supplies the
`NeuralNets` enum; the network object handed to the reference is a deterministic
integer-hash "stub net" whose formula is restated in oracle/oz_oracle.c
(orc_stub_predict).  The three random draws on the path (random.random,
np.random.choice, random.choice - training.py:51,56, othelo_mcts.py:59) are
patched to read the counter-based streams of orc_rng keyed (seed, game, ply).

Only DATA is written (inputs and the reference's outputs); no reference source.

Fixtures:
  rules.npz      positions -> legal masks, per-move flipped boards, finished, points, winner, play() transitions
  pairwise.npz   float64 vectors (16/36/64) -> np.sum
  symmetries.npz training_example_symmetries gather tables for n = 4, 6, 8
  mcts.npz       search traces: expanded boards in order + full (Ns, Nsa, Qsa, Psa) tables, simulate() returns
  episodes.npz   execute_episode traces (moves, snapshots, root counts, returned examples, z)
  arena.npz      duel_between_agents traces
  episodes_bnn.npz  execute_episode through the one-channel (BaseNN) board view
  policy_temps.npz  get_policy_action_probabilities at temperatures whose N ** (1 / T) is not an exact power (1/3, 0.7, 1.5, ...) and one
                    episode at T = 0.5, every move's pi recorded
"""
import enum
import os
import random
import runpy
import sys
import types
import io
import contextlib

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
M64 = (1 << 64) - 1


# ---------------------------------------------------------------- reference import
class NeuralNets(enum.Enum):
    ONN = enum.auto()
    BNN = enum.auto()


def import_reference():
    sys.path.insert(0, REF)
    net_pkg = types.ModuleType("Net")
    nnet = types.ModuleType("Net.NNet")
    nnet.NeuralNets = NeuralNets
    net_pkg.NNet = nnet
    sys.modules["Net"] = net_pkg
    sys.modules["Net.NNet"] = nnet
    import Othello, MCTS, othelo_mcts, training, agents  # noqa: E401
    return Othello, MCTS, othelo_mcts, training, agents


Othello, MCTS, othelo_mcts, training, agents = import_reference()
OthelloGame, OthelloPlayer, BoardView = Othello.OthelloGame, Othello.OthelloPlayer, Othello.BoardView


# ---------------------------------------------------------------- shared integer mixers (= oz_oracle.c)
def sm64(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def rotl64(x, r):
    return ((x << r) | (x >> (64 - r))) & M64


def rng(seed, game, move, stream):
    a = sm64((seed + 0x632BE59BD9B4E019 * game) & M64)
    return sm64(a ^ ((move * 0x9E3779B97F4A7C15) & M64) ^ ((stream * 0xD1B54A32D192ED03) & M64))


RNG_COIN, RNG_EXPLORE, RNG_TIE = 0, 1, 2


def rng_unit(u):
    return (u >> 11) * (1.0 / 9007199254740992.0)


def stub_h(own, opp, salt, i):
    return sm64(sm64(own ^ salt) ^ rotl64(opp, 29) ^ (((i + 1) * 0xD6E8FEB86659FD93) & M64))


def pack(board):
    b = np.asarray(board)
    n = b.shape[0]
    c0 = c1 = 0
    for r in range(n):
        for c in range(n):
            if b[r, c, 0]:
                c0 |= 1 << (r * 8 + c)
            if b[r, c, 1]:
                c1 |= 1 << (r * 8 + c)
    return c0, c1


def unpack(c0, c1, n):
    b = np.zeros((n, n, 2), dtype=bool)
    for r in range(n):
        for c in range(n):
            b[r, c, 0] = (c0 >> (r * 8 + c)) & 1
            b[r, c, 1] = (c1 >> (r * 8 + c)) & 1
    return b


class StubNet:
    """Duck-typed stand-in for NNetWrapper (Net/NNet.py:22-101): .network_type + .predict."""
    network_type = NeuralNets.ONN

    def __init__(self, n, salt, keep_mask, regime):
        self.n, self.salt, self.keep, self.regime = n, salt, keep_mask, regime
        self.calls = []          # boards in predict order == expansion order

    def predict(self, board):
        n = self.n
        own, opp = pack(board)
        self.calls.append((own, opp))
        w = np.zeros(n * n, dtype=np.uint32)
        for r in range(n):
            for c in range(n):
                u = stub_h(own, opp, self.salt, r * 8 + c)
                wi = (u >> 40) & 0xFFFF
                if ((u >> 8) & self.keep) != 0:
                    wi = 0
                w[r * n + c] = wi
        s = int(w.sum())
        if s == 0:
            w[0] = 1
            s = 1
        pi = (w.astype(np.float32) / np.float32(s)).reshape(n, n)
        m = stub_h(own, opp, self.salt, 64) >> 40
        v = np.float32(m - 8388608) / np.float32(8388608.0)
        assert isinstance(v, np.float32)
        if self.regime == "f64":       # what NumPy 1.18 promotion computes == all-float64 arithmetic
            v = float(v)
        return pi, v


def mask_of(actions):
    m = 0
    for a in actions:
        m |= 1 << (int(a[0]) * 8 + int(a[1]))
    return m


# ---------------------------------------------------------------- rules fixtures
def rules_fixture():
    rnd = random.Random(20240601)
    pos = []      # (n, black, white)
    plays = []    # (n, black, white, player, sq, black', white', player', finished')
    for n, games in ((4, 30), (6, 25), (8, 25)):
        for _ in range(games):
            g = OthelloGame(n)
            while not g.has_finished():
                b0, w0 = pack(g.board(BoardView.TWO_CHANNELS))
                pos.append((n, b0, w0))
                acts = [tuple(int(x) for x in a) for a in g.get_valid_actions()]
                a = rnd.choice(acts)
                pl = g.current_player.value
                g.play(*a)
                b1, w1 = pack(g.board(BoardView.TWO_CHANNELS))
                plays.append((n, b0, w0, pl, a[0] * 8 + a[1], b1, w1, g.current_player.value, int(g.has_finished())))
            pos.append((n,) + pack(g.board(BoardView.TWO_CHANNELS)))
    # random (mostly unreachable) fillings: dense runs exercise the flip-through walk
    for n, count in ((4, 150), (6, 400), (8, 700)):
        for _ in range(count):
            p_empty = rnd.choice([0.15, 0.3, 0.5])
            b = w = 0
            for r in range(n):
                for c in range(n):
                    u = rnd.random()
                    if u < p_empty:
                        continue
                    if rnd.random() < 0.5:
                        b |= 1 << (r * 8 + c)
                    else:
                        w |= 1 << (r * 8 + c)
            pos.append((n, b, w))
    # hand-made: flip-through row  . O O X O X .  (BLACK=X to move at col 0), draw, empty-ish
    def row(n, r, s):
        b = w = 0
        for c, ch in enumerate(s):
            if ch == "X":
                b |= 1 << (r * 8 + c)
            elif ch == "O":
                w |= 1 << (r * 8 + c)
        return b, w
    for s in (".OOXOX..", ".OXOXOX.", ".OOOOOOX", "X.OXOOX.", ".OX.OX.."):
        pos.append((8,) + row(8, 2, s))
    pos.append((6,) + row(6, 1, ".OXOX."))
    full_draw_b = sum(1 << (r * 8 + c) for r in range(4) for c in range(4) if (r + c) % 2 == 0)
    full_draw_w = sum(1 << (r * 8 + c) for r in range(4) for c in range(4) if (r + c) % 2 == 1)
    pos.append((4, full_draw_b, full_draw_w))
    # the only position literal in the reference (Othello/__init__.py:277-322 demo), obtained by running it
    with contextlib.redirect_stdout(io.StringIO()):
        demo = runpy.run_path(os.path.join(REF, "Othello", "__init__.py"), run_name="__main__")
    pos.append((6,) + pack(demo["game"]._board))
    demo_one_channel = np.asarray(demo["board"]).astype(np.int8)      # convert_to_one_channel_board (R9)

    P = len(pos)
    out = dict(
        n=np.zeros(P, np.int32), black=np.zeros(P, np.uint64), white=np.zeros(P, np.uint64),
        legal_black=np.zeros(P, np.uint64), legal_white=np.zeros(P, np.uint64),
        finished=np.zeros(P, np.uint8), pts_black=np.zeros(P, np.int32), pts_white=np.zeros(P, np.int32),
        winner=np.zeros(P, np.int8), winner_pts=np.zeros(P, np.int32),
    )
    mv = []   # (pos index, player, sq, black', white')
    for i, (n, b, w) in enumerate(pos):
        board = unpack(b, w, n)
        out["n"][i] = n; out["black"][i] = b; out["white"][i] = w
        lb = [tuple(a) for a in OthelloGame.get_player_valid_actions(board, OthelloPlayer.BLACK)]
        lw = [tuple(a) for a in OthelloGame.get_player_valid_actions(board, OthelloPlayer.WHITE)]
        assert lb == sorted(lb) and lw == sorted(lw)      # ascending row-major order (R4)
        out["legal_black"][i] = mask_of(lb); out["legal_white"][i] = mask_of(lw)
        out["finished"][i] = int(OthelloGame.has_board_finished(board))
        pts = OthelloGame.get_board_players_points(board)
        out["pts_black"][i] = pts[OthelloPlayer.BLACK]; out["pts_white"][i] = pts[OthelloPlayer.WHITE]
        win, wp = OthelloGame.get_board_winning_player(board)
        out["winner"][i] = win.value; out["winner_pts"][i] = wp
        for player, acts in ((OthelloPlayer.BLACK, lb), (OthelloPlayer.WHITE, lw)):
            for a in acts:
                nb = np.copy(board)
                OthelloGame.flip_board_squares(nb, player, int(a[0]), int(a[1]))
                mv.append((i, player.value, int(a[0]) * 8 + int(a[1])) + pack(nb))
    mv = np.array(mv, dtype=object)
    out["mv_pos"] = mv[:, 0].astype(np.int32); out["mv_player"] = mv[:, 1].astype(np.int8)
    out["mv_sq"] = mv[:, 2].astype(np.uint8)
    out["mv_black"] = np.array([int(x) for x in mv[:, 3]], dtype=np.uint64)
    out["mv_white"] = np.array([int(x) for x in mv[:, 4]], dtype=np.uint64)
    pl = np.array(plays, dtype=object)
    for j, name in enumerate(("pl_n", "pl_black", "pl_white", "pl_player", "pl_sq", "pl_black2", "pl_white2", "pl_player2", "pl_finished2")):
        dt = np.uint64 if "black" in name or "white" in name else np.int32
        out[name] = np.array([int(x) for x in pl[:, j]], dtype=dt)
    # initial boards (R2)
    for n in (4, 6, 8):
        b, w = pack(OthelloGame.initial_board(n))
        out[f"initial_{n}"] = np.array([b, w], dtype=np.uint64)
    out["demo6_one_channel"] = demo_one_channel
    out["demo6_index"] = np.array([P - 1], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "rules.npz"), **out)
    print("rules:", P, "positions,", len(mv), "moves,", len(plays), "plays")


# ---------------------------------------------------------------- np.sum pairwise
def pairwise_fixture():
    rs = np.random.RandomState(7)
    xs, ys, ls = [], [], []
    for length in (16, 36, 64):
        for _ in range(120):
            x = rs.random_sample(length)
            if rs.random_sample() < 0.7:
                x *= rs.random_sample(length) < rs.choice([0.1, 0.25, 0.5])
            x = np.ascontiguousarray(x.reshape(int(length ** 0.5), -1))
            pad = np.zeros(64); pad[:length] = x.ravel()
            xs.append(pad); ys.append(np.sum(x)); ls.append(length)
    np.savez_compressed(os.path.join(OUT, "pairwise.npz"), x=np.array(xs), y=np.array(ys), length=np.array(ls, np.int32))
    print("pairwise:", len(xs))


# ---------------------------------------------------------------- symmetries
def symmetries_fixture():
    out = {}
    for n in (4, 6, 8):
        grid = np.arange(n * n).reshape(n, n)
        two = np.stack([grid, grid + 1000], axis=2)
        syms = training.training_example_symmetries(two, grid)
        assert len(syms) == 8
        for b, p in syms:
            assert np.array_equal(b[:, :, 0], p) and np.array_equal(b[:, :, 1], p + 1000)
        out[f"perm_{n}"] = np.array([np.asarray(p).ravel() for _, p in syms], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "symmetries.npz"), **out)
    print("symmetries ok")


# ---------------------------------------------------------------- search traces
QT_INT, QT_F32, QT_F64 = 0, 1, 2


def qtype(x):
    if isinstance(x, (np.float32,)):
        return QT_F32
    if isinstance(x, (float, np.float64)):
        return QT_F64
    if isinstance(x, (int, np.integer)):
        return QT_INT
    raise TypeError(type(x))


def dump_tables(mcts, net, n):
    K = len(net.calls)
    boards = np.zeros((K, 2), np.uint64); Ns = np.zeros(K, np.int32); ei = np.zeros(K, np.uint8)
    N = np.zeros((K, 64), np.int32); Q = np.zeros((K, 64), np.float64); qt = np.zeros((K, 64), np.uint8)
    P = np.zeros((K, 64), np.float64); legal = np.zeros(K, np.uint64)
    for i, (own, opp) in enumerate(net.calls):
        h = MCTS.hash_ndarray(unpack(own, opp, n))
        boards[i] = (own, opp); Ns[i] = mcts._Ns[h]
        nsa, qsa, psa = mcts._Nsa[h], mcts._Qsa[h], mcts._Psa[h]
        ei[i] = 1 if nsa else 0
        acts = mcts._state_actions[h]
        legal[i] = mask_of(acts)
        for r in range(n):
            for c in range(n):
                P[i, r * 8 + c] = psa[r, c]
        for a, cnt in nsa.items():
            sq = a[0] * 8 + a[1]
            N[i, sq] = cnt; Q[i, sq] = float(qsa[a]); qt[i, sq] = qtype(qsa[a])
    assert len(mcts._Ns) == K
    return dict(boards=boards, Ns=Ns, edges_init=ei, N=N, Q=Q, qtype=qt, P=P, legal=legal)


def mcts_fixture():
    cases = []
    roots8 = [OthelloGame.initial_board(8)]
    # a mid-game and a late-game root reached by a fixed playout
    rnd = random.Random(5)
    g = OthelloGame(8)
    mid = late = None
    while not g.has_finished():
        if g.round == 21:
            mid = (np.copy(g.board(BoardView.TWO_CHANNELS)), g.current_player)
        if g.round == 52:
            late = (np.copy(g.board(BoardView.TWO_CHANNELS)), g.current_player)
        g.play(*rnd.choice([tuple(a) for a in g.get_valid_actions()]))
    specs = [
        # name, n, root(board, player), salt, keep, regime, c, checkpoints
        ("init8_nep50", 8, (roots8[0], OthelloPlayer.BLACK), 11, 0, "nep50", 1, (1, 2, 3, 10, 25, 100)),
        ("init8_f64", 8, (roots8[0], OthelloPlayer.BLACK), 11, 0, "f64", 1, (1, 2, 3, 10, 25, 100)),
        ("mid8_nep50", 8, mid, 12, 0, "nep50", 1, (25, 200)),
        ("late8_nep50", 8, late, 13, 0, "nep50", 1, (50, 400)),
        ("late8_f64", 8, late, 13, 0, "f64", 1.5, (50, 400)),
        ("init6_nep50", 6, (OthelloGame.initial_board(6), OthelloPlayer.BLACK), 21, 0, "nep50", 1, (1, 2, 25, 300)),
        ("init6_sparse", 6, (OthelloGame.initial_board(6), OthelloPlayer.BLACK), 22, 7, "nep50", 1, (100,)),
        ("init4_f64", 4, (OthelloGame.initial_board(4), OthelloPlayer.BLACK), 31, 0, "f64", 2, (10, 300)),
        ("init4_sparse", 4, (OthelloGame.initial_board(4), OthelloPlayer.BLACK), 32, 3, "nep50", 1, (300,)),
    ]
    out = {}
    names = []
    for name, n, (root, player), salt, keep, regime, c, cps in specs:
        net = StubNet(n, salt, keep, regime)
        m = othelo_mcts.OthelloMCTS(n, net, c)
        rets, rts = [], []
        done = 0
        for cp in cps:
            while done < cp:
                r = m.simulate(root, player)
                rets.append(float(r)); rts.append(qtype(r))
                done += 1
            t = dump_tables(m, net, n)
            for k, v in t.items():
                out[f"{name}/cp{cp}/{k}"] = v
        rb, rw = pack(root)
        out[f"{name}/meta"] = np.array([n, player.value, salt, keep, 0 if regime == "nep50" else 1, len(rets)], dtype=np.int64)
        out[f"{name}/c"] = np.array([float(c)])
        out[f"{name}/root"] = np.array([rb, rw], dtype=np.uint64)
        out[f"{name}/cps"] = np.array(cps, dtype=np.int32)
        out[f"{name}/ret"] = np.array(rets); out[f"{name}/ret_type"] = np.array(rts, dtype=np.uint8)
        # root policy at T=1 and T=0 (ties -> stream draw 0 => first)
        state = root if player is OthelloPlayer.BLACK else OthelloGame.invert_board(root)
        out[f"{name}/pi_T1"] = m.get_policy_action_probabilities(state, 1)
        names.append(name)
        print("mcts", name, "nodes", len(net.calls), "float-typed Q edges:",
              sum(int((out[f'{name}/cp{cps[-1]}/qtype'] == QT_F64).sum()) for _ in [0]))
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "mcts.npz"), **out)


# ---------------------------------------------------------------- episode / arena traces
class Ctx:
    seed = 0
    game = 0
    ply = 0
    counts = None


def patched_rng():
    saved = (random.random, random.choice, np.random.choice)
    random.random = lambda: rng_unit(rng(Ctx.seed, Ctx.game, Ctx.ply, RNG_COIN))
    random.choice = lambda seq: seq[rng(Ctx.seed, Ctx.game, Ctx.ply, RNG_TIE) % len(seq)]
    np.random.choice = lambda k: rng(Ctx.seed, Ctx.game, Ctx.ply, RNG_EXPLORE) % k
    return saved


def restore_rng(saved):
    random.random, random.choice, np.random.choice = saved


class CountingGame(OthelloGame):
    log = None

    def play(self, row, col):
        b, w = pack(self._board)
        CountingGame.log.append((b, w, self.current_player.value, int(row) * 8 + int(col)))
        super().play(row, col)
        Ctx.ply += 1


class RecordingMCTS(othelo_mcts.OthelloMCTS):
    def get_policy_action_probabilities(self, state, temperature):
        cnt = np.zeros(64, np.int32)
        for a in self._get_state_actions(state):
            cnt[a[0] * 8 + a[1]] = self.N(state, a)
        Ctx.counts.append(cnt)
        return super().get_policy_action_probabilities(state, temperature)


def episodes_fixture():
    specs = [
        # name, n, sims, c, T, e_greedy, seed, game, salt, keep, regime
        ("ep8_25", 8, 25, 1, 1, 0.9, 1234, 0, 101, 0, "nep50"),
        ("ep8_25_f64", 8, 25, 1, 1, 0.9, 1234, 1, 101, 0, "f64"),
        ("ep8_100", 8, 100, 1, 1, 0.9, 1234, 2, 102, 0, "nep50"),
        ("ep8_T0", 8, 30, 1, 0, 1.0, 99, 3, 103, 0, "nep50"),
        ("ep6_100", 6, 100, 1, 1, 0.9, 1234, 4, 104, 0, "nep50"),
        ("ep6_50_f64_c2", 6, 50, 2.5, 1, 0.5, 77, 5, 105, 0, "f64"),
        ("ep6_sparse", 6, 40, 1, 1, 0.9, 5, 6, 106, 7, "nep50"),
        ("ep4_60", 4, 60, 1, 0, 0.8, 3, 7, 107, 0, "nep50"),
    ]
    out = {}
    names = []
    saved = patched_rng()
    training.OthelloGame = CountingGame
    training.OthelloMCTS = RecordingMCTS
    try:
        for name, n, sims, c, T, eg, seed, game, salt, keep, regime in specs:
            Ctx.seed, Ctx.game, Ctx.ply, Ctx.counts = seed, game, 0, []
            CountingGame.log = []
            net = StubNet(n, salt, keep, regime)
            ex = training.execute_episode(n, net, c, sims, T, eg)
            log = CountingGame.log
            k = len(log)
            assert len(ex) == 8 * k
            out[f"{name}/meta"] = np.array([n, sims, seed, game, salt, keep, 0 if regime == "nep50" else 1, k], dtype=np.int64)
            out[f"{name}/params"] = np.array([float(c), float(T), float(eg)])
            out[f"{name}/black"] = np.array([x[0] for x in log], dtype=np.uint64)
            out[f"{name}/white"] = np.array([x[1] for x in log], dtype=np.uint64)
            out[f"{name}/player"] = np.array([x[2] for x in log], dtype=np.int8)
            out[f"{name}/action"] = np.array([x[3] for x in log], dtype=np.uint8)
            out[f"{name}/counts"] = np.array(Ctx.counts, dtype=np.int32)
            # as-returned examples (aliasing quirk T2: boards are views of the final position)
            eb = np.zeros((8 * k, 2), np.uint64); ep = np.zeros(8 * k, np.int32); ez = np.zeros(8 * k, np.int8)
            for i, (b, p, z) in enumerate(ex):
                assert b.dtype == np.bool_ and p.dtype == np.float64 and isinstance(z, int)
                eb[i] = pack(b)
                nz = np.argwhere(p == 1.0)
                assert nz.shape[0] == 1 and p.sum() == 1.0
                ep[i] = nz[0][0] * n + nz[0][1]
                ez[i] = z
            out[f"{name}/ex_board"] = eb; out[f"{name}/ex_policy"] = ep; out[f"{name}/ex_z"] = ez
            out[f"{name}/n_expansions"] = np.array([len(net.calls)], dtype=np.int64)
            names.append(name)
            print("episode", name, "moves", k, "expansions", len(net.calls))
    finally:
        restore_rng(saved)
        training.OthelloGame = OthelloGame
        training.OthelloMCTS = othelo_mcts.OthelloMCTS
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "episodes.npz"), **out)


def arena_fixture():
    specs = [
        # name, n, sims, c, seed, game, saltA, saltB, regime
        ("ar8_50", 8, 50, 1, 42, 0, 201, 202, "nep50"),
        ("ar8_100", 8, 100, 1, 42, 1, 203, 204, "nep50"),
        ("ar6_200_f64", 6, 200, 1, 42, 2, 205, 206, "f64"),
        ("ar6_800", 6, 800, 1, 42, 3, 207, 208, "nep50"),
    ]
    out = {}
    names = []
    saved = patched_rng()
    try:
        for name, n, sims, c, seed, game, sa, sb, regime in specs:
            Ctx.seed, Ctx.game, Ctx.ply, Ctx.counts = seed, game, 0, []
            CountingGame.log = []
            g = CountingGame(n)
            a1 = agents.NeuralNetworkOthelloAgent(g, StubNet(n, sa, 0, regime), sims, c)
            a2 = agents.NeuralNetworkOthelloAgent(g, StubNet(n, sb, 0, regime), sims, c)
            winner, points = agents.duel_between_agents(g, a1, a2)
            log = CountingGame.log
            out[f"{name}/meta"] = np.array([n, sims, seed, game, sa, sb, 0 if regime == "nep50" else 1, len(log)], dtype=np.int64)
            out[f"{name}/c"] = np.array([float(c)])
            out[f"{name}/player"] = np.array([x[2] for x in log], dtype=np.int8)
            out[f"{name}/action"] = np.array([x[3] for x in log], dtype=np.uint8)
            fb, fw = pack(g.board(BoardView.TWO_CHANNELS))
            out[f"{name}/final"] = np.array([fb, fw], dtype=np.uint64)
            out[f"{name}/result"] = np.array([1 if winner is a1 else -1, points], dtype=np.int32)
            names.append(name)
            print("arena", name, "moves", len(log), "winner", 1 if winner is a1 else -1, points)
    finally:
        restore_rng(saved)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "arena.npz"), **out)




def arena_random_fixture():
    """NeuralNetworkOthelloAgent vs RandomOthelloAgent (agents.py:20-24), both colour assignments: what main.py's
    evaluation rounds play (main.py:163-233).  random.choice is the patched TIE stream for both agents (distinct plies)."""
    specs = [
        # name, n, sims, c, seed, game, salt of the network, network colour (+1 BLACK / -1 WHITE), regime
        ("rnd8_nn_black", 8, 30, 1, 77, 0, 301, 1, "nep50"),
        ("rnd8_nn_white", 8, 30, 1, 77, 1, 302, -1, "nep50"),
        ("rnd6_nn_black_f64", 6, 60, 1, 77, 2, 303, 1, "f64"),
        ("rnd6_nn_white", 6, 25, 1, 77, 3, 304, -1, "nep50"),
    ]
    out, names = {}, []
    saved = patched_rng()
    try:
        for name, n, sims, c, seed, game, salt, colour, regime in specs:
            Ctx.seed, Ctx.game, Ctx.ply, Ctx.counts = seed, game, 0, []
            CountingGame.log = []
            g = CountingGame(n)
            nn_agent = agents.NeuralNetworkOthelloAgent(g, StubNet(n, salt, 0, regime), sims, c)
            rnd_agent = agents.RandomOthelloAgent(g)
            pair = (nn_agent, rnd_agent) if colour == 1 else (rnd_agent, nn_agent)
            winner, points = agents.duel_between_agents(g, *pair)
            log = CountingGame.log
            out[f"{name}/meta"] = np.array([n, sims, seed, game, salt, colour, 0 if regime == "nep50" else 1, len(log)], dtype=np.int64)
            out[f"{name}/c"] = np.array([float(c)])
            out[f"{name}/player"] = np.array([x[2] for x in log], dtype=np.int8)
            out[f"{name}/action"] = np.array([x[3] for x in log], dtype=np.uint8)
            fb, fw = pack(g.board(BoardView.TWO_CHANNELS))
            out[f"{name}/final"] = np.array([fb, fw], dtype=np.uint64)
            out[f"{name}/result"] = np.array([1 if winner is pair[0] else -1, points, 1 if winner is nn_agent else 0], dtype=np.int32)
            names.append(name)
            print("arena_random", name, "moves", len(log), "black wins" if winner is pair[0] else "white wins", points)
    finally:
        restore_rng(saved)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "arena_random.npz"), **out)


# ---------------------------------------------------------------- BaseNN (one-channel view) episode
class StubNetBNN(StubNet):
    """same integer-hash net, reached through the one-channel board view (othelo_mcts.py:17-18,85-86)"""
    network_type = NeuralNets.BNN

    def predict(self, board):
        b = np.asarray(board)
        assert b.ndim == 2
        return StubNet.predict(self, np.stack([b == 1, b == -1], axis=2))


def bnn_fixture():
    specs = [("ep6_bnn", 6, 40, 1, 1, 0.9, 1234, 9, 111, 0, "nep50"), ("ep8_bnn_f64", 8, 25, 1, 1, 0.9, 1234, 10, 112, 0, "f64")]
    out, names = {}, []
    saved = patched_rng()
    training.OthelloGame = CountingGame
    try:
        for name, n, sims, c, T, eg, seed, game, salt, keep, regime in specs:
            Ctx.seed, Ctx.game, Ctx.ply, Ctx.counts = seed, game, 0, []
            CountingGame.log = []
            net = StubNetBNN(n, salt, keep, regime)
            ex = training.execute_episode(n, net, c, sims, T, eg)
            log = CountingGame.log
            k = len(log)
            assert len(ex) == 8 * k
            out[f"{name}/meta"] = np.array([n, sims, seed, game, salt, keep, 0 if regime == "nep50" else 1, k], dtype=np.int64)
            out[f"{name}/params"] = np.array([float(c), float(T), float(eg)])
            out[f"{name}/action"] = np.array([x[3] for x in log], dtype=np.uint8)
            out[f"{name}/player"] = np.array([x[2] for x in log], dtype=np.int8)
            eb = np.zeros((8 * k, 2), np.uint64); ep = np.zeros(8 * k, np.int32); ez = np.zeros(8 * k, np.int8)
            for i, (b, p, z) in enumerate(ex):
                assert b.ndim == 2 and p.dtype == np.float64 and isinstance(z, int)
                eb[i] = pack(np.stack([b == 1, b == -1], axis=2))
                nz = np.argwhere(p == 1.0)
                ep[i] = nz[0][0] * n + nz[0][1]
                ez[i] = z
            out[f"{name}/ex_board"] = eb; out[f"{name}/ex_policy"] = ep; out[f"{name}/ex_z"] = ez
            out[f"{name}/dtype"] = np.array([str(ex[0][0].dtype)])
            names.append(name)
            print("bnn episode", name, "moves", k, "board dtype", ex[0][0].dtype)
    finally:
        restore_rng(saved)
        training.OthelloGame = OthelloGame
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "episodes_bnn.npz"), **out)


# ---------------------------------------------------------------- policy extraction at awkward temperatures (M10)
POLICY_TEMPS = (1.0 / 3.0, 0.5, 0.7, 1.5, 2.0, 3.0)


class PolicyRecordingMCTS(RecordingMCTS):
    pis = None

    def get_policy_action_probabilities(self, state, temperature):
        pi = super().get_policy_action_probabilities(state, temperature)
        PolicyRecordingMCTS.pis.append(np.array(pi, dtype=np.float64))
        return pi


def policy_temps_fixture():
    """othelo_mcts.py:64-67: `self.N(state, action) ** (1 / temperature)` is Python int ** float, i.e. libm pow on a float64, then np.sum"""
    out, names = {}, []
    specs = [("pt8", 8, 11, 0, "f64", 1, 100), ("pt6", 6, 21, 0, "nep50", 1, 300), ("pt6_sparse", 6, 22, 7, "nep50", 1, 60), ("pt4", 4, 31, 0, "f64", 2, 120)]
    for name, n, salt, keep, regime, c, sims in specs:
        root = OthelloGame.initial_board(n)
        m = othelo_mcts.OthelloMCTS(n, StubNet(n, salt, keep, regime), c)
        for _ in range(sims):
            m.simulate(root, OthelloPlayer.BLACK)
        cnt = np.zeros(64, np.int32)
        for a in m._get_state_actions(root):
            cnt[a[0] * 8 + a[1]] = m.N(root, a)
        out[f"{name}/meta"] = np.array([n, salt, keep, 0 if regime == "nep50" else 1, sims], dtype=np.int64)
        out[f"{name}/c"] = np.array([float(c)])
        out[f"{name}/root"] = np.array(pack(root), dtype=np.uint64)
        out[f"{name}/counts"] = cnt
        out[f"{name}/pi"] = np.stack([m.get_policy_action_probabilities(root, T) for T in POLICY_TEMPS])
        names.append(name)
        print("policy", name, "root visits", int(cnt.sum()), "max pi at T=1/3", float(out[f"{name}/pi"][0].max()))
    out["temps"] = np.array(POLICY_TEMPS)
    # one whole episode at T = 0.5 (6x6, 40 sims): moves as usual plus the pi of every move
    saved = patched_rng()
    training.OthelloGame = CountingGame
    training.OthelloMCTS = PolicyRecordingMCTS
    try:
        name, n, sims, c, T, eg, seed, game, salt, keep, regime = ("ep6_T05", 6, 40, 1, 0.5, 0.9, 4321, 11, 113, 0, "nep50")
        Ctx.seed, Ctx.game, Ctx.ply, Ctx.counts = seed, game, 0, []
        CountingGame.log = []
        PolicyRecordingMCTS.pis = []
        ex = training.execute_episode(n, StubNet(n, salt, keep, regime), c, sims, T, eg)
        log = CountingGame.log
        k = len(log)
        assert len(ex) == 8 * k and len(PolicyRecordingMCTS.pis) == k
        out[f"{name}/meta"] = np.array([n, sims, seed, game, salt, keep, 0 if regime == "nep50" else 1, k], dtype=np.int64)
        out[f"{name}/params"] = np.array([float(c), float(T), float(eg)])
        out[f"{name}/black"] = np.array([x[0] for x in log], dtype=np.uint64)
        out[f"{name}/white"] = np.array([x[1] for x in log], dtype=np.uint64)
        out[f"{name}/player"] = np.array([x[2] for x in log], dtype=np.int8)
        out[f"{name}/action"] = np.array([x[3] for x in log], dtype=np.uint8)
        out[f"{name}/counts"] = np.array(Ctx.counts, dtype=np.int32)
        out[f"{name}/pi"] = np.stack(PolicyRecordingMCTS.pis)
        out[f"{name}/ex_z"] = np.array([z for _, _, z in ex], dtype=np.int8)
        print("policy episode", name, "moves", k)
    finally:
        restore_rng(saved)
        training.OthelloGame = OthelloGame
        training.OthelloMCTS = othelo_mcts.OthelloMCTS
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "policy_temps.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["rules", "pairwise", "symmetries", "mcts", "episodes", "arena", "bnn", "arena_random", "policy_temps"]
    for w in which:
        globals()[w + "_fixture"]()
