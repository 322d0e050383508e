"""ctypes front end of the C oracle (test infrastructure only; see __init__)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboz_oracle.so")
_LIB = None

QMODE_NEP50, QMODE_F64 = 0, 1
VT_INT, VT_F32, VT_F64 = 0, 1, 2

EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float))


class EpisodeOut(C.Structure):
    _fields_ = [
        ("n_moves", C.c_int),
        ("black", C.c_uint64 * 64), ("white", C.c_uint64 * 64),
        ("player", C.c_int8 * 64), ("action", C.c_uint8 * 64), ("z", C.c_int8 * 64), ("greedy", C.c_uint8 * 64),
        ("counts", (C.c_int32 * 64) * 64),
        ("final_black", C.c_uint64), ("final_white", C.c_uint64),
        ("winner", C.c_int), ("points", C.c_int),
        ("stats", C.c_long * 4),
    ]


class ArenaOut(C.Structure):
    _fields_ = [
        ("n_moves", C.c_int),
        ("action", C.c_uint8 * 128), ("player", C.c_int8 * 128),
        ("final_black", C.c_uint64), ("final_white", C.c_uint64),
        ("winner", C.c_int), ("points", C.c_int),
    ]


class NNCtx(C.Structure):
    _fields_ = [("W", C.POINTER(C.POINTER(C.c_float))), ("C", C.c_int), ("nthreads", C.c_int), ("calls", C.c_long)]


def build(force=False):
    """Compile the oracle (gcc); a prebuilt .so that travelled with the tree is kept."""
    if os.path.exists(_SO) and not force:
        srcs = [os.path.join(_HERE, f) for f in ("oz_oracle.c", "oz_oracle_nn.c", "Makefile")]
        if all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs):
            return _SO
    subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True, capture_output=True)
    return _SO


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    try:
        build()
    except Exception:
        if not os.path.exists(_SO):
            raise
    L = C.CDLL(_SO)
    u64, i32, dbl, vp = C.c_uint64, C.c_int, C.c_double, C.c_void_p
    P = C.POINTER
    L.orc_rng.restype = u64; L.orc_rng.argtypes = [u64, u64, u64, u64]
    L.orc_stub_predict.restype = None
    L.orc_stub_predict.argtypes = [u64, u64, i32, u64, u64, P(C.c_float), P(C.c_float)]
    L.orc_initial_board.restype = None; L.orc_initial_board.argtypes = [i32, P(u64), P(u64)]
    L.orc_legal_mask.restype = u64; L.orc_legal_mask.argtypes = [u64, u64, i32, i32]
    L.orc_flip_mask.restype = u64; L.orc_flip_mask.argtypes = [u64, u64, i32, i32, i32]
    L.orc_apply_move.restype = None; L.orc_apply_move.argtypes = [P(u64), P(u64), i32, i32, i32]
    L.orc_finished.restype = i32; L.orc_finished.argtypes = [u64, u64, i32]
    L.orc_winner.restype = i32; L.orc_winner.argtypes = [u64, u64, i32, P(i32)]
    L.orc_game_play.restype = None; L.orc_game_play.argtypes = [P(u64), P(u64), i32, P(i32), P(i32), i32]
    L.orc_pairwise_sum.restype = dbl; L.orc_pairwise_sum.argtypes = [P(dbl), i32]
    L.orc_mcts_new.restype = vp; L.orc_mcts_new.argtypes = [i32, dbl, i32, EVAL_FN, vp]
    L.orc_mcts_new_stub.restype = vp; L.orc_mcts_new_stub.argtypes = [i32, dbl, i32, u64, u64]
    L.orc_mcts_free.restype = None; L.orc_mcts_free.argtypes = [vp]
    L.orc_mcts_simulate.restype = dbl; L.orc_mcts_simulate.argtypes = [vp, u64, u64, i32, P(i32)]
    L.orc_mcts_num_nodes.restype = i32; L.orc_mcts_num_nodes.argtypes = [vp]
    L.orc_mcts_stats.restype = None; L.orc_mcts_stats.argtypes = [vp, P(C.c_long), P(i32)]
    L.orc_mcts_dump_node.restype = None
    L.orc_mcts_dump_node.argtypes = [vp, i32, P(u64), P(u64), P(i32), P(i32), P(u64), P(i32), P(dbl), P(C.c_uint8), P(dbl)]
    L.orc_mcts_find.restype = i32; L.orc_mcts_find.argtypes = [vp, u64, u64]
    L.orc_mcts_counts.restype = i32; L.orc_mcts_counts.argtypes = [vp, u64, u64, P(i32), P(u64)]
    L.orc_mcts_policy.restype = i32; L.orc_mcts_policy.argtypes = [vp, u64, u64, dbl, u64, P(dbl), P(i32)]
    L.orc_policy_from_counts.restype = i32; L.orc_policy_from_counts.argtypes = [i32, P(i32), u64, dbl, u64, P(dbl)]
    L.orc_episode.restype = i32; L.orc_episode.argtypes = [vp, i32, dbl, dbl, u64, u64, i32, P(EpisodeOut)]
    L.orc_episode_sched.restype = i32; L.orc_episode_sched.argtypes = [vp, i32, i32, i32, dbl, dbl, u64, u64, i32, P(EpisodeOut)]
    L.orc_arena.restype = i32; L.orc_arena.argtypes = [vp, vp, i32, u64, u64, P(ArenaOut)]
    L.orc_arena_plies.restype = i32; L.orc_arena_plies.argtypes = [vp, vp, i32, u64, u64, i32, P(ArenaOut)]
    L.orc_symmetry_perms.restype = None; L.orc_symmetry_perms.argtypes = [i32, P(C.c_int32)]
    L.orc_nn_num_weights.restype = i32
    L.orc_nn_forward_f32.restype = None
    L.orc_nn_forward_f32.argtypes = [P(P(C.c_float)), i32, i32, P(u64), P(u64), i32, P(C.c_float), P(C.c_float), i32]
    L.orc_nn_max_threads.restype = i32
    _LIB = L
    return L


# ----------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------
def pack_board(board):
    """(n,n,2) array-like -> (ch0, ch1) uint64 bitboards, bit r*8+c."""
    b = np.asarray(board).astype(bool)
    n = b.shape[0]
    c0 = c1 = 0
    for r in range(n):
        for c in range(n):
            if b[r, c, 0]:
                c0 |= 1 << (r * 8 + c)
            if b[r, c, 1]:
                c1 |= 1 << (r * 8 + c)
    return c0, c1


def unpack_board(c0, c1, n):
    b = np.zeros((n, n, 2), dtype=bool)
    for r in range(n):
        for c in range(n):
            b[r, c, 0] = (int(c0) >> (r * 8 + c)) & 1
            b[r, c, 1] = (int(c1) >> (r * 8 + c)) & 1
    return b


def mask_to_squares(m):
    return [s for s in range(64) if (int(m) >> s) & 1]


def stub_predict(own, opp, n, salt=0, keep_mask=0):
    L = lib()
    pi = (C.c_float * (n * n))()
    v = C.c_float()
    L.orc_stub_predict(own, opp, n, salt, keep_mask, pi, C.byref(v))
    return np.frombuffer(pi, dtype=np.float32).reshape(n, n).copy(), np.float32(v.value)


def policy_from_counts(n, counts, legal, T, tie_u=0):
    """get_policy_action_probabilities (othelo_mcts.py:51-67) from root visit counts by square and the legal mask"""
    cnt = np.ascontiguousarray(counts, dtype=np.int32)
    out = np.zeros(n * n, np.float64)
    lib().orc_policy_from_counts(n, cnt.ctypes.data_as(C.POINTER(C.c_int)), int(legal), float(T), int(tie_u), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out.reshape(n, n)


def pairwise_sum(a):
    a = np.ascontiguousarray(a, dtype=np.float64).ravel()
    return lib().orc_pairwise_sum(a.ctypes.data_as(C.POINTER(C.c_double)), a.size)


def symmetry_perms(n):
    out = np.zeros((8, n * n), dtype=np.int32)
    lib().orc_symmetry_perms(n, out.ctypes.data_as(C.POINTER(C.c_int32)))
    return out


class Mcts:
    """One OthelloMCTS instance of the oracle.  evaluator: None -> builtin stub
    (salt, keep_mask); otherwise a Python callable (own, opp, n) -> (pi (n*n,) float32, v float)."""

    def __init__(self, n, c=1.0, qmode=QMODE_NEP50, evaluator=None, salt=0, keep_mask=0):
        self.L = lib()
        self.n = n
        self._cb = None
        if evaluator is None:
            self.h = self.L.orc_mcts_new_stub(n, float(c), qmode, salt, keep_mask)
        elif isinstance(evaluator, tuple) and evaluator[0] == "c":
            # ("c", function pointer, context pointer, keepalive)
            self._keep = evaluator[3]
            self.h = self.L.orc_mcts_new(n, float(c), qmode, C.cast(evaluator[1], EVAL_FN), evaluator[2])
        else:
            def _cb(ctx, own, opp, nn, pi_p, v_p):
                pi, v = evaluator(int(own), int(opp), nn)
                pi = np.asarray(pi, dtype=np.float32).ravel()
                for i in range(nn * nn):
                    pi_p[i] = pi[i]
                v_p[0] = float(v)
            self._cb = EVAL_FN(_cb)
            self.h = self.L.orc_mcts_new(n, float(c), qmode, self._cb, None)

    def __del__(self):
        try:
            if self.h:
                self.L.orc_mcts_free(self.h)
                self.h = None
        except Exception:
            pass

    def simulate(self, black, white, player):
        vt = C.c_int()
        v = self.L.orc_mcts_simulate(self.h, black, white, player, C.byref(vt))
        return v, vt.value

    def num_nodes(self):
        return self.L.orc_mcts_num_nodes(self.h)

    def stats(self):
        s = (C.c_long * 4)()
        d = C.c_int()
        self.L.orc_mcts_stats(self.h, s, C.byref(d))
        return dict(visits=s[0], expansions=s[1], terminal=s[2], fallback=s[3], max_depth=d.value)

    def dump_node(self, i):
        k0, k1, legal = C.c_uint64(), C.c_uint64(), C.c_uint64()
        Ns, ei = C.c_int(), C.c_int()
        N = np.zeros(64, np.int32); Q = np.zeros(64, np.float64); qt = np.zeros(64, np.uint8); Pp = np.zeros(64, np.float64)
        self.L.orc_mcts_dump_node(self.h, i, C.byref(k0), C.byref(k1), C.byref(Ns), C.byref(ei), C.byref(legal),
                                  N.ctypes.data_as(C.POINTER(C.c_int)), Q.ctypes.data_as(C.POINTER(C.c_double)),
                                  qt.ctypes.data_as(C.POINTER(C.c_uint8)), Pp.ctypes.data_as(C.POINTER(C.c_double)))
        return dict(k0=k0.value, k1=k1.value, Ns=Ns.value, edges_init=ei.value, legal=legal.value, N=N, Q=Q, qtag=qt, P=Pp)

    def dump(self):
        return [self.dump_node(i) for i in range(self.num_nodes())]

    def counts(self, k0, k1):
        cnt = np.zeros(64, np.int32)
        legal = C.c_uint64()
        rc = self.L.orc_mcts_counts(self.h, k0, k1, cnt.ctypes.data_as(C.POINTER(C.c_int)), C.byref(legal))
        return rc, cnt, legal.value

    def policy(self, k0, k1, T, tie_u=0):
        out = np.zeros(self.n * self.n, np.float64)
        arg = C.c_int()
        rc = self.L.orc_mcts_policy(self.h, k0, k1, float(T), tie_u, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(arg))
        if rc == 2:
            raise KeyError("node never selected from")
        return out.reshape(self.n, self.n), arg.value

    def episode(self, sims, T, e_greedy, seed, game_id, max_moves=-1, sims_pre=None, pre_plies=0):
        """execute_episode (training.py:26-72); sims_pre / pre_plies: num_simulations on the first pre_plies plies
        (a slot of the build's engine after oz_selfplay_stagger)"""
        out = EpisodeOut()
        rc = self.L.orc_episode_sched(self.h, sims, sims if sims_pre is None else sims_pre, pre_plies, float(T), float(e_greedy),
                                      seed, game_id, max_moves, C.byref(out))
        if rc < 0:
            raise KeyError("orc_episode: KeyError path (num_simulations too small)")
        k = out.n_moves
        return dict(
            n_moves=k, finished=(rc == 0),
            black=np.array(out.black[:k], dtype=np.uint64), white=np.array(out.white[:k], dtype=np.uint64),
            player=np.array(out.player[:k], dtype=np.int8), action=np.array(out.action[:k], dtype=np.uint8),
            z=np.array(out.z[:k], dtype=np.int8), greedy=np.array(out.greedy[:k], dtype=np.uint8),
            counts=np.array([list(out.counts[i]) for i in range(k)], dtype=np.int32).reshape(k, 64),
            final_black=out.final_black, final_white=out.final_white, winner=out.winner, points=out.points,
            stats=dict(visits=out.stats[0], expansions=out.stats[1], terminal=out.stats[2], fallback=out.stats[3]),
        )


def arena(ma, mb, sims, seed, game_id, max_plies=-1):
    """ma / mb = Mcts of the BLACK / WHITE agent; None = RandomOthelloAgent on that colour; max_plies >= 0 stops the duel
    after that many plies (finished=False if the game is not over: winner / points then describe the unfinished board)"""
    out = ArenaOut()
    rc = lib().orc_arena_plies(ma.h if ma is not None else None, mb.h if mb is not None else None, sims, seed, game_id, max_plies, C.byref(out))
    if rc < 0:
        raise KeyError("orc_arena: KeyError path")
    k = out.n_moves
    return dict(n_moves=k, finished=(rc == 0), action=np.array(out.action[:k], dtype=np.uint8), player=np.array(out.player[:k], dtype=np.int8),
                final_black=out.final_black, final_white=out.final_white, winner=out.winner, points=out.points)


class CNet:
    """float32 C restatement of OthelloNN with a weight list in Keras get_weights() order."""

    def __init__(self, weights, n, channels=512, nthreads=0):
        self.L = lib()
        self.n, self.C = n, channels
        self.w = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        assert len(self.w) == 40
        arr = (C.POINTER(C.c_float) * 40)(*[w.ctypes.data_as(C.POINTER(C.c_float)) for w in self.w])
        self._arr = arr
        self.ctx = NNCtx(C.cast(arr, C.POINTER(C.POINTER(C.c_float))), channels, nthreads, 0)
        self.nthreads = nthreads

    def forward(self, own, opp):
        own = np.ascontiguousarray(own, dtype=np.uint64); opp = np.ascontiguousarray(opp, dtype=np.uint64)
        B = own.size
        pi = np.zeros((B, self.n * self.n), np.float32); v = np.zeros(B, np.float32)
        self.L.orc_nn_forward_f32(self._arr, self.n, self.C, own.ctypes.data_as(C.POINTER(C.c_uint64)),
                                  opp.ctypes.data_as(C.POINTER(C.c_uint64)), B,
                                  pi.ctypes.data_as(C.POINTER(C.c_float)), v.ctypes.data_as(C.POINTER(C.c_float)), self.nthreads)
        return pi, v

    def evaluator(self):
        """("c", fnptr, ctx, keepalive) tuple accepted by Mcts(evaluator=...)."""
        fn = C.cast(self.L.orc_nn_eval_cb, C.c_void_p)
        return ("c", fn, C.cast(C.pointer(self.ctx), C.c_void_p), self)
