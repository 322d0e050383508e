"""OthelloMCTS -- drop-in for othelo_mcts.py:9-88 + MCTS/__init__.py:19-187 on the HIP search kernels.

One instance = one per-game search table on the GPU (`oz_mcts` with a single slot).  `simulate(state,
player)` runs ONE simulation like the reference; when the network is a native `NNetWrapper` /
`StubNetWrapper` the leaf is evaluated on the device, otherwise (any duck-typed object with `.predict`)
the leaf board is handed to `neural_network.predict` on the host and its (pi, v) go back to the expand /
backup kernel -- the tree never leaves the GPU either way.
"""
import ctypes as C
import random

import numpy as np

from . import _lib
from .NNet import NeuralNets
from .Othello import OthelloGame, OthelloPlayer


class OthelloMCTS:
    def __init__(self, board_size, neural_network, degree_exploration, q_mode=_lib.QMODE_F64,
                 node_cap=8192):
        """q_mode: OZ_QMODE_F64 = the NumPy 1.18.5 promotion the reference pins (requirements.txt:19),
        OZ_QMODE_NEP50 = what NumPy >= 2 computes (SURVEY.md R-FP)."""
        self._board_size = board_size
        self._neural_network = neural_network
        # othelo_mcts.py:15-18: ONN nets see the two-channel board, BNN nets the one-channel (+1 / -1) view
        self._one_channel = getattr(neural_network.network_type, "name", "") == "BNN"
        self.degree_explorarion = degree_exploration
        self._q_mode = q_mode
        self._node_cap = node_cap
        self._h = C.c_void_p()
        lib = _lib.require_gpu()
        _lib.check(lib.oz_mcts_create(C.byref(self._h), board_size, 1, node_cap, float(degree_exploration), q_mode))
        self._native = getattr(neural_network, "_h", None) is not None
        self._root = None

    def __del__(self):
        try:
            if self._h:
                _lib.load().oz_mcts_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    # ---- helpers
    def _set_root(self, own, opp):
        if self._root != (own, opp):
            a, b = np.array([own], np.uint64), np.array([opp], np.uint64)
            _lib.check(_lib.load().oz_mcts_set_roots(self._h, _lib.p_u64(a), _lib.p_u64(b), None))
            self._root = (own, opp)

    def _last_value(self):
        val, vt = np.zeros(1, np.float64), np.zeros(1, np.int32)
        _lib.check(_lib.load().oz_mcts_last_value(self._h, _lib.p_f64(val), _lib.p_i32(vt), None))
        if vt[0] == _lib.VT_INT:
            return int(val[0])
        if vt[0] == _lib.VT_F32:
            return np.float32(val[0])
        return float(val[0])

    def _canonical(self, state, player):
        c0, c1 = _lib.pack_board(state)
        return (c1, c0) if player is OthelloPlayer.WHITE or getattr(player, "value", 1) == -1 else (c0, c1)

    # ---- reference surface
    def simulate(self, state, player):
        """othelo_mcts.py:22-26 + MCTS/__init__.py:30-71; `state` is not mutated."""
        return self.simulate_n(state, player, 1)

    def simulate_n(self, state, player, count):
        """`count` consecutive simulate(state, player) calls; returns the value of the last one."""
        lib = _lib.load()
        own, opp = self._canonical(state, player)
        self._set_root(own, opp)
        n = self._board_size
        if self._native:
            _lib.check(lib.oz_mcts_simulate(self._h, self._neural_network._h, int(count)))
            return self._last_value()
        status, lo, lp = np.zeros(1, np.int32), np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        pi, v = np.zeros((1, n * n), np.float32), np.zeros(1, np.float32)
        for _ in range(int(count)):
            _lib.check(lib.oz_mcts_select(self._h))
            _lib.check(lib.oz_mcts_leaves(self._h, _lib.p_i32(status), _lib.p_u64(lo), _lib.p_u64(lp)))
            if status[0] == _lib.LEAF_EVAL:
                leaf = _lib.unpack_board(int(lo[0]), int(lp[0]), n)
                if self._one_channel:                      # othelo_mcts.py:85-86
                    leaf = OthelloGame.convert_to_one_channel_board(leaf)
                p, val = self._neural_network.predict(leaf)
                pi[0] = np.asarray(p, dtype=np.float32).reshape(-1)
                v[0] = val
            _lib.check(lib.oz_mcts_backup(self._h, _lib.p_f32(pi), _lib.p_f32(v)))
        return self._last_value()

    def _counts(self, state):
        """visit counts of a mover-canonical state: (rc, counts[64], legal mask)"""
        own, opp = _lib.pack_board(state)
        self._set_root(own, opp)
        cnt, legal, rc = np.zeros(64, np.int32), np.zeros(1, np.uint64), np.zeros(1, np.int32)
        _lib.check(_lib.load().oz_mcts_root_counts(self._h, _lib.p_i32(cnt), _lib.p_u64(legal), _lib.p_i32(rc)))
        return int(rc[0]), cnt, int(legal[0])

    def N(self, state, action=None):
        """MCTS/__init__.py:73-84,172-175: 0 for an unknown state, Ns without an action, Nsa[action] otherwise
        (KeyError if the state was never selected from)."""
        rc, cnt, legal = self._counts(state)
        if rc == 1:
            return 0
        if action is None:
            return int(cnt.sum())                     # Ns == sum of Nsa (both are incremented together)
        sq = int(action[0]) * 8 + int(action[1])
        if rc == 2 or not (legal >> sq) & 1:
            raise KeyError(tuple(action))
        return int(cnt[sq])

    def get_state_actions(self, state):
        """othelo_mcts.py:40-41: legal actions of BLACK (= channel 0) as tuples, ascending row-major."""
        return [tuple(int(x) for x in a) for a in OthelloGame.get_player_valid_actions(state, OthelloPlayer.BLACK)]

    def get_policy_action_probabilities(self, state, temperature):
        """othelo_mcts.py:51-67, evaluated with the same NumPy / random calls as the reference."""
        n = self._board_size
        rc, cnt, legal = self._counts(state)
        if rc == 2:
            raise KeyError("state was expanded but never selected from (num_simulations < 2)")
        probabilities = np.zeros((n, n))
        actions = [(s >> 3, s & 7) for s in range(64) if (legal >> s) & 1]
        if temperature == 0:
            for row, col in actions:
                probabilities[row, col] = int(cnt[row * 8 + col])
            bests = np.argwhere(probabilities == probabilities.max())
            row, col = random.choice(bests)
            probabilities = np.zeros((n, n))
            probabilities[row, col] = 1
            return probabilities
        for row, col in actions:
            probabilities[row, col] = int(cnt[row * 8 + col]) ** (1 / temperature)
        return probabilities / (np.sum(probabilities) or 1)

    # ---- the game hooks MCTS/__init__.py:86-166 declares abstract and othelo_mcts.py:28-49,69-88 fills in.  The search never calls
    # them here (the kernels carry the rules); callers that use them directly get the LIBRARY's rules kernels through the C ABI
    # (oz_rules_status / oz_rules_apply_moves / oz_rules_legal_moves on the packed board), i.e. the same bit-parallel code the search runs.
    # (no CPU fallback by design: without a GPU / the built library they raise OzLibraryError through require_gpu, like every compute call)
    def _rules_status(self, state):
        _lib.require_gpu()
        c0, c1 = (np.array([x], np.uint64) for x in _lib.pack_board(state))
        fin, p0, p1, win = np.zeros(1, np.uint8), np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(1, np.int8)
        _lib.check(_lib.load().oz_rules_status(_lib.p_u64(c0), _lib.p_u64(c1), self._board_size, 1, _lib.p_u8(fin), _lib.p_i32(p0),
                                               _lib.p_i32(p1), _lib.p_i8(win)))
        return bool(fin[0]), int(win[0])

    def _legal_mask(self, own, opp):
        _lib.require_gpu()
        legal = np.zeros(1, np.uint64)
        _lib.check(_lib.load().oz_rules_legal_moves(_lib.p_u64(np.array([own], np.uint64)), _lib.p_u64(np.array([opp], np.uint64)),
                                                    self._board_size, 1, _lib.p_u64(legal)))
        return int(legal[0])

    def is_terminal_state(self, state):
        return self._rules_status(state)[0]

    def get_state_reward(self, state):
        return self._rules_status(state)[1]                    # +1: channel 0 holds at least as many discs (a draw counts for it), else -1

    def get_next_state(self, state, action):
        own, opp = _lib.pack_board(state)
        sq = np.array([int(action[0]) * 8 + int(action[1])], np.uint8)
        o2, p2 = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        _lib.check(_lib.load().oz_rules_apply_moves(_lib.p_u64(np.array([own], np.uint64)), _lib.p_u64(np.array([opp], np.uint64)), _lib.p_u8(sq),
                                                    self._board_size, 1, _lib.p_u64(o2), _lib.p_u64(p2)))
        own, opp = int(o2[0]), int(p2[0])
        if self._legal_mask(opp, own):                         # the opponent can answer: it becomes the mover (channel 0); a pass keeps the orientation
            own, opp = opp, own
        nxt = _lib.unpack_board(own, opp, self._board_size)
        return nxt.astype(state.dtype) if isinstance(state, np.ndarray) and nxt.dtype != state.dtype else nxt      # the reference returns the input's dtype (np.copy + in-place flips)

    def _neural_network_predict(self, state):
        """one evaluation per distinct board of this search (the reference's per-instance dict, keyed by the exact board)"""
        seen = self.__dict__.setdefault("_predict_cache", {})
        key = _lib.pack_board(state)
        hit = seen.get(key)
        if hit is None:
            hit = seen[key] = self._neural_network.predict(OthelloGame.convert_to_one_channel_board(state) if self._one_channel else state)
        return hit

    def get_state_value(self, state):
        return self._neural_network_predict(state)[1]

    def get_state_actions_propabilities(self, state):          # (sic: the reference's spelling)
        return self._neural_network_predict(state)[0]

    def _mask_valid_moves(self, state):
        n = self._board_size
        legal = self._legal_mask(*_lib.pack_board(state))
        return np.array([[(legal >> (r * 8 + c)) & 1 for c in range(n)] for r in range(n)], dtype=np.float64)

    def moves_scaled_by_valid_moves(self, state):
        return self.get_state_actions_propabilities(state) * self._mask_valid_moves(state)

    # ---- inspection (parity tests)
    def dump(self):
        lib = _lib.load()
        nn = np.zeros(1, np.int32)
        _lib.check(lib.oz_mcts_num_nodes(self._h, _lib.p_i32(nn)))
        out = []
        for i in range(int(nn[0])):
            own, opp, legal, Ns = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int32()
            N, Q, qt, P = np.zeros(64, np.int32), np.zeros(64, np.float64), np.zeros(64, np.uint8), np.zeros(64, np.float64)
            _lib.check(lib.oz_mcts_dump_node(self._h, 0, i, C.byref(own), C.byref(opp), C.byref(Ns), C.byref(legal),
                                             _lib.p_i32(N), _lib.p_f64(Q), _lib.p_u8(qt), _lib.p_f64(P)))
            out.append(dict(k0=own.value, k1=opp.value, Ns=Ns.value, legal=legal.value, N=N, Q=Q, qtag=qt, P=P))
        return out

    def stats(self):
        s = np.zeros(5, np.int64)
        _lib.check(_lib.load().oz_mcts_stats(self._h, _lib.p_i64(s)))
        return dict(simulations=int(s[0]), visits=int(s[1]), expansions=int(s[2]), terminal=int(s[3]), fallback=int(s[4]))
