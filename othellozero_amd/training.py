"""Self-play drivers -- drop-in for training.py:13-72 plus the batched engine the reference lacks.

`execute_episode(...)` keeps the reference's signature, return layout and host-side random draws
(`random.random`, `np.random.choice`, `random.choice`) so existing callers (workers.py:83-84,
pickle_training.py:26-27) see the same behaviour; all rule / search / network work runs on the GPU.
`selfplay_batch(...)` is the MI355X-native path: thousands of games in lock step, resident in HBM,
counter-based RNG streams, one batched network evaluation per simulation step.
"""
import ctypes as C
import logging
import random

import numpy as np

from . import _lib
from .NNet import NeuralNets
from .Othello import BoardView, OthelloGame, OthelloPlayer
from .othelo_mcts import OthelloMCTS


def training_example_symmetries(board, policy):
    """training.py:13-23: 8 (board, policy) pairs, rot90 k=1..4 (CCW) each with fliplr first, then without."""
    symetric_examples = []
    for rotation in range(1, 5):
        for flip in (True, False):
            b, p = np.rot90(board, k=rotation), np.rot90(policy, k=rotation)
            if flip:
                b, p = np.fliplr(b), np.fliplr(p)
            symetric_examples.append((b, p))
    return symetric_examples


def execute_episode(board_size, neural_network, degree_exploration, num_simulations, policy_temperature, e_greedy,
                    q_mode=_lib.QMODE_F64, snapshot_boards=False):
    """training.py:26-72.  Returns [(board (n,n,2) bool, one-hot policy (n,n) float64, z int), ...], 8 per move.

    snapshot_boards=False reproduces the reference exactly, including its aliasing quirk: the
    returned boards are views of the live game array, so every example shows the FINAL position.  Pass
    snapshot_boards=True to store the position at the time of the move instead (what training wants)."""
    examples = []
    game = OthelloGame(board_size)
    mcts = OthelloMCTS(board_size, neural_network, degree_exploration, q_mode=q_mode,
                       node_cap=num_simulations * (board_size * board_size - 3) + 64)
    # training.py:34-37 (BNN examples are one-channel boards, a fresh array per round: no aliasing for them)
    board_view_type = BoardView.ONE_CHANNEL if getattr(neural_network.network_type, "name", "") == "BNN" else BoardView.TWO_CHANNELS

    while not game.has_finished():
        state = game.board(BoardView.TWO_CHANNELS)
        mcts.simulate_n(state, game.current_player, num_simulations)
        if game.current_player == OthelloPlayer.WHITE:
            state = OthelloGame.invert_board(state)
        policy = mcts.get_policy_action_probabilities(state, policy_temperature)

        coin = random.random()                      # e-greedy, training.py:51-56
        if coin <= e_greedy:
            action = np.argwhere(policy == policy.max())[0]
        else:
            actions = mcts.get_state_actions(state)
            action = actions[np.random.choice(len(actions))]

        action_choosed = np.zeros((board_size, board_size))
        action_choosed[action[0]][action[1]] = 1
        board_now = game.board(board_view_type)
        if snapshot_boards:
            board_now = np.copy(board_now)
        for board_example, policy_example in training_example_symmetries(board_now, action_choosed):
            examples.append((board_example, policy_example, game.current_player))
        game.play(*action)

    winner, winner_points = game.get_winning_player()
    logging.info(f'Episode finished: The winner obtained {winner_points} points.')
    return [(state, policy, 1 if winner == player else -1) for state, policy, player in examples]


def duel_between_neural_networks(board_size, neural_network_1, neural_network_2, degree_exploration, num_simulations,
                                 fixed=False):
    """One game, neural_network_1 as BLACK against neural_network_2 (the name workers.py:14-15 and pickle_training.py:36 import;
    reference body: training.py:75-88).  What the reference really does is play the game and then fail: it looks the whole
    `(agent, points)` result of duel_between_agents up in a dict keyed by agent, i.e. KeyError -- recorded from a run of the
    reference in tests/golden/drivers_misc.json and reproduced here.  fixed=True returns what was meant: 0 if the first
    network won, 1 otherwise."""
    from .agents import NeuralNetworkOthelloAgent, duel_between_agents
    board = OthelloGame(board_size)
    black, white = (NeuralNetworkOthelloAgent(board, net, num_simulations, degree_exploration)
                    for net in (neural_network_1, neural_network_2))
    outcome = duel_between_agents(board, black, white)          # (winning agent, its points)
    if not fixed:
        raise KeyError(outcome)
    return 0 if outcome[0] is black else 1


def evaluate_neural_network(board_size, total_iterations, neural_network, num_simulations, degree_exploration,
                            agent_class, agent_arguments, fixed=False):
    """`total_iterations` games of the network's search agent against agent_class(game, *agent_arguments); returns the number
    of games counted as won (reference body: training.py:91-118, imported by workers.py:14-15).  Colours come from one
    random.shuffle of [opponent, network] per game, as there.  The reference compares the `(agent, points)` result with
    the agent itself, so it never counts a win and returns 0 (tests/golden/drivers_misc.json); fixed=True compares the agent."""
    from .agents import NeuralNetworkOthelloAgent, duel_between_agents
    wins = 0
    for _ in range(total_iterations):
        board = OthelloGame(board_size)
        ours = NeuralNetworkOthelloAgent(board, neural_network, num_simulations, degree_exploration)
        seats = [agent_class(board, *agent_arguments), ours]
        random.shuffle(seats)
        victor, _points = duel_between_agents(board, *seats)
        wins += int(fixed and victor is ours)
    logging.info(f'Neural Network Evaluation: {wins} of {total_iterations} games counted as won')
    return wins


# ---------------------------------------------------------------- batched engine
class SelfPlayEngine:
    """num_games concurrent execute_episode instances on one GPU (C ABI: oz_selfplay_*)."""

    def __init__(self, neural_network, board_size=8, num_games=4096, num_simulations=100, degree_exploration=1.0,
                 policy_temperature=1.0, e_greedy=0.9, seed=1234, first_game_id=0, game_id_stride=0,
                 q_mode=_lib.QMODE_F64, refill=False, node_cap=0, record_cap=0, dedup=True, batch_cap=0, eval_cache=False):
        """dedup: cross-game leaf de-duplication (a board several games reach in one batch is evaluated once; no record changes);
        batch_cap: leaves per network batch of the free-running driver (0 = none; see preferred_batch_cap);
        eval_cache: take (pi, v) of boards the network has evaluated before from its persistent cache (NNetWrapper.set_eval_cache) --
        the reference's per-search _predict_cache (othelo_mcts.py:82-88) across batches, games and refilled slots; no record changes"""
        lib = _lib.require_gpu()
        assert getattr(neural_network, "_h", None) is not None, "SelfPlayEngine needs a native NNetWrapper / StubNetWrapper"
        self.net = neural_network
        self.cfg = _lib.SelfplayConfig(
            n=board_size, num_games=num_games, sims=num_simulations, q_mode=q_mode, c=float(degree_exploration),
            temperature=float(policy_temperature), e_greedy=float(e_greedy), seed=seed, first_game_id=first_game_id,
            game_id_stride=game_id_stride, refill=1 if refill else 0, node_cap=node_cap,
            record_cap=record_cap, dedup=_lib.DEDUP_ON if dedup else _lib.DEDUP_OFF, batch_cap=int(batch_cap),
            eval_cache=1 if eval_cache else 0)
        self._h = C.c_void_p()
        _lib.check(lib.oz_selfplay_create(C.byref(self._h), C.byref(self.cfg), neural_network._h))
        self.n, self.num_games = board_size, num_games

    def __del__(self):
        try:
            if self._h:
                _lib.load().oz_selfplay_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def run(self, rounds=1, sync=True):
        """`rounds` move rounds: every live game runs num_simulations simulations and plays one move."""
        _lib.check(_lib.load().oz_selfplay_run(self._h, int(rounds)))
        if sync:
            self.sync()

    def run_steps(self, steps, sync=True):
        """free-running form of run(): `steps` network batches; every live game plays on by itself (its move when the
        simulations of the move are complete, simulations that need no network, the next first-visit leaf), so batches stay
        full; per game the simulations / moves / records are exactly those of run()"""
        _lib.check(_lib.load().oz_selfplay_run_steps(self._h, int(steps)))
        if sync:
            self.sync()

    def set_batch_cap(self, cap):
        """free-running driver: at most `cap` leaves per network batch (0 = no cap); a leaf that finds no slot waits for the next batch,
        the slot order rotates so that every game is served.  Records do not change; the launches become whole grid rounds at the right
        cap (see `preferred_batch_cap`)"""
        _lib.check(_lib.load().oz_selfplay_set_batch_cap(self._h, int(cap)))

    def set_dedup(self, enable):
        """cross-game leaf de-duplication on / off from the next batch on"""
        _lib.check(_lib.load().oz_selfplay_set_dedup(self._h, 1 if enable else 0))

    def stagger(self, sims_pre=None, sync=True):
        """continuous self-play (refill=True), first call only: advance slot g (g * P) // num_games plies into its first game
        (P = n*n - 4) by searched moves at `sims_pre` simulations each (default: num_simulations), so that the engine holds
        games at every stage like a long-running service and every later move round completes about num_games / P games"""
        _lib.check(_lib.load().oz_selfplay_stagger(self._h, int(sims_pre or self.cfg.sims)))
        if sync:
            self.sync()

    def profile(self, enable=True):
        """HIP-event timing of the tree kernels on the launch stream (the evaluator's launches are always timed)"""
        _lib.check(_lib.load().oz_selfplay_profile(self._h, 1 if enable else 0))

    def profile_read(self, reset=False):
        """{kernel: (ms_total, launches)} for _lib.TREE_KERNELS"""
        k = len(_lib.TREE_KERNELS)
        ms, cnt = np.zeros(k, np.float64), np.zeros(k, np.int64)
        _lib.check(_lib.load().oz_selfplay_profile_read(self._h, _lib.p_f64(ms), _lib.p_i64(cnt), 1 if reset else 0))
        return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(_lib.TREE_KERNELS)}

    def sync(self):
        _lib.check(_lib.load().oz_selfplay_sync(self._h))

    def stats(self):
        s = _lib.SelfplayStats()
        _lib.check(_lib.load().oz_selfplay_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def state(self):
        G = self.num_games
        b, w, gid = np.zeros(G, np.uint64), np.zeros(G, np.uint64), np.zeros(G, np.uint64)
        p, f, ply = np.zeros(G, np.int8), np.zeros(G, np.uint8), np.zeros(G, np.int32)
        _lib.check(_lib.load().oz_selfplay_state(self._h, _lib.p_u64(b), _lib.p_u64(w), _lib.p_i8(p), _lib.p_u8(f),
                                                 _lib.p_i32(ply), _lib.p_u64(gid)))
        return dict(black=b, white=w, player=p, finished=f, ply=ply, game_id=gid)

    def last_counts(self):
        c = np.zeros((self.num_games, 64), np.int32)
        _lib.check(_lib.load().oz_selfplay_last_counts(self._h, _lib.p_i32(c)))
        return c

    def records(self):
        """move records of the games completed so far (numpy structured array, _lib.RECORD_DTYPE),
        sorted by (game_id, ply)."""
        total = self.stats()["records"]
        out = np.zeros(max(total, 1), dtype=_lib.RECORD_DTYPE)
        written = C.c_int64()
        _lib.check(_lib.load().oz_selfplay_records(self._h, out.ctypes.data_as(C.c_void_p), total, C.byref(written)))
        out = out[:written.value]
        return out[np.lexsort((out["ply"], out["game_id"]))]

    def records_to_device(self, device_ptr, max_records):
        written = C.c_int64()
        _lib.check(_lib.load().oz_selfplay_records_device(self._h, C.c_void_p(device_ptr), max_records, C.byref(written)))
        return written.value

    def eval_time(self):
        ms, launches, leaves = C.c_double(), C.c_int64(), C.c_int64()
        _lib.check(_lib.load().oz_selfplay_eval_time(self._h, C.byref(ms), C.byref(launches), C.byref(leaves)))
        return dict(ms=ms.value, launches=launches.value, leaves=leaves.value)

    def play_to_end(self, max_rounds=None):
        max_rounds = max_rounds or self.n * self.n
        for _ in range(max_rounds):
            self.run(4)
            if self.stats()["live_games"] == 0:
                break
        return self.records()


def preferred_batch_cap(board_size, num_games, channels=512, precision="f16x2"):
    """the batch cap at which conv3 -- the dominant launch -- is a whole number of rounds of 256 x 256 tiles on the chip's 256 CUs:
    the largest L <= num_games with ceil((n-2)^2 * L / 256) * (channels / 256) = 256 * r; 0 (no cap) when that is num_games itself or the
    network does not use those tiles (4096 8x8 games, 512 filters: 3640 = 4.0 rounds; 6x6: no cap).
    precision "bf16x3": no cap -- k_gemm_b3's one tile is 128 x 256, on which 4096 leaves are 9.0 rounds of conv3 AND 4.0 of conv4 (3640: 8.0 and
    3.55, paid as 4); measured 898 k against 886 k expansions/s with the cap (the free-running batches hold ~3900 leaves)"""
    if channels % 256 or num_games < 1024 or precision == "bf16x3":
        return 0
    P, cols = (board_size - 2) ** 2, channels // 256
    rounds = (num_games * P // 256) * cols // 256
    if rounds < 1:
        return 0
    cap = (256 * rounds // cols) * 256 // P
    return 0 if cap >= num_games else int(cap)


def expand_examples(records, board_size, alias_final=False):
    """8-fold symmetry expansion of move records on the GPU (training.py:13-23,58-65).
    Returns boards (R*8, n, n, 2) uint8, policy_index (R*8,) int32 (position of the one-hot), z (R*8,) int8."""
    rec = np.ascontiguousarray(records, dtype=_lib.RECORD_DTYPE)
    R, n = rec.size, board_size
    boards = np.zeros((R * 8, n, n, 2), np.uint8)
    pol, z = np.zeros(R * 8, np.int32), np.zeros(R * 8, np.int8)
    if R:
        _lib.check(_lib.require_gpu().oz_examples_expand(rec.ctypes.data_as(C.c_void_p), R, n, 1 if alias_final else 0,
                                                         _lib.p_u8(boards), _lib.p_i32(pol), _lib.p_i8(z)))
    return boards, pol, z


def selfplay_batch(neural_network, board_size=8, num_games=4096, num_simulations=100, degree_exploration=1.0,
                   policy_temperature=1.0, e_greedy=0.9, seed=1234, first_game_id=0, q_mode=_lib.QMODE_F64,
                   expand=False, alias_final=False):
    """Play num_games complete games; returns the move records (or the expanded examples)."""
    eng = SelfPlayEngine(neural_network, board_size, num_games, num_simulations, degree_exploration, policy_temperature,
                         e_greedy, seed, first_game_id, q_mode=q_mode)
    rec = eng.play_to_end()
    return expand_examples(rec, board_size, alias_final) if expand else rec
