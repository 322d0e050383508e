"""Agents and arena -- drop-in for agents.py:9-84 plus the batched arena (config 5).

`NeuralNetworkOthelloAgent` / `RandomOthelloAgent` / `duel_between_agents` keep the reference's
behaviour (one OthelloMCTS per agent, temperature forced to 0, BLACK = agent_1, a draw goes to BLACK).
`arena_batch` plays many deterministic best-vs-candidate games in lock step on the GPU.
The reference's GreedyOthelloAgent is dead code (undefined names, agents.py:27-41) and is not reproduced.
"""
import ctypes as C
import logging
import random

import numpy as np

from . import _lib
from .Othello import BoardView, OthelloGame, OthelloPlayer
from .othelo_mcts import OthelloMCTS


class OthelloAgent:
    """agents.py:9-17: an agent is bound to ONE game object and moves on it when asked"""

    def __init__(self, game):
        self.game = game

    def play(self):
        raise NotImplementedError


class RandomOthelloAgent(OthelloAgent):
    def play(self):
        """agents.py:20-24: one `random.choice` over the valid actions in row-major order"""
        moves = tuple(self.game.get_valid_actions())
        self.game.play(*random.choice(moves))


class NeuralNetworkOthelloAgent(OthelloAgent):
    """agents.py:44-68: a private OthelloMCTS per agent (its table persists over the game), `num_simulations`
    simulations before every move, the temperature argument ignored and forced to 0 (agents.py:46), the move =
    the first valid action with the largest policy entry."""

    def __init__(self, game, neural_network, num_simulations, degree_exploration, temperature=0,
                 q_mode=_lib.QMODE_F64):
        super().__init__(game)
        self.neural_network, self.num_simulations, self.temperature = neural_network, num_simulations, 0
        side = game.board_size
        # an agent searches on its own turns only: about half the plies
        self.mcts = OthelloMCTS(side, neural_network, degree_exploration, q_mode=q_mode,
                                node_cap=num_simulations * (side * side // 2) + 64)

    def play(self):
        game = self.game
        mover = game.current_player
        board = game.board(BoardView.TWO_CHANNELS)
        self.mcts.simulate_n(board, mover, self.num_simulations)
        canonical = board if mover == OthelloPlayer.BLACK else OthelloGame.invert_board(board)
        policy = self.mcts.get_policy_action_probabilities(canonical, self.temperature)
        best = None
        for action in game.get_valid_actions():               # max() keeps the FIRST maximum: strict '>' below
            if best is None or policy[tuple(action)] > policy[tuple(best)]:
                best = action
        game.play(*best)


def duel_between_agents(game, agent_1, agent_2):
    """agents.py:71-84: agent_1 moves for BLACK, agent_2 for WHITE, until the game is over.
    -> (winning agent, its points); a draw goes to BLACK's agent (get_winning_player)."""
    by_colour = {OthelloPlayer.BLACK: agent_1, OthelloPlayer.WHITE: agent_2}
    logging.info('Duel - Started')
    while not game.has_finished():
        logging.info(f'Duel - Round: {game.round}')
        by_colour[game.current_player].play()
    colour, points = game.get_winning_player()
    return by_colour[colour], points


def arena_batch(net_a, net_b, board_size=8, num_games=512, num_simulations=800, degree_exploration=1.0, seed=0,
                first_game_id=0, q_mode=_lib.QMODE_F64, node_cap=0, max_rounds=0, dedup=True, profile=False, eval_cache=False):
    """num_games games of net_a (BLACK) vs net_b (WHITE), temperature 0, max-visit ties broken by the RNG_TIE
    stream keyed (seed, game id, ply).  One of the two may be None: RandomOthelloAgent plays that colour.
    max_rounds > 0 stops after that many plies per game (unfinished boards: winner / points then describe the position reached).
    Returns dict(winner (+1 = BLACK's agent), points, n_moves, actions, players, final boards, stats_black / stats_white =
    the two agents' search counters [simulations, node visits, expansions, terminal hits, fallbacks], leaves_evaluated = positions the
    networks evaluated: fewer than the expansions with dedup=True (the default), where a board several games reach in one step is evaluated once;
    tree_kernels = {slot: (ms, launches)} of both searches' tree kernels with profile=True, else None)."""
    lib = _lib.require_gpu()
    h = C.c_void_p()
    _lib.check(lib.oz_arena_create(C.byref(h), board_size, num_games, num_simulations, float(degree_exploration), q_mode,
                                   seed, first_game_id, net_a._h if net_a is not None else None,
                                   net_b._h if net_b is not None else None, node_cap))
    try:
        if not dedup:                                        # every expansion evaluated by itself (identical results; bench.py's config5 headline)
            _lib.check(lib.oz_arena_set_dedup(h, 0))
        if eval_cache:                                       # leaves looked up in / inserted into the two networks' evaluation caches (net.set_eval_cache first)
            _lib.check(lib.oz_arena_set_eval_cache(h, 1))
        if profile:                                          # HIP events around the tree kernels of both searches (bench.py's config5 kernels[])
            _lib.check(lib.oz_arena_profile(h, 1))
        _lib.check(lib.oz_arena_run_rounds(h, int(max_rounds)))
        tree = None
        if profile:
            ms, cnt = np.zeros(len(_lib.TREE_KERNELS), np.float64), np.zeros(len(_lib.TREE_KERNELS), np.int64)
            _lib.check(lib.oz_arena_profile_read(h, _lib.p_f64(ms), _lib.p_i64(cnt), 0))
            tree = {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(_lib.TREE_KERNELS)}
        G = num_games
        sa, sb = np.zeros(5, np.int64), np.zeros(5, np.int64)
        _lib.check(lib.oz_arena_stats(h, _lib.p_i64(sa), _lib.p_i64(sb)))
        ea, eb = C.c_int64(), C.c_int64()
        _lib.check(lib.oz_arena_leaves_evaluated(h, C.byref(ea), C.byref(eb)))
        winner, points, nm = np.zeros(G, np.int8), np.zeros(G, np.int32), np.zeros(G, np.int32)
        acts, pls = np.zeros((G, 128), np.uint8), np.zeros((G, 128), np.int8)
        fb, fw = np.zeros(G, np.uint64), np.zeros(G, np.uint64)
        _lib.check(lib.oz_arena_results(h, _lib.p_i8(winner), _lib.p_i32(points), _lib.p_i32(nm), _lib.p_u8(acts),
                                        _lib.p_i8(pls), _lib.p_u64(fb), _lib.p_u64(fw)))
    finally:
        lib.oz_arena_destroy(h)
    return dict(winner=winner, points=points, n_moves=nm, actions=acts, players=pls, final_black=fb, final_white=fw,
                stats_black=sa, stats_white=sb, leaves_evaluated=ea.value + eb.value, tree_kernels=tree)
