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
    def __init__(self, game):
        self.game = game

    def play(self):
        raise NotImplementedError


class RandomOthelloAgent(OthelloAgent):
    def play(self):                                   # agents.py:20-24
        possible_moves = tuple(self.game.get_valid_actions())
        move = random.choice(possible_moves)
        self.game.play(*move)


class NeuralNetworkOthelloAgent(OthelloAgent):
    def __init__(self, game, neural_network, num_simulations, degree_exploration, temperature=0,
                 q_mode=_lib.QMODE_F64):
        self.temperature = 0                          # the argument is ignored, as in agents.py:46
        self.neural_network = neural_network
        self.num_simulations = num_simulations
        n = game.board_size
        self.mcts = OthelloMCTS(n, neural_network, degree_exploration, q_mode=q_mode,
                                node_cap=num_simulations * (n * n // 2) + 64)
        super().__init__(game)

    def play(self):                                   # agents.py:52-68
        state = self.game.board(BoardView.TWO_CHANNELS)
        self.mcts.simulate_n(state, self.game.current_player, self.num_simulations)
        if self.game.current_player == OthelloPlayer.WHITE:
            state = OthelloGame.invert_board(state)
        action_probabilities = self.mcts.get_policy_action_probabilities(state, self.temperature)
        valid_actions = self.game.get_valid_actions()
        best_action = max(valid_actions, key=lambda position: action_probabilities[tuple(position)])
        self.game.play(*best_action)


def duel_between_agents(game, agent_1, agent_2):
    """agents.py:71-84 -> (winning agent, points)"""
    players_agents = {OthelloPlayer.BLACK: agent_1, OthelloPlayer.WHITE: agent_2}
    logging.info('Duel - Started')
    while not game.has_finished():
        logging.info(f'Duel - Round: {game.round}')
        players_agents[game.current_player].play()
    winner, points = game.get_winning_player()
    return players_agents[winner], points


def arena_batch(net_a, net_b, board_size=8, num_games=512, num_simulations=800, degree_exploration=1.0, seed=0,
                first_game_id=0, q_mode=_lib.QMODE_F64, node_cap=0, edge_cap=0):
    """num_games games of net_a (BLACK) vs net_b (WHITE), temperature 0, max-visit ties broken by the RNG_TIE
    stream keyed (seed, game id, ply).  One of the two may be None: RandomOthelloAgent plays that colour.
    Returns dict(winner (+1 = BLACK's agent), points, n_moves, actions, players, final)."""
    lib = _lib.require_gpu()
    h = C.c_void_p()
    _lib.check(lib.oz_arena_create(C.byref(h), board_size, num_games, num_simulations, float(degree_exploration), q_mode,
                                   seed, first_game_id, net_a._h if net_a is not None else None,
                                   net_b._h if net_b is not None else None, node_cap, edge_cap))
    try:
        _lib.check(lib.oz_arena_run(h))
        G = num_games
        winner, points, nm = np.zeros(G, np.int8), np.zeros(G, np.int32), np.zeros(G, np.int32)
        acts, pls = np.zeros((G, 128), np.uint8), np.zeros((G, 128), np.int8)
        fb, fw = np.zeros(G, np.uint64), np.zeros(G, np.uint64)
        _lib.check(lib.oz_arena_results(h, _lib.p_i8(winner), _lib.p_i32(points), _lib.p_i32(nm), _lib.p_u8(acts),
                                        _lib.p_i8(pls), _lib.p_u64(fb), _lib.p_u64(fw)))
    finally:
        lib.oz_arena_destroy(h)
    return dict(winner=winner, points=points, n_moves=nm, actions=acts, players=pls, final_black=fb, final_white=fw)
