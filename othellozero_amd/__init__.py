"""othellozero_amd -- MI355X-native self-play engine behind the OthelloZero Python surface.

Hot path only: Othello rules, PUCT search, OthelloNN leaf evaluation and the
episode / arena drivers, as hand-written HIP kernels for gfx950 in `lib/libothellozero_amd.so`
(C ABI: include/othellozero_amd.h).  The modules mirror the reference's names:

    Othello      OthelloGame, OthelloPlayer, BoardView                (Othello/__init__.py)
    NNet         NNetWrapper, NeuralNets, StubNetWrapper              (Net/NNet.py)
    othelo_mcts  OthelloMCTS                                          (othelo_mcts.py, MCTS/__init__.py)
    training     execute_episode, training_example_symmetries,
                 SelfPlayEngine, selfplay_batch, expand_examples      (training.py)
    agents       NeuralNetworkOthelloAgent, RandomOthelloAgent,
                 duel_between_agents, arena_batch                     (agents.py)
    distributed  shard_games, gather_records (RCCL all-gather),
                 arena_sharded                                        (workers.py's role)

Importing the package loads nothing heavy; the first call that computes loads the HIP library and
fails loudly if it (or a GPU) is missing -- there is no CPU fallback.
"""
from . import _lib  # noqa: F401
from ._lib import QMODE_F64, QMODE_NEP50, OzError, OzLibraryError  # noqa: F401

__all__ = ["Othello", "NNet", "othelo_mcts", "training", "agents", "distributed", "QMODE_F64", "QMODE_NEP50"]
