"""Replay buffer, iteration loop and promotion rules -- the caller of the hot path (main.py:21-259), SURVEY.md 8(f) item 3.

`CircularArray` is a drop-in for main.py:21-53 (same ring arithmetic, including its quirks: the write index only starts
moving once the buffer is full, and `random.shuffle(buffer)` permutes the slots the ring later overwrites).
`training(...)` keeps main.py:56-259's structure and decision rules, with the reference's WorkerManager fan-out replaced by
the batched GPU engines:

  episodes            worker_manager.run(EXECUTE_EPISODE, ...)          -> training.selfplay_batch (lock-step games in HBM)
  new-vs-old matches  worker_manager.run(DUEL_BETWEEN_NEURAL_NETWORKS)  -> agents.arena_batch (half the games per colour)
  fit                 neural_network.train(examples)                    -> oz_trainer_* (NNet.py)
  evaluation          duel_between_agents vs RandomOthelloAgent         -> the same drop-in agents (agents.py), sequential

Decision rules kept verbatim: promote after self-play when `new_net_victories >= self_play_threshold` (main.py:138);
after an evaluation round keep the new network when `net_wins > old_net_wins * 1.1` (main.py:238), otherwise fall back to
the old one; temperature drops to 0 from iteration `temperature_threshold` on (main.py:74-77).

Deliberate differences (documented, switchable where it matters):
* `reference_aliasing=True` reproduces `old_neural_network = neural_network` (main.py:142,243): after the first promotion
  both names are ONE object, so later training also changes the "old" network.  False keeps a real copy.
* the reference's worker-side `duel_between_neural_networks` (training.py:75-88) cannot run (it indexes a dict with the
  tuple `duel_between_agents` returns); the intent -- `self_play_total_games` temperature-0 games, half per colour -- is what
  `arena_batch` plays.
* `examples-<n>.txt` (a str() dump of the whole buffer every iteration, main.py:252-253) is only written when asked.
"""
import logging
import random

import numpy as np

from . import _lib
from .agents import NeuralNetworkOthelloAgent, RandomOthelloAgent, arena_batch, duel_between_agents
from .Othello import OthelloGame, OthelloPlayer
from .training import expand_examples, selfplay_batch


class CircularArray:
    """Ring buffer with the semantics of main.py:21-53, quirks included: it grows like a list up to `max_`; once full,
    item k overwrites slot `_index % len` and `_index` becomes that slot + 1 (so it runs 1..len and restarts at slot 0
    through the modulo).  Indexing, slicing, assignment, iteration, len, str and repr go to the underlying list."""

    def __init__(self, max_):
        self._list, self._max, self._index = [], max_, 0

    def append(self, item):
        held = len(self._list)
        if held < self._max:
            self._list.append(item)
            return
        slot = self._index % held
        self._list[slot] = item
        self._index = slot + 1

    def extend(self, items):
        for item in items:
            self.append(item)

    def __len__(self):
        return len(self._list)

    def __getitem__(self, key):
        return self._list[key]

    def __setitem__(self, key, value):
        self._list[key] = value

    def __iter__(self):
        return iter(self._list)

    def __str__(self):
        return str(self._list)

    def __repr__(self):
        return f'{type(self).__name__}({len(self._list)!r})'


def examples_from_records(records, board_size, alias_final=True, in_channels=2):
    """move records of finished games -> the reference's example tuples [(board, one-hot policy (n,n), z)], 8 per move in
    training.py:13-23's order.

    in_channels=2 (OthelloNN): board (n,n,2) bool; alias_final=True shows every board as the game's FINAL position, which is
    what execute_episode really returns (the examples are views of the live game array); False stores the
    position at the move.
    in_channels=1 (BaseNN): board (n,n) with +1 BLACK / -1 WHITE, the position AT THE MOVE whatever alias_final says -- the
    reference builds a fresh one-channel array per round for BNN (training.py:34-37, Othello/__init__.py:79-84,266-270), so
    those examples are never aliased."""
    one_channel = in_channels == 1
    boards, pol, z = expand_examples(records, board_size, alias_final=alias_final and not one_channel)
    n = board_size
    if one_channel:
        boards = boards[..., 0].astype(np.int64) - boards[..., 1].astype(np.int64)      # convert_to_one_channel_board
    else:
        boards = boards.astype(bool)
    out = []
    for b, p, zz in zip(boards, pol, z):
        policy = np.zeros((n, n))
        policy[p // n, p % n] = 1
        out.append((b, policy, int(zz)))
    return out


def evaluate_against_random(board_size, neural_network, games, num_simulations, degree_exploration, label="Network"):
    """main.py:163-192 / :197-233: `games` duels against RandomOthelloAgent, colours drawn by random.shuffle.
    -> dict(wins, black_wins, black_games, white_wins, white_games)"""
    r = dict(wins=0, black_wins=0, black_games=0, white_wins=0, white_games=0)
    if getattr(neural_network, "max_batch", 1) > 32 and hasattr(neural_network, "get_weights"):
        # one position per call: a twin with max_batch 1 takes the library's latency path (k loops split over idle CUs)
        from .NNet import NNetWrapper
        neural_network = NNetWrapper((board_size, board_size), network=neural_network.network_type,
                                     num_channels_1=neural_network.num_channels, max_batch=1,
                                     weights=neural_network.get_weights(), precision=neural_network.precision)
    for k in range(games):
        game = OthelloGame(board_size, current_player=OthelloPlayer.BLACK)
        nn_agent = NeuralNetworkOthelloAgent(game, neural_network, num_simulations, degree_exploration)
        random_agent = RandomOthelloAgent(game)
        agents = [nn_agent, random_agent]
        random.shuffle(agents)
        agent_winner, points = duel_between_agents(game, *agents)
        winner = OthelloPlayer.BLACK if agents[0] is agent_winner else OthelloPlayer.WHITE
        colour = "black" if winner == OthelloPlayer.BLACK else "white"
        r[colour + "_games"] += 1
        if agent_winner is nn_agent:
            r["wins"] += 1
            r[colour + "_wins"] += 1
        logging.info('%s: %d of %d won (%.2f); as black %d/%d, as white %d/%d', label, r["wins"], k + 1, r["wins"] / (k + 1),
                     r["black_wins"], r["black_games"], r["white_wins"], r["white_games"])
    return r


def evaluate_against_random_batch(board_size, neural_network, games, num_simulations, degree_exploration, seed=0):
    """The same evaluation as `evaluate_against_random` played in lock step on the GPU (agents.arena_batch with a random
    mover): the network takes BLACK in the first games // 2 + games % 2 games and WHITE in the rest (the reference draws the
    colours with random.shuffle; the split here is fixed).  -> dict(wins, black_wins, black_games, white_wins, white_games)"""
    as_black, as_white = games // 2 + games % 2, games // 2
    r = dict(wins=0, black_wins=0, black_games=0, white_wins=0, white_games=0)
    if as_black:
        res = arena_batch(neural_network, None, board_size, as_black, num_simulations, degree_exploration, seed=seed)
        r["black_wins"] = int((res["winner"] == 1).sum())
    if as_white:
        res = arena_batch(None, neural_network, board_size, as_white, num_simulations, degree_exploration, seed=seed, first_game_id=as_black)
        r["white_wins"] = int((res["winner"] == -1).sum())
        r["black_games"] = as_white - r["white_wins"]              # games BLACK (the random agent) won
    r["white_games"] = r["white_wins"] + (as_black - r["black_wins"])
    r["black_games"] += r["black_wins"]
    r["wins"] = r["black_wins"] + r["white_wins"]
    return r


def self_play_match(board_size, neural_network, old_neural_network, total_games, num_simulations, degree_exploration, seed=0):
    """main.py:110-134: total_games // 2 games with the new network as BLACK, the rest with it as WHITE.
    -> number of games the new network won (a drawn game goes to BLACK, like get_winning_player)."""
    as_black, as_white = total_games // 2, total_games // 2 + total_games % 2
    wins = 0
    if as_black:
        res = arena_batch(neural_network, old_neural_network, board_size, as_black, num_simulations, degree_exploration, seed=seed)
        wins += int((res["winner"] == 1).sum())
    if as_white:
        res = arena_batch(old_neural_network, neural_network, board_size, as_white, num_simulations, degree_exploration,
                          seed=seed, first_game_id=as_black)
        wins += int((res["winner"] == -1).sum())
    return wins


def training(board_size, num_iterations, num_episodes, num_simulations, degree_exploration, temperature, neural_network,
             e_greedy, evaluation_interval, evaluation_iterations, temperature_threshold, self_play_training,
             self_play_interval, self_play_total_games, self_play_threshold, checkpoint_filepath, training_buffer_size,
             seed=1234, reference_aliasing=True, alias_final_boards=True, dump_examples=False, q_mode=_lib.QMODE_F64,
             distributed=False, batched_evaluation=False):
    """main.py:56-259 on the GPU engines; returns `historic` = [(episodes done, win rate vs random), ...]

    batched_evaluation=True plays the evaluation games against RandomOthelloAgent in lock step on the GPU
    (evaluate_against_random_batch) instead of one by one through the drop-in agents.

    distributed=True (torch.distributed initialised, one process per GPU): the episodes of an iteration are sharded over
    the ranks by global game id and pooled with one all-gather of move records, every rank then holds the same replay
    buffer (same `random` stream, seeded here), trains on its 1/world slice of it with one gradient all-reduce per step,
    and plays the (deterministic) matches / evaluations redundantly, so all ranks take the same promotion decisions
    without further communication; rank 0 writes the files."""
    if self_play_training:
        assert self_play_threshold <= self_play_total_games, 'Self-play threshold must be less than self-play games'

    world, rank, allreduce = 1, 0, None
    if distributed:
        import torch
        import torch.distributed as dist
        from .distributed import GradientAllReduce, pooled_selfplay_records, shard_games
        from .training import SelfPlayEngine
        world, rank = dist.get_world_size(), dist.get_rank()
        assert num_episodes >= world, "fewer episodes than ranks"
        device = torch.device("cuda", torch.cuda.current_device())
        # pin the library to the device torch selected for this rank (objects created on this thread from here on)
        _lib.check(_lib.load().oz_set_device(torch.cuda.current_device()))
        allreduce = GradientAllReduce(board_size, neural_network.num_channels, neural_network.in_channels, device=device)
        random.seed(seed)

    def save(net):
        if rank == 0:
            net.save_checkpoint(checkpoint_filepath)

    historic = []
    total_episodes_done = 0
    training_examples = CircularArray(training_buffer_size)
    old_neural_network = neural_network.copy()
    for i in range(1, num_iterations + 1):
        logging.info('[%d/%d] begin', i, num_iterations)
        if temperature_threshold and i >= temperature_threshold:
            logging.info('[%d/%d] past the temperature threshold: moves are now the arg-max of the visit counts', i, num_iterations)
            temperature = 0

        logging.info('[%d/%d] self-play: %d games x %d simulations on the GPU', i, num_iterations, num_episodes, num_simulations)
        if distributed:
            first, count = shard_games(num_episodes, rank, world)
            eng = SelfPlayEngine(neural_network, board_size, count, num_simulations, degree_exploration, temperature, e_greedy,
                                 seed=seed, first_game_id=total_episodes_done + first, q_mode=q_mode)
            eng.play_to_end()
            records = pooled_selfplay_records(eng, device)          # the only exchange of the self-play phase
            del eng
        else:
            records = selfplay_batch(neural_network, board_size, num_games=num_episodes, num_simulations=num_simulations,
                                     degree_exploration=degree_exploration, policy_temperature=temperature, e_greedy=e_greedy,
                                     seed=seed, first_game_id=total_episodes_done, q_mode=q_mode)
        training_examples.extend(examples_from_records(records, board_size, alias_final=alias_final_boards,
                                                       in_channels=getattr(neural_network, "in_channels", 2)))
        total_episodes_done += num_episodes
        logging.info('[%d/%d] self-play done: %d records, buffer holds %d examples', i, num_iterations, len(records), len(training_examples))

        logging.info('[%d/%d] fit on the buffer', i, num_iterations)
        random.shuffle(training_examples)
        verbose = 2 if logging.root.level <= logging.DEBUG else None
        if distributed:
            usable = len(training_examples) - len(training_examples) % world     # equal step counts on every rank
            neural_network.train([training_examples[j] for j in range(rank, usable, world)], verbose=verbose, allreduce=allreduce)
        else:
            neural_network.train(training_examples, verbose=verbose)

        if self_play_training and i % self_play_interval == 0:
            logging.info('[%d/%d] arena: trained network against the previous one', i, num_iterations)
            new_net_victories = self_play_match(board_size, neural_network, old_neural_network, self_play_total_games,
                                                num_simulations, degree_exploration, seed=seed + i)
            logging.info('[%d/%d] arena: %d of %d games to the trained network', i, num_iterations, new_net_victories, self_play_total_games)
            if new_net_victories >= self_play_threshold:
                logging.info('[%d/%d] trained network promoted', i, num_iterations)
                save(neural_network)
                logging.info('[%d/%d] checkpoint -> %s', i, num_iterations, checkpoint_filepath)
                old_neural_network = neural_network if reference_aliasing else neural_network.copy()
            else:
                neural_network = old_neural_network if reference_aliasing else old_neural_network.copy()
                logging.info('[%d/%d] trained network rejected, previous one kept', i, num_iterations)
        else:
            save(neural_network)

        if i % evaluation_interval == 0:
            logging.info('[%d/%d] evaluation against the random agent: current network', i, num_iterations)
            if batched_evaluation:
                new = evaluate_against_random_batch(board_size, neural_network, evaluation_iterations, num_simulations,
                                                    degree_exploration, seed=seed + 7919 * i)
                old = evaluate_against_random_batch(board_size, old_neural_network, evaluation_iterations, num_simulations,
                                                    degree_exploration, seed=seed + 7919 * i + 1)
            else:
                new = evaluate_against_random(board_size, neural_network, evaluation_iterations, num_simulations, degree_exploration,
                                              label=f'after {total_episodes_done} episodes, current network')
                logging.info('[%d/%d] evaluation against the random agent: previous network', i, num_iterations)
                old = evaluate_against_random(board_size, old_neural_network, evaluation_iterations, num_simulations, degree_exploration,
                                              label=f'after {total_episodes_done} episodes, previous network')
            if new["wins"] > (old["wins"] * 1.1):
                logging.info('[%d/%d] evaluation: current network kept (%d wins vs %d)', i, num_iterations, new['wins'], old['wins'])
                historic.append((total_episodes_done, (new["wins"] / evaluation_iterations)))
                save(neural_network)
                old_neural_network = neural_network if reference_aliasing else neural_network.copy()
            else:
                logging.info('[%d/%d] evaluation: back to the previous network (%d wins vs %d)', i, num_iterations, new['wins'], old['wins'])
                historic.append((total_episodes_done, (old["wins"] / evaluation_iterations)))
                save(old_neural_network)
                neural_network = old_neural_network if reference_aliasing else old_neural_network.copy()
            logging.info('history (episodes, win rate): %s', historic)

        logging.info('[%d/%d] end: %d episodes so far', i, num_iterations, total_episodes_done)
        if dump_examples and rank == 0:
            with open(f'examples-{board_size}.txt', 'w') as output:
                output.write(str(training_examples))
        if rank == 0:
            with open(f'historic-last-training-session-{board_size}.txt', 'w') as output:
                output.write(str(historic))

    training.last_network = neural_network
    return historic
