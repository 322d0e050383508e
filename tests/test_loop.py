"""Replay buffer + iteration loop (reference main.py:21-259)."""
import json
import os
import random
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_circular_array_matches_reference_trace():
    """every state of the scripted operation sequence equals what the reference's own class produced
    (tests/golden/circular_array.json, generator gen_loop_golden.py)"""
    import gen_loop_golden as G
    from othellozero_amd.loop import CircularArray
    want = json.load(open(os.path.join(HERE, "golden", "circular_array.json")))
    got = G.script(CircularArray)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert json.loads(json.dumps(g)) == w, (g, w)


def test_circular_array_quirks_spelled_out():
    from othellozero_amd.loop import CircularArray
    ca = CircularArray(3)
    ca.extend([0, 1, 2])
    assert list(ca) == [0, 1, 2] and ca._index == 0
    ca.append(3)                                    # full: slot 0 is overwritten, the index becomes 1
    ca.append(4)
    assert list(ca) == [3, 4, 2] and ca._index == 2
    ca.extend([5, 6])                               # index wraps as (index % len) + 1, so it runs 1..len, never 0 again
    assert list(ca) == [6, 4, 5] and ca._index == 1 and len(ca) == 3 and repr(ca) == "CircularArray(3)"


@pytest.mark.gpu
def test_examples_from_records_match_execute_episode_layout():
    """records -> example tuples: same count, order and aliasing as training.execute_episode's return value"""
    from othellozero_amd.loop import examples_from_records
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.training import selfplay_batch
    n = 6
    net = StubNetWrapper((n, n), salt=3, max_batch=4)
    rec = selfplay_batch(net, n, num_games=4, num_simulations=6, seed=5)
    ex = examples_from_records(rec, n, alias_final=True)
    assert len(ex) == 8 * len(rec)
    b, p, z = ex[0]
    assert b.shape == (n, n, 2) and b.dtype == bool and p.shape == (n, n) and p.sum() == 1 and z in (-1, 1)
    g0 = rec[rec["game_id"] == rec["game_id"][0]]
    final = ex[8 * (len(g0) - 1) + 7][0]            # last move of game 0, 8th symmetry = identity rotation (k=4, no flip)
    first = ex[7][0]
    assert np.array_equal(first, final)             # aliasing quirk: every example of a game shows the final board
    ex2 = examples_from_records(rec, n, alias_final=False)
    assert ex2[7][0].sum() == 4 + 0                 # the real first position: 4 discs


def test_pack_examples_accepts_one_channel_boards():
    """trainer.pack_examples with in_channels=1 takes (n,n) +1 / -1 boards (what examples_from_records emits for BNN) --
    two-channel boards handed to it by mistake fail loudly instead of being silently reshaped"""
    from othellozero_amd.trainer import pack_examples
    n = 6
    rs = np.random.RandomState(2)
    boards = rs.randint(-1, 2, size=(5, n, n)).astype(np.int64)
    ex = [(b, np.eye(n * n)[i].reshape(n, n), 1 if i % 2 else -1) for i, b in enumerate(boards)]
    own, opp, pi, z = pack_examples(ex, n, in_channels=1)
    for i, b in enumerate(boards):
        o = sum(1 << (r * 8 + c) for r in range(n) for c in range(n) if b[r, c] == 1)
        p = sum(1 << (r * 8 + c) for r in range(n) for c in range(n) if b[r, c] == -1)
        assert (int(own[i]), int(opp[i])) == (o, p)
    assert pi.shape == (5, n * n) and np.array_equal(pi.argmax(axis=1), np.arange(5)) and list(z) == [-1, 1, -1, 1, -1]
    two = [(np.stack([b == 1, b == -1], axis=2), e[1], e[2]) for b, e in zip(boards, ex)]
    with pytest.raises(ValueError):
        pack_examples(two, n, in_channels=1)


@pytest.mark.gpu
def test_examples_from_records_one_channel_for_basenn():
    """BNN examples are (n,n) +1 BLACK / -1 WHITE boards of the position AT THE MOVE (training.py:34-37: a fresh one-channel
    array per round, never aliased to the final board), in the same 8-symmetry order as the two-channel ones"""
    from othellozero_amd.loop import examples_from_records
    from othellozero_amd.NNet import StubNetWrapper
    from othellozero_amd.training import selfplay_batch
    n = 6
    rec = selfplay_batch(StubNetWrapper((n, n), salt=3, max_batch=4), n, num_games=4, num_simulations=6, seed=5)
    ex1 = examples_from_records(rec, n, alias_final=True, in_channels=1)
    ex2 = examples_from_records(rec, n, alias_final=False, in_channels=2)
    assert len(ex1) == len(ex2) == 8 * len(rec)
    for (b1, p1, z1), (b2, p2, z2) in zip(ex1, ex2):
        assert b1.shape == (n, n) and np.array_equal(b1, b2[:, :, 0].astype(np.int64) - b2[:, :, 1].astype(np.int64))
        assert np.array_equal(p1, p2) and z1 == z2
    assert np.abs(ex1[7][0]).sum() == 4             # first position of game 0: four discs, not the final board


@pytest.mark.gpu
def test_training_loop_basenn_iteration(tmp_path, monkeypatch):
    """one iteration of the loop with NNetWrapper(network=BNN): one-channel examples reach NNetWrapper.train (ADVICE r1:
    this used to raise ValueError in pack_examples), the network changes and stays finite"""
    from othellozero_amd.loop import training
    from othellozero_amd.NNet import NNetWrapper, NeuralNets
    monkeypatch.chdir(tmp_path)
    random.seed(4)
    np.random.seed(4)
    n = 6
    net = NNetWrapper((n, n), num_channels_1=128, batch_size=32, epochs=1, max_batch=8, network=NeuralNets.BNN)
    w0 = net.get_weights()
    historic = training(board_size=n, num_iterations=1, num_episodes=6, num_simulations=6, degree_exploration=1, temperature=1,
                        neural_network=net, e_greedy=0.9, evaluation_interval=1, evaluation_iterations=2, temperature_threshold=0,
                        self_play_training=False, self_play_interval=1, self_play_total_games=2, self_play_threshold=1,
                        checkpoint_filepath=str(tmp_path / "bnn.h5"), training_buffer_size=8 * 40, seed=12, batched_evaluation=True)
    assert len(historic) == 1
    w1 = net.get_weights()                            # `net` is trained in place whatever the promotion rules then decide
    assert w1[0].shape == (3, 3, 1, 128) and all(np.isfinite(a).all() for a in w1)
    assert any(not np.array_equal(a, b) for a, b in zip(w0, w1))          # the optimiser steps happened


@pytest.mark.gpu
def test_training_loop_end_to_end(tmp_path, monkeypatch):
    """two iterations of main.training()'s structure on the GPU engines: episodes -> ring buffer -> shuffle -> fit ->
    new-vs-old matches -> promotion rule -> evaluation vs random -> 1.1x rule -> checkpoint"""
    from othellozero_amd import keras_h5
    from othellozero_amd.loop import training
    from othellozero_amd.NNet import NNetWrapper
    monkeypatch.chdir(tmp_path)
    random.seed(3)
    np.random.seed(3)
    n = 6
    net = NNetWrapper((n, n), num_channels_1=128, batch_size=32, epochs=1, max_batch=8)
    w0 = net.get_weights()
    ckpt = str(tmp_path / "othelo_model_weights.h5")
    historic = training(board_size=n, num_iterations=2, num_episodes=8, num_simulations=8, degree_exploration=1, temperature=1,
                        neural_network=net, e_greedy=0.9, evaluation_interval=1, evaluation_iterations=2, temperature_threshold=2,
                        self_play_training=True, self_play_interval=1, self_play_total_games=3, self_play_threshold=2,
                        checkpoint_filepath=ckpt, training_buffer_size=8 * 40, seed=11, batched_evaluation=True)
    assert len(historic) == 2 and all(0 <= rate <= 1 for _, rate in historic) and [e for e, _ in historic] == [8, 16]
    assert os.path.exists(ckpt) and len(keras_h5.flat_weights(keras_h5.load_keras_weights(ckpt))) == 40
    assert os.path.exists(tmp_path / f"historic-last-training-session-{n}.txt")
    final = training.last_network
    probe = NNetWrapper((n, n), num_channels_1=128, max_batch=1)
    probe.load_checkpoint(ckpt)                      # the saved file is a loadable network of the right shape
    assert all(np.isfinite(a).all() for a in probe.get_weights())
    # (the final network equals w0 when neither promotion rule fired: both outcomes are legal)
    assert len(final.get_weights()) == len(w0) and all(np.isfinite(a).all() for a in final.get_weights())


LOOP_WORKER = r'''
import os, sys, random, json
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from othellozero_amd.loop import training
from othellozero_amd.NNet import NNetWrapper
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)          # control-flow rehearsal: both ranks share GPU 0
torch.cuda.set_device(0)
os.chdir(sys.argv[2])
n = 6
net = NNetWrapper((n, n), num_channels_1=128, batch_size=16, epochs=1, max_batch=8, seed=0)
hist = training(board_size=n, num_iterations=2, num_episodes=6, num_simulations=6, degree_exploration=1, temperature=1,
                neural_network=net, e_greedy=0.9, evaluation_interval=1, evaluation_iterations=2, temperature_threshold=0,
                self_play_training=True, self_play_interval=1, self_play_total_games=2, self_play_threshold=1,
                checkpoint_filepath=os.path.join(sys.argv[2], "dp.h5"), training_buffer_size=8 * 64, seed=21, distributed=True)
final = training.last_network.get_weights()
flat = torch.from_numpy(np.concatenate([a.ravel() for a in final]))
both = [torch.zeros_like(flat) for _ in range(2)]
dist.all_gather(both, flat)
assert torch.equal(both[0], both[1]), "the ranks ended with different networks"
hs = [None, None]
dist.all_gather_object(hs, hist)
assert hs[0] == hs[1] and len(hist) == 2, hs
dist.barrier()
if rank == 0:
    assert os.path.exists(os.path.join(sys.argv[2], "dp.h5"))
print("RANK_OK", rank)
'''


@pytest.mark.gpu
def test_distributed_loop_two_ranks(tmp_path):
    """training(..., distributed=True) on two ranks (gloo, both on GPU 0): sharded episodes pooled by all-gather, data-parallel
    fit, redundant deterministic matches -- both ranks finish with the same network and the same history"""
    import socket
    import subprocess
    root = os.path.dirname(HERE)
    script = tmp_path / "loop_worker.py"
    script.write_text(LOOP_WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), root, str(tmp_path)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=400)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {rank}" in out, out
