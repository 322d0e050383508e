"""CPU-side checks of the training row: the oracle itself (oracle/train_ref.py against
independent torch.nn modules and a by-hand evaluation of the reference's row-normalised cross entropy), the host
helpers of othellozero_amd/trainer.py, and the data-parallel gradient averaging over gloo with world_size 2."""
import os
import socket
import subprocess
import sys

import numpy as np
import torch

from oracle.train_ref import TRAINABLE, TrainRef, dropout_keep, planes
from othellozero_amd.weights import init_weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _batch(n, B, seed):
    rs = np.random.RandomState(seed)
    valid = np.uint64(sum(1 << (r * 8 + c) for r in range(n) for c in range(n)))
    own = rs.randint(0, 2**63, size=B, dtype=np.uint64) & valid
    opp = rs.randint(0, 2**63, size=B, dtype=np.uint64) & valid & ~own
    pi = rs.dirichlet(np.ones(n * n), size=B)
    z = rs.choice([-1.0, 1.0], B)
    return own, opp, pi, z


def test_oracle_forward_matches_torch_modules_in_training_mode():
    n, C, B = 6, 8, 5
    w = init_weights(n, seed=1, channels=C, randomize_all=True)
    ref = TrainRef(w, n, dropout=0.0)
    own, opp, pi, z = _batch(n, B, 3)
    loss, lpi, lv = ref.forward_backward(own, opp, pi, z)
    # independent forward with torch.nn modules (double precision, train mode)
    x = torch.tensor(planes(own, opp, n)).permute(0, 3, 1, 2)
    h = x
    for layer, same in enumerate((True, True, False, False)):
        k, b, g, be, mu, var = (torch.tensor(np.asarray(a, np.float64)) for a in w[6 * layer:6 * layer + 6])
        h = torch.nn.functional.conv2d(h, k.permute(3, 2, 0, 1), b, padding=1 if same else 0)
        rm, rv = mu.clone(), var.clone()
        h = torch.nn.functional.batch_norm(h, rm, rv, g, be, training=True, momentum=0.01, eps=1e-3)
        # torch's running_var update is the unbiased one, like Keras' fused BN: the oracle's staged statistics must agree
        assert torch.allclose(ref._new_stats[6 * layer + 4], rm, atol=1e-12) and torch.allclose(ref._new_stats[6 * layer + 5], rv, atol=1e-12)
        h = torch.relu(h)
    f = h.permute(0, 2, 3, 1).reshape(B, -1)
    for blk in (24, 30):
        k, b, g, be, mu, var = (torch.tensor(np.asarray(a, np.float64)) for a in w[blk:blk + 6])
        zd = f @ k + b
        f = torch.relu(torch.nn.functional.batch_norm(zd, None, None, g, be, training=True, eps=1e-3))
        M = zd.shape[0]
        assert torch.allclose(ref._new_stats[blk + 5], var * 0.99 + zd.var(dim=0, unbiased=False) * 0.01, atol=1e-12)   # biased here
    p = torch.softmax(f @ torch.tensor(np.asarray(w[36], np.float64)) + torch.tensor(np.asarray(w[37], np.float64)), dim=1).numpy()
    v = torch.tanh(f @ torch.tensor(np.asarray(w[38], np.float64)) + torch.tensor(np.asarray(w[39], np.float64))).numpy()[:, 0]
    assert np.abs(p - ref.outputs["p"]).max() < 1e-12 and np.abs(v - ref.outputs["v"]).max() < 1e-12
    # the reference's loss, by hand: per board row renormalise, clip, -sum t log q; mean over batch x rows
    tot = 0.0
    for b in range(B):
        P, T = p[b].reshape(n, n), pi[b].reshape(n, n)
        for r in range(n):
            q = np.clip(P[r] / P[r].sum(), 1e-7, 1 - 1e-7)
            tot += -(T[r] * np.log(q)).sum()
    assert abs(tot / (B * n) - lpi) < 1e-12
    assert abs(np.mean((v - z) ** 2) - lv) < 1e-12 and abs(loss - (lpi + lv)) < 1e-12
    # bias gradients behind a training-mode BN vanish
    assert all(ref.grads[6 * l + 1].abs().max() < 1e-12 for l in range(6))


def test_oracle_adam_is_tf_keras_adam():
    n, C = 6, 8
    w = init_weights(n, seed=2, channels=C, randomize_all=True)
    ref = TrainRef(w, n, lr=1e-3, clipvalue=0.5, dropout=0.0)
    g = {i: torch.tensor(np.random.RandomState(i).standard_normal(np.shape(w[i]))) for i in TRAINABLE}
    ref._new_stats = {}
    m = {i: np.zeros(np.shape(w[i])) for i in TRAINABLE}
    v = {i: np.zeros(np.shape(w[i])) for i in TRAINABLE}
    want = {i: np.asarray(w[i], np.float64).copy() for i in TRAINABLE}
    for t in (1, 2, 3):
        ref.apply(grads=g)
        lr_t = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        for i in TRAINABLE:
            gc = np.clip(g[i].numpy(), -0.5, 0.5)
            m[i] = 0.9 * m[i] + 0.1 * gc
            v[i] = 0.999 * v[i] + 0.001 * gc * gc
            want[i] -= lr_t * m[i] / (np.sqrt(v[i]) + 1e-7)
    got = ref.weights()
    assert all(np.abs(got[i] - want[i]).max() < 1e-13 for i in TRAINABLE)
    assert all(np.array_equal(got[i], np.asarray(w[i], np.float64)) for i in range(40) if i not in TRAINABLE)


def test_dropout_mask_statistics_and_keying():
    k = dropout_keep(5, 0, 0, 200000, 0.3)
    assert abs(k.mean() - 0.7) < 0.005
    assert not np.array_equal(k, dropout_keep(5, 1, 0, 200000, 0.3)) and not np.array_equal(k, dropout_keep(5, 0, 1, 200000, 0.3))
    assert np.array_equal(k, dropout_keep(5, 0, 0, 200000, 0.3)) and dropout_keep(5, 0, 0, 100, 0.0).all()


def test_pack_examples_both_views():
    from othellozero_amd.trainer import pack_examples
    n = 6
    rs = np.random.RandomState(0)
    ex2, ex1 = [], []
    for _ in range(9):
        occ = rs.rand(n, n) < 0.5
        black = occ & (rs.rand(n, n) < 0.5)
        white = occ & ~black
        pol = rs.dirichlet(np.ones(n * n)).reshape(n, n)
        z = int(rs.choice([-1, 1]))
        ex2.append((np.stack([black, white], axis=2), pol, z))
        ex1.append((black.astype(int) - white.astype(int), pol, z))
    o2, p2, pi2, z2 = pack_examples(ex2, n, 2)
    o1, p1, pi1, z1 = pack_examples(ex1, n, 1)
    assert np.array_equal(o1, o2) and np.array_equal(p1, p2) and np.array_equal(pi1, pi2) and np.array_equal(z1, z2)
    x = planes(o2, p2, n)
    assert np.array_equal(x, np.asarray([e[0] for e in ex2], dtype=np.float64))
    assert pi2.shape == (9, n * n) and pi2.dtype == np.float32 and z2.dtype == np.float32


DP_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from othellozero_amd.distributed import GradientAllReduce, average_moving_statistics
from othellozero_amd.trainer import Trainer
from othellozero_amd.weights import init_weights
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)
n, C = 6, 128
ar = GradientAllReduce(n, C, 2, device="cpu")
assert ar.flat.numel() == Trainer.arena_size(n, C, 2) + 1 and ar.size % 4 == 0          # the arena + the step's status word
class FakeTrainer:
    synced = 0
    def sync(self): self.synced += 1
ft = FakeTrainer()
ar.flat[:ar.size].copy_(torch.arange(ar.size, dtype=torch.float32) % 97 + rank * 10)
ar(ft)
want = torch.arange(ar.size, dtype=torch.float32) % 97 + 5.0
assert ft.synced == 1 and torch.equal(ar.flat[:ar.size], want) and float(ar.flat[ar.size]) == 0.0, "arena not averaged"
w = init_weights(n, seed=rank, channels=C, randomize_all=True)
avg = average_moving_statistics(w)
w0, w1 = init_weights(n, seed=0, channels=C, randomize_all=True), init_weights(n, seed=1, channels=C, randomize_all=True)
for i in range(40):
    if i < 36 and i % 6 in (4, 5):
        assert np.allclose(avg[i], (w0[i] + w1[i]) / 2, atol=1e-7)
    else:
        assert np.array_equal(avg[i], w[i])
dist.barrier()
print("RANK_OK", rank)
'''


def test_gloo_world_size_2_gradient_averaging(tmp_path):
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {rank}" in out, out
