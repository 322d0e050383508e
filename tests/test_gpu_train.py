"""Training step parity (reference Net/NNet.py:53-68): the HIP forward / backward / Adam kernels,
called through the C ABI, against oracle/train_ref.py (float64 torch autograd restatement of the Keras arithmetic).

Tolerances (floating point, fp32 kernels vs float64 oracle): outputs and losses 2e-5 absolute; gradients
max|d| <= 3e-4 * max|g_ref| + 1e-7 per tensor; the Adam update and the BN moving statistics 2e-6 absolute per step when both
sides are fed the same gradients (step-locked test).  The biases that sit behind a training-mode BatchNormalization have a
true gradient of exactly 0: what either side computes is rounding noise (asserted <= 1e-5), and Adam turns noise of any
scale into steps of up to lr -- which is why free-running comparisons only bound those (they cannot influence any output).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BN_BIASES = [6 * l + 1 for l in range(6)]


def _batch(n, B, seed, cin=2):
    rs = np.random.RandomState(seed)
    valid = np.uint64(sum(1 << (r * 8 + c) for r in range(n) for c in range(n)))
    own = rs.randint(0, 2**63, size=B, dtype=np.uint64) & valid
    opp = rs.randint(0, 2**63, size=B, dtype=np.uint64) & valid & ~own
    pi = np.zeros((B, n * n), np.float32)
    pi[np.arange(B), rs.randint(0, n * n, B)] = 1                      # the reference's one-hot targets
    pi[0] = rs.dirichlet(np.ones(n * n)).astype(np.float32)            # and one dense target (visit-count style)
    z = rs.choice([-1.0, 1.0], B).astype(np.float32)
    return own, opp, pi, z


def _both(ref, gpu, batch):
    """GPU step first, then the oracle with the GPU's ReLU decisions for the units within 1e-5 of the kink
    (oracle/train_ref.py:_relu); returns (gpu losses, oracle losses)"""
    own, opp, pi, z = batch
    lg = gpu.forward_backward(own, opp, pi, z)
    # (dense blocks: activations are post-dropout; a dropped unit has a == 0 whatever relu decided, and its gradient is 0
    # on both sides, so its mask value does not matter)
    masks = [(gpu.activation(l, len(z)) > 0).astype(np.float64) for l in range(6)]
    lr_ = ref.forward_backward(own, opp, pi, z, relu_masks=masks)
    return lg, lr_


def _pair(n, C, cin, B, seed, dropout=0.3, clip=0.5, precision="f32"):
    from oracle.train_ref import TrainRef
    from othellozero_amd.trainer import Trainer
    from othellozero_amd.weights import init_weights
    w = init_weights(n, seed=seed, channels=C, randomize_all=True, in_channels=cin)
    ref = TrainRef(w, n, lr=1e-3, clipvalue=clip, dropout=dropout, seed=77)
    gpu = Trainer(n, C, cin, max_batch=B, lr=1e-3, clipvalue=clip, dropout=dropout, seed=77, precision=precision)
    gpu.set_weights(w)
    return ref, gpu


def _check_grads(ref, gpu, scale=3e-4):
    g = gpu.get_grads()
    worst = 0.0
    for i, gr in ref.grads.items():
        gr = gr.numpy()
        tol = scale * np.abs(gr).max() + 1e-7
        if i in BN_BIASES:
            assert np.abs(g[i]).max() <= 1e-5, f"bias {i} behind BN: gradient should be rounding noise, got {np.abs(g[i]).max()}"
            continue
        err = np.abs(g[i].astype(np.float64) - gr).max()
        worst = max(worst, err / tol)
        assert err <= tol, f"gradient {i} (shape {gr.shape}): max err {err:.3e} > {tol:.3e} (max |g| {np.abs(gr).max():.3e})"
    return worst


@pytest.mark.parametrize("n,C,cin,B", [(6, 128, 2, 8), (8, 128, 2, 5), (6, 128, 1, 7), (8, 256, 2, 32), (8, 512, 2, 6),
                                       # large batches: pixel-major GEMM tiles that skip the taps reading only zeros (forward 'same'
                                       # borders, the zero-bordered data-gradient buffers), streaming BN reductions, the board-resident
                                       # weight-gradient kernel with its row splits
                                       (8, 128, 2, 256), (6, 128, 1, 384)])
def test_forward_backward_matches_autograd(n, C, cin, B):
    ref, gpu = _pair(n, C, cin, B, seed=3)
    lg, lr_ = _both(ref, gpu, _batch(n, B, 11, cin))
    assert np.allclose(lg, lr_, atol=2e-5, rtol=2e-5), (lg, lr_)
    p, v = gpu.outputs(B)
    assert np.abs(p - ref.outputs["p"]).max() <= 2e-5 and np.abs(v - ref.outputs["v"]).max() <= 2e-5
    _check_grads(ref, gpu)


@pytest.mark.parametrize("n,C,cin,B", [(8, 256, 2, 32), (8, 512, 2, 6), (6, 256, 1, 40), (8, 256, 2, 192), (8, 512, 2, 37)])      # (>= 32: the f16x2 weight-gradient kernel; 37, 40: a partly filled octet)
def test_forward_backward_f16x2_matches_autograd(n, C, cin, B):
    """precision "f16x2" (3x3 layers' forward and data gradient on the fp16 matrix cores, fp32 values as two fp16 planes, per-tensor
    power-of-two scaling): the same tolerances against the float64 oracle as the fp32 kernels"""
    ref, gpu = _pair(n, C, cin, B, seed=3, precision="f16x2")
    lg, lr_ = _both(ref, gpu, _batch(n, B, 11, cin))
    assert np.allclose(lg, lr_, atol=2e-5, rtol=2e-5), (lg, lr_)
    p, v = gpu.outputs(B)
    assert np.abs(p - ref.outputs["p"]).max() <= 2e-5 and np.abs(v - ref.outputs["v"]).max() <= 2e-5
    _check_grads(ref, gpu)


def test_f16x2_step_at_a_large_batch():
    """512 boards x 512 filters (the 256 x 256 ping-pong tile of the f16x2 GEMM) against the float64 oracle, same tolerances; a second
    call reproduces the first bit for bit"""
    n, C, B = 8, 512, 512
    ref, gpu = _pair(n, C, 2, B, seed=2, precision="f16x2")
    batch = _batch(n, B, 5)
    lg, lr_ = _both(ref, gpu, batch)
    assert np.allclose(lg, lr_, atol=2e-5, rtol=2e-5), (lg, lr_)
    _check_grads(ref, gpu)
    g1 = gpu.get_grads()
    l2 = gpu.forward_backward(*batch)
    g2 = gpu.get_grads()
    assert lg == l2 and all(np.array_equal(g1[i], g2[i]) for i in g1)


def test_f16x2_range_error_is_reported_once_and_the_trainer_recovers():
    """an activation above the fp16 range makes the f16x2 step fail with OZ_ERR_STATE -- once: the sticky flag is cleared when
    it has been reported, so after set_weights reloads good weights the same trainer steps again (and matches a fresh trainer bit for
    bit); weights holding Inf fail loudly too instead of producing garbage power-of-two scales"""
    from othellozero_amd import _lib
    from othellozero_amd.trainer import Trainer
    from othellozero_amd.weights import init_weights
    n, C, B = 6, 256, 16
    w = init_weights(n, seed=5, channels=C, randomize_all=True)
    batch = _batch(n, B, 3)
    make = lambda: Trainer(n, C, 2, max_batch=B, lr=1e-3, clipvalue=0.5, dropout=0.0, seed=1, precision="f16x2")
    good = make(); good.set_weights(w)
    want = good.forward_backward(*batch)
    t = make()
    big = [a.copy() for a in w]
    big[2] = big[2] * 1e7                                     # gamma of the first BN: activations ~1e7
    t.set_weights(big)
    with pytest.raises(_lib.OzError) as e:
        t.forward_backward(*batch)
    assert e.value.code == _lib.OZ_ERR_STATE and "fp16 range" in str(e.value)
    t.set_weights(w)
    got = t.forward_backward(*batch)                          # the flag was cleared when it was reported
    assert np.array_equal(np.asarray(got), np.asarray(want))
    bad = [a.copy() for a in w]
    bad[12][0, 0, 0, 0] = np.inf                              # conv3 kernel
    t.set_weights(bad)
    with pytest.raises(_lib.OzError) as e:
        t.forward_backward(*batch)
    assert e.value.code == _lib.OZ_ERR_STATE
    t.set_weights(w)
    assert np.array_equal(np.asarray(t.forward_backward(*batch)), np.asarray(want))


def test_no_dropout_no_clip_and_determinism():
    ref, gpu = _pair(6, 128, 2, 16, seed=4, dropout=0.0, clip=0.0)
    own, opp, pi, z = _batch(6, 16, 12)
    l1, _ = _both(ref, gpu, (own, opp, pi, z))
    g1 = gpu.get_grads()
    _check_grads(ref, gpu)
    l2 = gpu.forward_backward(own, opp, pi, z)
    g2 = gpu.get_grads()
    assert l1 == l2 and all(np.array_equal(g1[i], g2[i]) for i in g1)          # fixed-order reductions: bit-reproducible


@pytest.mark.parametrize("precision,C", [("f32", 128), ("f16x2", 256)])
def test_adam_steps_and_moving_statistics_match(precision, C):
    """Step-locked comparison: at every step the gradients are checked against autograd at the (matched) weights, then
    BOTH sides apply Adam to the GPU's gradients, so the update rule and the BN statistics are compared exactly.
    (Letting each side use its own gradients is not a parity test: where |g| is at rounding level Adam's m / sqrt(v)
    amplifies the noise to steps of +-lr, in TensorFlow as much as here.)"""
    import torch
    n, B, steps = 6, 8, 4             # (f16x2: the weight operands and their power-of-two scales are rebuilt from the moved weights every step)
    ref, gpu = _pair(n, C, 2, B, seed=5, precision=precision)
    w0 = ref.weights()
    for s in range(steps):
        lg, lr_ = _both(ref, gpu, _batch(n, B, 20 + s))
        assert np.allclose(lg, lr_, atol=5e-5, rtol=5e-5), (s, lg, lr_)
        _check_grads(ref, gpu)
        ref.apply(grads={i: torch.tensor(g.astype(np.float64)) for i, g in gpu.get_grads().items()})
        gpu.apply()
        wr, wg = ref.weights(), gpu.get_weights()
        for i in range(40):
            err = np.abs(wg[i].astype(np.float64) - wr[i]).max()
            assert err <= 2e-6, f"step {s}, weight {i}: {err:.3e}"
    assert gpu.step == steps
    assert np.abs(wg[6] - w0[6]).max() > 5e-4                 # the steps really moved the weights (~lr per element per step)
    assert np.abs(wg[4] - w0[4]).max() > 1e-4 and np.abs(wg[29] - w0[29]).max() > 1e-4      # moving statistics too


def test_free_running_steps_stay_close():
    """each side on its own gradients for a few steps: everything but the rounding-level-gradient elements agrees"""
    n, C, B, steps = 6, 128, 8, 3
    ref, gpu = _pair(n, C, 2, B, seed=8)
    for s in range(steps):
        own, opp, pi, z = _batch(n, B, seed=40 + s)
        lr_ = ref.forward_backward(own, opp, pi, z)
        lg = gpu.forward_backward(own, opp, pi, z)
        assert np.allclose(lg, lr_, atol=2e-4, rtol=2e-4), (s, lg, lr_)
        ref.apply()
        gpu.apply()
    wr, wg = ref.weights(), gpu.get_weights()
    for i in range(40):
        d = np.abs(wg[i].astype(np.float64) - wr[i])
        assert d.max() <= steps * 2e-3 * 1.01, f"weight {i}: {d.max():.3e}"              # never more than 2 lr per step
        if i not in BN_BIASES and d.size >= 1000:
            assert (d > 3e-5).mean() < 0.02, f"weight {i}: {(d > 3e-5).mean():.4f} of the elements off by > 3e-5"


def test_clipvalue_is_applied():
    """a huge loss scale (targets far outside tanh's range) pushes gradients past 0.5: with clipvalue the first Adam
    step is the same +-lr, but m and v differ -- checked through the second step against the oracle."""
    n, C, B = 6, 128, 8
    ref, gpu = _pair(n, C, 2, B, seed=6, dropout=0.0, clip=0.5)
    own, opp, pi, z = _batch(n, B, seed=31)
    z = z * 500.0
    for _ in range(2):
        ref.forward_backward(own, opp, pi, z)
        gpu.forward_backward(own, opp, pi, z)
        assert max(np.abs(g.numpy()).max() for g in ref.grads.values()) > 0.5
        ref.apply()
        gpu.apply()
    wr, wg = ref.weights(), gpu.get_weights()
    for i in (0, 6, 24, 30, 36, 38):
        d = np.abs(wg[i].astype(np.float64) - wr[i])
        assert (d > 3e-5).mean() < 0.02 and d.max() <= 4.1e-3


def test_nnetwrapper_train_drop_in():
    """NNetWrapper.train(examples) with the reference's example tuples; loss goes down on a fixed set, the inference
    network afterwards carries the trained weights (predict == oracle forward of get_weights())."""
    from oracle import nn_numpy
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.trainer import pack_examples
    n, N = 6, 96
    rs = np.random.RandomState(2)
    examples = []
    for _ in range(N):
        occ = rs.rand(n, n) < 0.6
        black = occ & (rs.rand(n, n) < 0.5)
        board = np.stack([black, occ & ~black], axis=2)
        pol = np.zeros((n, n))
        pol[rs.randint(n), rs.randint(n)] = 1
        examples.append((board, pol, int(rs.choice([-1, 1]))))
    net = NNetWrapper((n, n), num_channels_1=128, batch_size=32, epochs=6, max_batch=4)
    w_before = net.get_weights()
    hist = net.train(examples)
    assert len(hist.history["loss"]) == 6 and hist.history["loss"][-1] < hist.history["loss"][0]
    assert all(np.isfinite(hist.history[k]).all() for k in hist.history)
    w_after = net.get_weights()
    assert np.abs(w_after[6] - w_before[6]).max() > 1e-3 and not np.array_equal(w_after[4], w_before[4])
    own, opp, _, _ = pack_examples(examples[:4], n)
    pi, v = net.predict_batch(own, opp)
    pr, vr = nn_numpy.forward(w_after, own, opp, n)
    assert np.abs(pi.reshape(4, -1) - pr).max() <= 1e-5 and np.abs(v - vr).max() <= 1e-5
    hist2 = net.train(examples[:40])                                   # short last batch (40 = 32 + 8), optimiser state kept
    assert net._trainer.step == 6 * 3 + 6 * 2 and np.isfinite(hist2.history["loss"]).all()


@pytest.mark.parametrize("precision,C", [("f32", 128), ("f16x2", 256)])
def test_resident_dataset_fit_equals_stepwise_fit(precision, C):
    """trainer.fit with the examples resident in HBM (one upload, one library call per epoch, batches gathered by index on
    the device, no per-step synchronisation) takes exactly the optimiser steps of the step-wise loop: identical weights
    bit for bit after two epochs with a short last batch, the same epoch-mean losses"""
    from othellozero_amd.trainer import Trainer, fit
    from othellozero_amd.weights import init_weights
    n, N, bs = 6, 75, 16
    own, opp, pi, z = _batch(n, N, seed=77)
    runs = []
    for resident in (False, True):
        tr = Trainer(n, C, 2, max_batch=bs, seed=9, precision=precision)
        tr.set_weights(init_weights(n, seed=1, channels=C))
        h = fit(tr, own, opp, pi, z, batch_size=bs, epochs=2, shuffle_seed=5, resident=resident)
        runs.append((tr.get_weights(), h.history, tr.step))
    (w0, h0, s0), (w1, h1, s1) = runs
    assert s0 == s1 == 2 * 5
    assert all(np.array_equal(a, b) for a, b in zip(w0, w1))
    for k in h0:
        assert np.allclose(h0[k], h1[k], rtol=1e-6, atol=1e-7), (k, h0[k], h1[k])


DP_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from othellozero_amd.distributed import GradientAllReduce, average_moving_statistics
from othellozero_amd.trainer import Trainer, fit
from othellozero_amd.weights import init_weights
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)          # control-flow rehearsal: both ranks share GPU 0
torch.cuda.set_device(0)
n, B, N = 6, 8, 24
precision = sys.argv[2]
C = 256 if precision == "f16x2" else 128
w = init_weights(n, seed=3, channels=C, randomize_all=True)
def shard(r):
    rs = np.random.RandomState(100 + r)
    valid = np.uint64(sum(1 << (y * 8 + x) for y in range(n) for x in range(n)))
    own = rs.randint(0, 2**63, size=N, dtype=np.uint64) & valid
    opp = rs.randint(0, 2**63, size=N, dtype=np.uint64) & valid & ~own
    pi = np.zeros((N, n * n), np.float32); pi[np.arange(N), rs.randint(0, n * n, N)] = 1
    return own, opp, pi, rs.choice([-1.0, 1.0], N).astype(np.float32)
ar = GradientAllReduce(n, C, 2, device="cuda")
tr = Trainer(n, C, 2, max_batch=B, seed=9, external_grads_ptr=ar.ptr, precision=precision)
tr.set_weights(w)
hist = fit(tr, *shard(rank), batch_size=B, epochs=2, shuffle_seed=5, allreduce=ar)
mine = tr.get_weights()
flat = torch.from_numpy(np.concatenate([mine[i].ravel() for i in range(40) if i >= 36 or i % 6 not in (4, 5)]))
both = [torch.zeros_like(flat) for _ in range(2)]
dist.all_gather(both, flat)
assert torch.equal(both[0], both[1]), "trainable weights diverged between the ranks"
avg = average_moving_statistics(mine)
if rank == 0:
    # single-process emulation of the same job: two trainers, gradients averaged by hand between backward and apply
    g = [torch.zeros(Trainer.arena_size(n, C, 2), dtype=torch.float32, device="cuda") for _ in range(2)]
    t2 = [Trainer(n, C, 2, max_batch=B, seed=9, external_grads_ptr=g[r].data_ptr(), precision=precision) for r in range(2)]
    data = [shard(r) for r in range(2)]
    for t in t2: t.set_weights(w)
    for ep in range(2):
        order = np.random.RandomState(5 + ep).permutation(N)
        for s in range(0, N, B):
            idx = order[s:s + B]
            for r in range(2):
                t2[r].forward_backward(data[r][0][idx], data[r][1][idx], data[r][2][idx], data[r][3][idx])
            m = (g[0] + g[1]) / 2
            g[0].copy_(m); g[1].copy_(m); torch.cuda.synchronize()
            for t in t2: t.apply()
    e0, e1 = t2[0].get_weights(), t2[1].get_weights()
    for i in range(40):
        assert np.array_equal(e0[i], mine[i]), f"weight {i} differs from the single-process emulation"
        if i < 36 and i % 6 in (4, 5):
            assert np.allclose(avg[i], (e0[i] + e1[i]) / 2, atol=1e-7)
    assert np.isfinite(hist.history["loss"]).all()
dist.barrier()
print("RANK_OK", rank)
'''


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
def test_data_parallel_two_ranks_equal_hand_averaged_gradients(tmp_path, precision):
    """GradientAllReduce + fit on two ranks (gloo, both on GPU 0): weights stay identical across ranks and equal, bit
    for bit, a single-process run that averages the two gradient arenas by hand -- in both trainer precisions."""
    import os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "dp_gpu_worker.py"
    script.write_text(DP_WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), root, precision], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {rank}" in out, out


RCCL_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from othellozero_amd import _lib
from othellozero_amd.NNet import NNetWrapper
from othellozero_amd.distributed import (GradientAllReduce, engine_records_tensor, gather_records, tensor_to_records)
from othellozero_amd.trainer import Trainer
from othellozero_amd.training import SelfPlayEngine
from othellozero_amd.weights import init_weights
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)            # backend "nccl" = RCCL
assert dist.get_backend() == "nccl"
# the self-play exchange: the engine's records, device to device, through all_gather + all_gather_into_tensor on uint8
n, C = 6, 128
net = NNetWrapper((n, n), num_channels_1=C, max_batch=16, seed=0)
eng = SelfPlayEngine(net, n, 16, 8, 1.0, 1.0, 0.9, seed=7, q_mode=_lib.QMODE_F64)
own = eng.play_to_end()
local = engine_records_tensor(eng, dev)
pooled = gather_records(local, single_rank_collective=True)
assert pooled.is_cuda and pooled.shape == local.shape and torch.equal(pooled, local)
rec = tensor_to_records(pooled)
key = lambda a: a[np.lexsort((a["ply"], a["game_id"]))].tobytes()
assert len(rec) == len(own) and key(rec) == key(np.ascontiguousarray(own))
empty = gather_records(local[:0], single_rank_collective=True)                  # a rank with no finished game
assert empty.shape[0] == 0
# the training exchange: one all-reduce over the gradient arena the backward kernels wrote, in place
ar = GradientAllReduce(n, C, 2, device="cuda", single_rank_collective=True)
tr = Trainer(n, C, 2, max_batch=8, seed=9, external_grads_ptr=ar.ptr)
tr.set_weights(init_weights(n, seed=3, channels=C, randomize_all=True))
rs = np.random.RandomState(1)
valid = np.uint64(sum(1 << (y * 8 + x) for y in range(n) for x in range(n)))
o = rs.randint(0, 2**63, size=8, dtype=np.uint64) & valid
p = rs.randint(0, 2**63, size=8, dtype=np.uint64) & valid & ~o
pi = np.zeros((8, n * n), np.float32); pi[np.arange(8), rs.randint(0, n * n, 8)] = 1
tr.forward_backward(o, p, pi, rs.choice([-1.0, 1.0], 8).astype(np.float32))
tr.sync()
before = ar.flat.clone()
assert float(before.abs().sum()) > 0
ar(tr)                                                                           # sum over one rank / 1: the same bits
assert torch.equal(ar.flat, before)
tr.apply()
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
'''


def test_rccl_calls_on_one_rank(tmp_path):
    """the two RCCL exchanges (all-gather of move records, all-reduce of the gradient arena) driven through a one-rank
    "nccl" process group on the card: the calls, dtypes and device buffers the N-GPU job uses, on the hardware this box has"""
    import os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout + r.stderr


def test_trained_network_in_split_precision_matches_float64():
    """after real optimiser steps (weights, biases and BN statistics no longer at their initial values) the f16x2 inference
    kernels still reproduce the float64 forward of the trained weights within the 1e-5 tolerance of the hot path"""
    from oracle import nn_numpy
    from othellozero_amd.NNet import NNetWrapper
    n, N = 6, 128
    rs = np.random.RandomState(5)
    examples = []
    for _ in range(N):
        occ = rs.rand(n, n) < 0.7
        black = occ & (rs.rand(n, n) < 0.5)
        pol = np.zeros((n, n))
        pol[rs.randint(n), rs.randint(n)] = 1
        examples.append((np.stack([black, occ & ~black], axis=2), pol, int(rs.choice([-1, 1]))))
    net = NNetWrapper((n, n), num_channels_1=256, batch_size=32, epochs=5, max_batch=64, precision="f16x2")
    net.train(examples)                                                    # 20 Adam steps
    w = net.get_weights()
    valid = np.uint64(sum(1 << (r * 8 + c) for r in range(n) for c in range(n)))
    own = rs.randint(0, 2**63, size=64, dtype=np.uint64) & valid
    opp = rs.randint(0, 2**63, size=64, dtype=np.uint64) & valid & ~own
    pi, v = net.predict_batch(own, opp)
    pr, vr = nn_numpy.forward(w, own, opp, n)
    assert np.abs(pi.reshape(64, -1) - pr).max() <= 1e-5 and np.abs(v - vr).max() <= 1e-5
    assert np.abs(np.asarray(w[4])).max() > 1e-3 and np.abs(np.asarray(w[5]) - 1).max() > 1e-3     # BN statistics moved
