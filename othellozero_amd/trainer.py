"""Training step on the GPU -- host mirror of NNetWrapper.train (Net/NNet.py:53-68); C ABI: oz_trainer_*.

`Trainer` is the step-level object (forward + backward into a gradient arena, Adam apply); `fit(...)` is the epoch loop
of keras Model.fit as the reference drives it (shuffle every epoch, batches of `batch_size`, last batch may be short,
per-epoch mean losses in a History-like object).  Data-parallel training: every rank runs `fit` on its shard of the
examples with the same seed and weights, gradients are averaged with one all-reduce per step (distributed.py).
"""
import ctypes as C

import numpy as np

from . import _lib
from .weights import onn_shapes

TRAINABLE = [i for i in range(40) if i >= 36 or i % 6 not in (4, 5)]


class History:
    """what keras Model.fit returns, as far as the reference uses it (main.py:108: kept, never read)"""

    def __init__(self):
        self.history = {"loss": [], "pi-reshaped_loss": [], "v_loss": []}
        self.epoch = []


class Trainer:
    def __init__(self, board_size=8, channels=512, in_channels=2, max_batch=32, lr=1e-3, clipvalue=0.5, dropout=0.3,
                 bn_momentum=0.99, seed=0, external_grads_ptr=None, precision="f32"):
        """precision: arithmetic of the 3x3 layers' forward / data-gradient GEMMs -- "f32" (fp32 matrix cores) or "f16x2"
        (fp32 values as two fp16 planes on the fp16 matrix cores, the inference kernels' arithmetic; channels % 256 == 0)"""
        assert precision in ("f32", "f16x2"), precision
        lib = _lib.require_gpu()
        self.n, self.channels, self.in_channels, self.max_batch = board_size, channels, in_channels, int(max_batch)
        self._h = C.c_void_p()
        _lib.check(lib.oz_trainer_create(C.byref(self._h), board_size, channels, in_channels, self.max_batch, lr,
                                         clipvalue if clipvalue else 0.0, dropout, bn_momentum, seed,
                                         C.c_void_p(external_grads_ptr) if external_grads_ptr else None))
        self.shapes = onn_shapes(board_size, channels, in_channels)
        self.precision = precision
        if precision == "f16x2":
            _lib.check(lib.oz_trainer_set_precision(self._h, 1))

    def __del__(self):
        try:
            if self._h:
                _lib.load().oz_trainer_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    @staticmethod
    def arena_size(board_size, channels, in_channels=2):
        n = C.c_int64()
        _lib.check(_lib.load().oz_trainer_arena_size(board_size, channels, in_channels, C.byref(n)))
        return n.value

    def set_weights(self, weights):
        lib = _lib.load()
        assert len(weights) == 40
        for i, (w, shp) in enumerate(zip(weights, self.shapes)):
            a = np.ascontiguousarray(w, dtype=np.float32)
            assert a.shape == tuple(shp), f"weight {i}: shape {a.shape} != {shp}"
            _lib.check(lib.oz_trainer_set_weight(self._h, i, _lib.p_f32(a), a.size))

    def get_weights(self):
        lib, out = _lib.load(), []
        for i, shp in enumerate(self.shapes):
            a = np.zeros(shp, dtype=np.float32)
            _lib.check(lib.oz_trainer_get_weight(self._h, i, _lib.p_f32(a), a.size))
            out.append(a)
        return out

    def get_grads(self):
        """{index: gradient} of the trainable arrays after forward_backward"""
        lib, out = _lib.load(), {}
        for i in TRAINABLE:
            a = np.zeros(self.shapes[i], dtype=np.float32)
            _lib.check(lib.oz_trainer_get_grad(self._h, i, _lib.p_f32(a), a.size))
            out[i] = a
        return out

    def grad_arena(self):
        ptr, n = C.c_void_p(), C.c_int64()
        _lib.check(_lib.load().oz_trainer_grad_arena(self._h, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def forward_backward(self, own, opp, pi_target, z_target):
        """-> (loss, pi loss, v loss) of the batch; gradients of the batch-mean loss are left in the arena"""
        own = np.ascontiguousarray(own, dtype=np.uint64).ravel()
        opp = np.ascontiguousarray(opp, dtype=np.uint64).ravel()
        B = own.size
        pit = np.ascontiguousarray(pi_target, dtype=np.float32).reshape(B, self.n * self.n)
        zt = np.ascontiguousarray(z_target, dtype=np.float32).reshape(B)
        losses = np.zeros(3, np.float32)
        _lib.check(_lib.load().oz_trainer_forward_backward(self._h, _lib.p_u64(own), _lib.p_u64(opp), _lib.p_f32(pit),
                                                           _lib.p_f32(zt), B, _lib.p_f32(losses)))
        return float(losses[0]), float(losses[1]), float(losses[2])

    def set_dataset(self, own, opp, pi, z):
        """upload the examples of a fit once; they stay in HBM until the next set_dataset"""
        own = np.ascontiguousarray(own, dtype=np.uint64); opp = np.ascontiguousarray(opp, dtype=np.uint64)
        pi = np.ascontiguousarray(pi, dtype=np.float32).reshape(own.size, self.n * self.n)
        z = np.ascontiguousarray(z, dtype=np.float32)
        _lib.check(_lib.load().oz_trainer_set_dataset(self._h, _lib.p_u64(own), _lib.p_u64(opp), _lib.p_f32(pi), _lib.p_f32(z), own.size))

    def fit_epoch(self, order, batch_size):
        """the optimiser steps of one epoch over the resident examples in `order` (batches of batch_size, last one short);
        -> sample-weighted mean (loss, policy loss, value loss)"""
        order = np.ascontiguousarray(order, dtype=np.int32)
        out = np.zeros(3, np.float32)
        _lib.check(_lib.load().oz_trainer_fit_epoch(self._h, _lib.p_i32(order), order.size, int(batch_size), _lib.p_f32(out)))
        return out

    def apply(self):
        _lib.check(_lib.load().oz_trainer_apply(self._h))

    def sync(self):
        _lib.check(_lib.load().oz_trainer_sync(self._h))

    def outputs(self, B):
        p, v = np.zeros((B, self.n * self.n), np.float32), np.zeros(B, np.float32)
        _lib.check(_lib.load().oz_trainer_outputs(self._h, B, _lib.p_f32(p), _lib.p_f32(v)))
        return p, v

    def activation(self, layer, B):
        """post-activation output of block `layer` (0-3 conv: (B, H, H, C); 4-5 dense: (B, units)) of the last forward pass"""
        n, C_ = self.n, self.channels
        shape = [(B, n, n, C_), (B, n, n, C_), (B, n - 2, n - 2, C_), (B, n - 4, n - 4, C_), (B, 1024), (B, 512)][layer]
        a = np.zeros(shape, np.float32)
        _lib.check(_lib.load().oz_trainer_get_activation(self._h, layer, B, _lib.p_f32(a), a.size))
        return a

    @property
    def step(self):
        s = C.c_int64()
        _lib.check(_lib.load().oz_trainer_step_count(self._h, C.byref(s)))
        return s.value


def pack_examples(examples, board_size, in_channels=2):
    """the reference's example tuples (board, policy (n,n), z) -> (own u64[N], opp u64[N], pi f32[N, n*n], z f32[N]).
    ONN boards are (n,n,2) {0,1} (channel 0 -> own, 1 -> opp); BNN boards are (n,n) with +1 / -1."""
    N, n = len(examples), board_size
    boards = np.asarray([e[0] for e in examples])
    if in_channels == 1:
        boards = np.stack([boards == 1, boards == -1], axis=-1)
    boards = boards.astype(bool).reshape(N, n, n, 2)
    weights = (np.uint64(1) << (np.arange(n, dtype=np.uint64)[:, None] * np.uint64(8) + np.arange(n, dtype=np.uint64)[None, :]))
    own = (boards[..., 0] * weights).sum(axis=(1, 2), dtype=np.uint64)
    opp = (boards[..., 1] * weights).sum(axis=(1, 2), dtype=np.uint64)
    pi = np.asarray([e[1] for e in examples], dtype=np.float32).reshape(N, n * n)
    z = np.asarray([e[2] for e in examples], dtype=np.float32).reshape(N)
    return own, opp, pi, z


def fit(trainer, own, opp, pi, z, batch_size=32, epochs=10, shuffle_seed=0, allreduce=None, verbose=None, resident=None):
    """keras Model.fit(x, y, batch_size, epochs) with shuffle=True (the default the reference relies on).
    `allreduce(trainer)` -- if given -- averages the gradient arena across ranks between backward and apply.

    resident (default: True without an all-reduce): the examples are uploaded ONCE (oz_trainer_set_dataset) and every
    epoch is one library call that runs its optimiser steps back to back on the device (oz_trainer_fit_epoch) -- the
    same shuffled order, batches, steps and weights as the step-wise loop, without a host copy or a synchronisation
    per step.  With an all-reduce (data-parallel training) the loop stays step-wise: the collective sits between
    backward and apply."""
    N = len(z)
    hist = History()
    if resident is None:
        resident = allreduce is None
    assert not (resident and allreduce is not None), "the resident path has no room for a per-step all-reduce"
    if resident:
        trainer.set_dataset(own, opp, pi, z)
    for ep in range(epochs):
        order = np.random.RandomState(shuffle_seed + ep).permutation(N)
        if resident:
            means = np.asarray(trainer.fit_epoch(order, batch_size), dtype=np.float64)
        else:
            tot, seen = np.zeros(3), 0
            for s in range(0, N, batch_size):
                idx = order[s:s + batch_size]
                failed = None
                try:
                    losses = trainer.forward_backward(own[idx], opp[idx], pi[idx], z[idx])
                except Exception as e:                           # the f16x2 range guard, a bad batch shape, an out-of-memory: this rank's step is invalid
                    if allreduce is None:
                        raise
                    failed, losses = e, np.zeros(3)
                if allreduce is not None:
                    # a rank that failed still joins the collective (with a poisoned arena), so no peer blocks in the all-reduce and
                    # EVERY rank raises together
                    allreduce(trainer, failed=failed)
                trainer.apply()
                tot += np.asarray(losses) * len(idx)            # keras reports the sample-weighted running mean
                seen += len(idx)
            means = tot / max(seen, 1)
        for k, val in zip(("loss", "pi-reshaped_loss", "v_loss"), means):
            hist.history[k].append(float(val))
        hist.epoch.append(ep)
        if verbose:
            print(f"Epoch {ep + 1}/{epochs} - loss: {hist.history['loss'][-1]:.4f} - pi-reshaped_loss: "
                  f"{hist.history['pi-reshaped_loss'][-1]:.4f} - v_loss: {hist.history['v_loss'][-1]:.4f}")
    trainer.sync()
    return hist
