"""NNetWrapper -- drop-in for Net/NNet.py:22-101 (inference side) on the HIP library.

Same constructor arguments, `.network_type`, `.predict(board) -> (pi (n,n) float32, v float32)`,
`.copy()`, `.save_checkpoint(path)`, `.load_checkpoint(path)`.  Added: `.predict_batch`,
`.get_weights()/.set_weights()` (keras Model.get_weights() order), `StubNetWrapper`.
Checkpoints are Keras HDF5 weight files (keras_h5.py).  `.train` (Net/NNet.py:53-68) runs the optimiser steps on the GPU
(oz_trainer_*, trainer.py).
"""
import ctypes as C
from enum import Enum, auto

import numpy as np

from . import _lib, keras_h5
from .weights import init_weights, onn_shapes


class NeuralNets(Enum):          # Net/NNet.py:14-16
    ONN = auto()
    BNN = auto()


class _NetHandle:
    def __init__(self):
        self._h = C.c_void_p()

    def __del__(self):
        try:
            if self._h:
                _lib.load().oz_net_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


PRECISION_MODES = {"f32": 0, "f16x2": 1, "bf16x3": 2}      # oz_net_set_precision


class NNetWrapper(_NetHandle):
    _models_built = 0          # Keras numbers layer names per process (conv2d, ..., conv2d_4, ...); checkpoints keep that

    def __init__(self, board_size=(8, 8), batch_size=32, epochs=10, num_channels_1=512, num_channels_2=256,
                 lr=0.001, dropout=0.3, network=NeuralNets.ONN, max_batch=1, seed=0, weights=None, precision="f32"):
        super().__init__()
        self._model_index = NNetWrapper._models_built
        NNetWrapper._models_built += 1
        self.board_size_x, self.board_size_y = board_size
        assert self.board_size_x == self.board_size_y, "square boards only"
        self.action_size = self.board_size_x * self.board_size_y
        self.batch_size, self.epochs, self.lr, self.dropout = batch_size, epochs, lr, dropout
        self.network_type = network
        self.num_channels = num_channels_1          # num_channels_2 is accepted but unused (OthelloNN.py:39)
        self.max_batch = int(max_batch)
        if network not in (NeuralNets.ONN, NeuralNets.BNN):
            raise Exception('Invalid Network Type.')
        # BNN (Net/BaseNN.py): the same trunk on the one-channel board (+1 BLACK / -1 WHITE, mover-canonical in the search)
        self.in_channels = 2 if network is NeuralNets.ONN else 1
        lib = _lib.require_gpu()
        create = lib.oz_net_create if network is NeuralNets.ONN else lib.oz_net_create_bnn
        _lib.check(create(C.byref(self._h), self.board_size_x, self.num_channels, self.max_batch))
        # precision: "f32" = exact fp32 matrix cores; "f16x2" = f32 via 2 x fp16 split on the 16-bit matrix cores
        # (fp32-equivalent: same <= 1e-5 tolerance, ~4x faster; needs channels % 256 == 0; every activation channel / weight column is moved into the
        #  fp16 window by an exact power of two at commit, a commit-time self-check against the exact-fp32 kernels refuses networks that amplify
        #  rounding, and a position whose activations leave the calibrated range raises OzError(OZ_ERR_STATE) instead of returning a degraded answer)
        self.precision = precision
        self.requested_precision = precision          # what the caller asked for: a refused f16x2 commit falls back for THAT set of weights only
        self.f16x2_refusals = 0                       # commits precision f16x2 refused so far (train() reports it in its history)
        _lib.check(lib.oz_net_set_precision(self._h, PRECISION_MODES[precision]))
        self.set_weights(weights if weights is not None else
                         init_weights(self.board_size_x, seed, self.num_channels, in_channels=self.in_channels))

    # ---- weights (model.get_weights / set_weights, Net/NNet.py:98-101)
    def set_weights(self, weights, on_refusal="raise"):
        """model.set_weights + oz_net_commit.  Precision f16x2 may REFUSE a network at commit (OZ_ERR_STATE: a guard fired on the calibration
        positions, or the self-check against the exact-fp32 kernels measured more than 8e-6).  on_refusal="raise" (default): the OzError
        propagates -- callers that load arbitrary weights decide themselves.  on_refusal="f32": the wrapper warns, switches THIS network to
        precision f32 (exact fp32 kernels on the GPU, always valid) and commits again -- what train() uses, so that a long f16x2 run is not
        ended by the weights of one iteration."""
        lib = _lib.load()
        if self.precision != self.requested_precision:        # an earlier set of weights was refused: THIS one gets the requested arithmetic again
            _lib.check(lib.oz_net_set_precision(self._h, PRECISION_MODES[self.requested_precision]))
            self.precision = self.requested_precision
        shapes = onn_shapes(self.board_size_x, self.num_channels, self.in_channels)
        assert len(weights) == len(shapes), f"expected {len(shapes)} arrays"
        for i, (w, shp) in enumerate(zip(weights, shapes)):
            a = np.ascontiguousarray(w, dtype=np.float32)
            assert a.shape == tuple(shp), f"weight {i}: shape {a.shape} != {shp}"
            _lib.check(lib.oz_net_set_weight(self._h, i, _lib.p_f32(a), a.size))
        try:
            _lib.check(lib.oz_net_commit(self._h))
        except _lib.OzError as e:
            if not (on_refusal == "f32" and e.code == _lib.OZ_ERR_STATE and self.precision == "f16x2"):
                raise
            import warnings
            self.f16x2_refusals += 1
            warnings.warn(f"othellozero_amd: precision f16x2 refused these weights at commit ({e}); THESE weights run in precision f32 "
                          f"(refusal {self.f16x2_refusals}; the next set_weights tries f16x2 again)")
            _lib.check(lib.oz_net_set_precision(self._h, 0))
            self.precision = "f32"
            _lib.check(lib.oz_net_commit(self._h))

    def init_random(self, seed):
        """a fresh network as Keras initialises it (glorot_uniform kernels, zero biases, identity BatchNormalization) from the LIBRARY's own
        deterministic stream (oz_net_init_random: what a C host gets; not NumPy's numbers for that seed), committed"""
        lib = _lib.load()
        _lib.check(lib.oz_net_init_random(self._h, int(seed)))
        _lib.check(lib.oz_net_commit(self._h))

    def get_weights(self):
        lib = _lib.load()
        out = []
        for i, shp in enumerate(onn_shapes(self.board_size_x, self.num_channels, self.in_channels)):
            a = np.zeros(shp, dtype=np.float32)
            _lib.check(lib.oz_net_get_weight(self._h, i, _lib.p_f32(a), a.size))
            out.append(a)
        return out

    # ---- inference
    def predict_batch(self, own, opp):
        """canonical bitboards (uint64 arrays) -> pi (B, n, n) float32, v (B,) float32"""
        own = np.ascontiguousarray(own, dtype=np.uint64).ravel()
        opp = np.ascontiguousarray(opp, dtype=np.uint64).ravel()
        B, n = own.size, self.board_size_x
        pi = np.zeros((B, n, n), dtype=np.float32)
        v = np.zeros(B, dtype=np.float32)
        lib = _lib.load()
        for s in range(0, B, self.max_batch):
            e = min(B, s + self.max_batch)
            _lib.check(lib.oz_net_predict(self._h, _lib.p_u64(own[s:e]), _lib.p_u64(opp[s:e]), e - s,
                                          _lib.p_f32(pi[s:e]), _lib.p_f32(v[s:e])))
        return pi, v

    def predict(self, board):
        """Net/NNet.py:70-87: board (n,n,2) [ONN] or one-channel (n,n) with +1 / -1 [BNN] -> (pi.reshape(n,n), v[0][0])"""
        if self.in_channels == 1:
            b = np.asarray(board)
            board = np.stack([b == 1, b == -1], axis=2)
        own, opp = _lib.pack_board(board)
        pi, v = self.predict_batch(np.array([own], np.uint64), np.array([opp], np.uint64))
        return pi[0], v[0]

    def train(self, examples, verbose=None, seed=None, allreduce=None):
        """Net/NNet.py:53-68: model.fit(x=boards, y=[pis, vs], batch_size=self.batch_size, epochs=self.epochs) on the GPU
        (oz_trainer_*: MFMA forward / backward in the wrapper's precision -- "f16x2" runs the 3x3 layers' forward and data
        gradient on the fp16 matrix cores like the inference kernels (needs num_channels % 256 == 0, else fp32) -- Adam
        lr=self.lr with clipvalue 0.5 for ONN / none for BNN, Dropout self.dropout, BN momentum 0.99).  Returns a History-like object (`.history['loss']`, ...).  The TensorBoard
        callback of the reference is not reproduced.  Optimiser state persists across calls like the compiled Keras model's."""
        from . import trainer as T
        if not examples:
            return T.History()
        own, opp, pi, z = T.pack_examples(examples, self.board_size_x, self.in_channels)
        if getattr(self, "_trainer", None) is None:
            assert self.num_channels % 128 == 0, "the training kernels need num_channels % 128 == 0"
            # a GradientAllReduce owns the gradient arena (a torch tensor RCCL can reduce in place): the library writes into it
            self._trainer = T.Trainer(self.board_size_x, self.num_channels, self.in_channels, max_batch=self.batch_size, lr=self.lr,
                                      clipvalue=0.5 if self.network_type is NeuralNets.ONN else 0.0, dropout=self.dropout,
                                      seed=self._model_index if seed is None else seed,
                                      external_grads_ptr=getattr(allreduce, "ptr", None),
                                      precision="f16x2" if self.precision == "f16x2" and self.num_channels % 256 == 0 else "f32")
            self._trainer_arena = getattr(allreduce, "ptr", None)
            self._fit_calls = 0
        assert getattr(allreduce, "ptr", None) == self._trainer_arena, "train() must keep using the GradientAllReduce it started with"
        self._trainer.set_weights(self.get_weights())
        hist = T.fit(self._trainer, own, opp, pi, z, batch_size=self.batch_size, epochs=self.epochs,
                     shuffle_seed=1000003 * self._fit_calls + (self._model_index if seed is None else seed), allreduce=allreduce,
                     verbose=verbose)
        self._fit_calls += 1
        weights = self._trainer.get_weights()
        if allreduce is not None:                       # BN moving statistics are per replica: average them
            from .distributed import average_moving_statistics
            weights = average_moving_statistics(weights, getattr(allreduce, "group", None))
        self.set_weights(weights, on_refusal="f32")
        if allreduce is not None:
            # the replicas hold the same weights, so a refusal is the same decision everywhere; should one rank's commit still be refused alone
            # (another calibration outcome on its device, say), EVERY rank evaluates these weights in precision f32 -- no rank decides by itself
            from .distributed import any_rank
            refused = self.precision != self.requested_precision
            if any_rank(refused, getattr(allreduce, "group", None)) and not refused:
                lib = _lib.load()
                _lib.check(lib.oz_net_set_precision(self._h, PRECISION_MODES["f32"]))
                self.precision = "f32"
                _lib.check(lib.oz_net_commit(self._h))
        hist.precision = self.precision                     # the arithmetic these weights are evaluated in until the next set_weights
        hist.f16x2_refusals = self.f16x2_refusals           # (attributes, not history entries: Keras' history holds per-epoch number lists only)
        return hist

    # ---- checkpoints (Net/NNet.py:90-96): Keras HDF5 weight files, read and written by keras_h5.py (no h5py needed);
    # files saved by the reference load here and the other way round.  A path ending in .npz selects a plain
    # NumPy archive of the same 40 arrays instead (an extension of this build).
    def save_checkpoint(self, filepath):
        """model.save_weights(filepath, save_format='h5')"""
        if filepath.endswith(".npz"):
            np.savez(filepath, *self.get_weights())
            return
        layers = keras_h5.keras_layer_table(self.get_weights(), self._model_index, self.network_type.name)
        keras_h5.save_keras_weights(filepath, layers)

    def load_checkpoint(self, filepath):
        """model.load_weights(filepath): layers with weights are matched BY ORDER, as Keras does; a file whose
        weighted-layer count or shapes differ from this network raises ValueError."""
        if filepath.endswith(".npz"):
            with np.load(filepath) as z:
                self.set_weights([z[f"arr_{i}"] for i in range(len(z.files))])
            return
        assert filepath.endswith('.h5'), 'Expecting a file with .h5 as extension'
        layers = [(name, ws) for name, ws in keras_h5.load_keras_weights(filepath) if ws]
        if len(layers) != 14:
            raise ValueError(f"You are trying to load a weight file containing {len(layers)} layers into a model with 14 layers.")
        flat = keras_h5.flat_weights(layers)
        shapes = onn_shapes(self.board_size_x, self.num_channels, self.in_channels)
        if len(flat) != len(shapes):
            raise ValueError(f"the file holds {len(flat)} weight arrays, this network has {len(shapes)}")
        for i, (a, shp) in enumerate(zip(flat, shapes)):
            if tuple(a.shape) != tuple(shp):
                raise ValueError(f"weight {i} of the file has shape {tuple(a.shape)}, this network expects {tuple(shp)}")
        self.set_weights(flat)

    def copy(self):
        return NNetWrapper((self.board_size_x, self.board_size_y), network=self.network_type,
                           num_channels_1=self.num_channels, max_batch=self.max_batch, weights=self.get_weights(),
                           precision=self.precision)

    # ---- profiling hooks used by bench.py
    def time_forward(self, count, iters=3):
        ms = C.c_float()
        _lib.check(_lib.load().oz_net_time_forward(self._h, count, iters, C.byref(ms)))
        return ms.value

    def profile(self, enable=True):
        """HIP-event timing on the launch stream: True / 1 = the dominant launch only, 2 = every kernel of the forward, False / 0 = off"""
        _lib.check(_lib.load().oz_net_profile(self._h, int(enable)))

    def profile_kernels(self, reset=False):
        """{kernel: (ms_total, launches)} for _lib.NET_KERNELS (slots the profile mode did not time stay at their last value)"""
        k = len(_lib.NET_KERNELS)
        ms, cnt = np.zeros(k, np.float64), np.zeros(k, np.int64)
        _lib.check(_lib.load().oz_net_profile_kernels(self._h, _lib.p_f64(ms), _lib.p_i64(cnt), 1 if reset else 0))
        return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(_lib.NET_KERNELS)}

    def profile_read(self):
        ms, cnt = C.c_double(), C.c_int64()
        _lib.check(_lib.load().oz_net_profile_read(self._h, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def set_tables(self, mode):
        """f16x2: 2 = conv1 + conv2 from pattern tables (default), 1 = conv1 table + conv2 GEMM, 0 = conv1 kernel + conv2 GEMM"""
        _lib.check(_lib.load().oz_net_set_tables(self._h, int(mode)))

    def set_eval_cache(self, entries):
        """persistent exact-key evaluation cache of this network for up to ~`entries` positions (0 frees it); used by engines created with
        eval_cache=True, emptied whenever the weights change"""
        _lib.check(_lib.load().oz_net_set_eval_cache(self._h, int(entries)))

    def eval_cache_stats(self):
        e, l, h, i = (C.c_int64() for _ in range(4))
        _lib.check(_lib.load().oz_net_eval_cache_stats(self._h, C.byref(e), C.byref(l), C.byref(h), C.byref(i)))
        return dict(entries=e.value, lookups=l.value, hits=h.value, inserts=i.value)

    def set_option(self, option, value):
        """switches (_lib.NET_OPT_*): NET_OPT_SIMPLE_LOOP = the one-barrier conv loop the race screen compares against; precision f16x2, effective
        at the next commit(): NET_OPT_ACT_TARGET_LOG2 / NET_OPT_W_TARGET_LOG2 (calibration / column maxima land below 2^value, default -2),
        NET_OPT_LOW_GUARD_LOG2 (row threshold of the low-side guard, default -17; <= -100 = off), NET_OPT_SELF_CHECK (0 off, 1 enforce, 2 measure);
        precision f32: NET_OPT_F32_STD_TILE = the 3x3 convolutions on the 128 x 128 tile even where the 256 x 256 one would be picked (bit-identical)"""
        _lib.check(_lib.load().oz_net_set_option(self._h, int(option), int(value)))

    def commit(self):
        """oz_net_commit again (options that take effect at commit: NET_OPT_ACT_TARGET_LOG2, NET_OPT_LOW_GUARD_LOG2)"""
        _lib.check(_lib.load().oz_net_commit(self._h))

    def scaling(self, which):
        """precision f16x2: the power-of-two exponents of the last commit -- which 0 .. 4 = per-channel activation exponents of the conv1 .. conv4,
        fc1 outputs, 5 .. 9 = per-column weight exponents of conv2 .. fc2 (oz_net_get_scaling)"""
        C_ = self.num_channels
        size = [C_, C_, C_, C_, 1024, C_, C_, C_, 1024, 512][which]
        out = np.zeros(size, np.int32)
        _lib.check(_lib.load().oz_net_get_scaling(self._h, int(which), _lib.p_i32(out), size))
        return out

    def self_check(self):
        """precision f16x2: what the last commit's self-check measured -- (max |d pi|, max |d v|, positions) between the f16x2 and the
        exact-fp32 kernels on the calibration positions (negative: it did not run); a commit above 8e-6 raises OzError(OZ_ERR_STATE)"""
        a, b, k = C.c_double(), C.c_double(), C.c_int()
        _lib.check(_lib.load().oz_net_self_check(self._h, C.byref(a), C.byref(b), C.byref(k)))
        return a.value, b.value, k.value

    def self_check_guard(self):
        """precision f16x2: the guard bits (1 = above the fp16 range, 4 = a low row) the last commit's self-check saw on the calibration positions
        (0 = none; with NET_OPT_SELF_CHECK = 2 the commit succeeds and this is how a caller learns of them)"""
        v = C.c_int()
        _lib.check(_lib.load().oz_net_get_info(self._h, _lib.NET_INFO_SELF_CHECK_GUARD, C.byref(v)))
        return v.value

    def conv3_tile_rows(self):
        """row-tile height the LAST forward ran conv3 on (f16x2: 256 at bench.py's batch cap, 192 for full 4096-leaf launches, 128 on the latency path;
        f32: 256 = the 256 x 256 tile of large batches, 128 = the standard tile, 64 = the weight-stream kernel of one-position networks)"""
        v = C.c_int()
        _lib.check(_lib.load().oz_net_get_info(self._h, _lib.NET_INFO_CONV3_TILE_ROWS, C.byref(v)))
        return v.value

    def arithmetic(self):
        """the arithmetic the GEMM layers really run in: self.precision, except "f32" for a bf16x3 network of max_batch < 128 (latency kernels)"""
        v = C.c_int()
        _lib.check(_lib.load().oz_net_get_info(self._h, _lib.NET_INFO_ARITHMETIC, C.byref(v)))
        return {0: "f32", 1: "f16x2", 2: "bf16x3"}[v.value]

    def profiled_layer(self):
        """which launch profile_read() timed: 2 = the conv2 GEMM, 3 = the conv3 GEMM (f16x2: conv1 + conv2 are a table gather-sum)"""
        layer = C.c_int()
        _lib.check(_lib.load().oz_net_profiled_layer(self._h, C.byref(layer)))
        return layer.value


class StubNetWrapper(_NetHandle):
    """Device-side deterministic test network (integer hash of the board; formula: oracle/oz_oracle.c
    orc_stub_predict).  Duck-types NNetWrapper for the search / self-play engines."""

    def __init__(self, board_size=(8, 8), salt=0, keep_mask=0, max_batch=1):
        super().__init__()
        self.board_size_x, self.board_size_y = board_size
        self.action_size = self.board_size_x * self.board_size_y
        self.network_type = NeuralNets.ONN
        self.max_batch = int(max_batch)
        self.salt, self.keep_mask = salt, keep_mask
        lib = _lib.require_gpu()
        _lib.check(lib.oz_net_create_stub(C.byref(self._h), self.board_size_x, salt, keep_mask, self.max_batch))

    predict_batch = NNetWrapper.predict_batch
    predict = NNetWrapper.predict
    set_eval_cache = NNetWrapper.set_eval_cache
    eval_cache_stats = NNetWrapper.eval_cache_stats
