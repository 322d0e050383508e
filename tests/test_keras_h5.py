"""Keras HDF5 weight files (reference Net/NNet.py:90-101).

CPU tests: the from-scratch reader against files made by the GENUINE libhdf5 1.10.6 with h5py/Keras' call sequence
(tests/golden/keras_weights_*.h5, generator tests/golden/gen_keras_h5.py), and the from-scratch writer read back
through the genuine library / its h5diff tool when this image has them (skipped otherwise -- the fixtures still pin
the reader, and writer -> reader round trips always run).
GPU tests: NNetWrapper.save_checkpoint / load_checkpoint / copy through the C ABI.
"""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
sys.path.insert(0, GOLDEN)

from othellozero_amd import keras_h5 as K          # noqa: E402
import gen_keras_h5 as G                            # noqa: E402  (only its pure-numpy helpers are used unless libhdf5 exists)

H5DIFF = shutil.which("h5diff") or ("/opt/conda/bin/h5diff" if os.path.exists("/opt/conda/bin/h5diff") else None)
HAVE_LIBHDF5 = os.path.exists(G.LIBHDF5)


def _same(layers, want):
    assert [l for l, _ in layers] == [l for l, _ in want]
    for (_, ws), (_, ws2) in zip(layers, want):
        assert [n for n, _ in ws] == [n for n, _ in ws2]
        for (_, a), (_, b) in zip(ws, ws2):
            assert a.dtype == np.float32 and a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("fixture", G.FIXTURES, ids=[f[0] for f in G.FIXTURES])
def test_reader_against_genuine_libhdf5_files(fixture):
    tag, n, ch, d1, d2, cin, seed, index = fixture
    path = os.path.join(GOLDEN, f"keras_weights_{tag}.h5")
    want = K.keras_layer_table(G.tiny_weights(n, ch, d1, d2, cin, seed), index, "BNN" if cin == 1 else "ONN")
    f = K.H5File(path)
    assert bytes(f.attrs["backend"]) == b"tensorflow" and bytes(f.attrs["keras_version"]) == b"2.4.0"
    assert sorted(f.keys()) == sorted(l for l, _ in want)
    layers = K.load_keras_weights(path)
    _same(layers, want)
    flat = K.flat_weights(layers)
    assert len(flat) == 40 and flat[0].shape == (3, 3, cin, ch) and flat[39].shape == (1,)


def test_layer_names_follow_keras_numbering():
    w = G.tiny_weights(6, 4, 8, 4, 2, 0)
    first = [l for l, _ in K.keras_layer_table(w, 0)]
    assert first[:4] == ["input_1", "conv2d", "batch_normalization", "activation"]
    assert first[-3:] == ["pi", "pi-reshaped", "v"] and "flatten" in first and "dropout_1" in first and len(first) == 25
    third = [l for l, _ in K.keras_layer_table(w, 2)]
    assert third[:3] == ["input_3", "conv2d_8", "batch_normalization_12"] and "dense_5" in third and "flatten_2" in third
    bnn = [l for l, _ in K.keras_layer_table(G.tiny_weights(6, 4, 8, 4, 1, 0), 0, "BNN")]
    assert bnn[:3] == ["input_1", "reshape", "conv2d"] and bnn[-3:] == ["pi", "reshape_1", "v"]
    names = [n for _, ws in K.keras_layer_table(w, 0) for n, _ in ws]
    assert names[:6] == ["conv2d/kernel:0", "conv2d/bias:0", "batch_normalization/gamma:0", "batch_normalization/beta:0",
                         "batch_normalization/moving_mean:0", "batch_normalization/moving_variance:0"]
    assert names[-4:] == ["pi/kernel:0", "pi/bias:0", "v/kernel:0", "v/bias:0"]


def test_reader_libver_latest_and_vlen_string():
    """superblock v3, v2 object headers, compact link messages, variable-length string attribute (global heap)"""
    path = os.path.join(GOLDEN, "keras_weights_small_libver_latest.h5")
    f = K.H5File(path)
    assert f.buf[8] == 3 and f.attrs["keras_version"] == b"2.4.0"
    _same(K.load_keras_weights(path), G.small_layers())


def test_writer_reader_round_trip(tmp_path):
    for cin, net in ((2, "ONN"), (1, "BNN")):
        w = G.tiny_weights(8, 8, 32, 16, cin, 3)
        layers = K.keras_layer_table(w, 1, net)
        p = str(tmp_path / f"rt_{net}.h5")
        K.save_keras_weights(p, layers)
        _same(K.load_keras_weights(p), layers)
        assert all(np.array_equal(a, b) for a, b in zip(K.flat_weights(K.load_keras_weights(p)), w))


def test_writer_handles_scalars_empty_groups_and_many_layers(tmp_path):
    rs = np.random.RandomState(0)
    layers = [(f"layer_with_a_rather_long_name_{i:04d}_" + "x" * 300, []) for i in range(200)]
    layers[7] = ("scalar_holder", [("scalar_holder/s:0", np.float32(2.5)), ("scalar_holder/deep/er/t:0", rs.rand(2, 3, 4).astype(np.float32))])
    p = str(tmp_path / "many.h5")
    K.save_keras_weights(p, layers)
    f = K.H5File(p)
    assert "layer_names" not in f.attrs and "layer_names0" in f.attrs and "layer_names1" in f.attrs      # keras' 64512-byte split
    got = K.load_keras_weights(p)
    assert [l for l, _ in got] == [l for l, _ in layers]
    assert got[7][1][0][1].shape == () and got[7][1][0][1] == np.float32(2.5)
    assert np.array_equal(got[7][1][1][1], layers[7][1][1][1])


def test_reader_rejects_garbage(tmp_path):
    p = tmp_path / "bad.h5"
    p.write_bytes(b"this is not an hdf5 file" * 10)
    with pytest.raises(K.H5FormatError):
        K.H5File(str(p))
    good = open(os.path.join(GOLDEN, "keras_weights_onn6.h5"), "rb").read()
    q = tmp_path / "trunc.h5"
    q.write_bytes(good[:len(good) // 3])
    with pytest.raises((K.H5FormatError, KeyError, ValueError)):
        K.load_keras_weights(str(q))


@pytest.mark.skipif(not HAVE_LIBHDF5, reason="genuine libhdf5 not in this image")
def test_writer_output_is_read_by_genuine_libhdf5(tmp_path):
    w = G.tiny_weights(6, 8, 24, 12, 2, 5)
    layers = K.keras_layer_table(w, 0)
    p = str(tmp_path / "mine.h5")
    K.save_keras_weights(p, layers)
    h5 = G.H5()
    f = h5.check(h5.h.H5Fopen(p.encode(), 0, 0), "H5Fopen")
    assert [x.decode() for x in h5.get_attr_bytes(f, "layer_names")] == [l for l, _ in layers]
    assert bytes(h5.get_attr_bytes(f, "backend")) == b"tensorflow"
    for lname, ws in layers:
        g = h5.check(h5.h.H5Gopen2(f, lname.encode(), 0), "H5Gopen2")
        if ws:
            assert [x.decode() for x in h5.get_attr_bytes(g, "weight_names")] == [n for n, _ in ws]
        for nme, arr in ws:
            got = h5.read_dataset_f32(g, nme)
            assert got.shape == arr.shape and np.array_equal(got, arr)
        h5.h.H5Gclose(g)
    h5.h.H5Fclose(f)


@pytest.mark.skipif(H5DIFF is None, reason="h5diff tool not in this image")
def test_writer_output_equals_genuine_file_under_h5diff(tmp_path):
    """objects, attributes and data identical to the libhdf5-written fixture according to libhdf5's own diff tool"""
    tag, n, ch, d1, d2, cin, seed, index = G.FIXTURES[0]
    p = str(tmp_path / "mine.h5")
    K.save_keras_weights(p, K.keras_layer_table(G.tiny_weights(n, ch, d1, d2, cin, seed), index))
    r = subprocess.run([H5DIFF, "-c", p, os.path.join(GOLDEN, f"keras_weights_{tag}.h5")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


# ---------------------------------------------------------------------------------------------------- GPU: the wrapper
@pytest.mark.gpu
def test_checkpoint_round_trip_and_copy(tmp_path):
    from othellozero_amd.NNet import NNetWrapper, NeuralNets
    from othellozero_amd.weights import init_weights
    rs = np.random.RandomState(1)
    own = rs.randint(0, 2**62, size=16, dtype=np.uint64)
    opp = rs.randint(0, 2**62, size=16, dtype=np.uint64) & ~own
    for network, cin in ((NeuralNets.ONN, 2), (NeuralNets.BNN, 1)):
        a = NNetWrapper((8, 8), num_channels_1=256, network=network, max_batch=16,
                        weights=init_weights(8, seed=9, channels=256, randomize_all=True, in_channels=cin))
        p = str(tmp_path / f"ckpt_{network.name}.h5")
        a.save_checkpoint(p)
        b = NNetWrapper((8, 8), num_channels_1=256, network=network, max_batch=16, seed=123)
        pa, va = a.predict_batch(own, opp)
        pb, vb = b.predict_batch(own, opp)
        assert not np.array_equal(pa, pb)
        b.load_checkpoint(p)
        pb, vb = b.predict_batch(own, opp)
        assert np.array_equal(pa, pb) and np.array_equal(va, vb)                  # same weights, same kernels: bit-identical
        names = [l for l, _ in K.load_keras_weights(p)]
        assert any(x.startswith("conv2d") for x in names) and names[-1] == "v"
        c = a.copy()                                                              # Net/NNet.py:98-101
        pc, vc = c.predict_batch(own, opp)
        assert np.array_equal(pa, pc) and np.array_equal(va, vc)
        with pytest.raises(AssertionError):
            b.load_checkpoint(str(tmp_path / "weights.hdf5"))                     # Net/NNet.py:95
    wrong = NNetWrapper((6, 6), num_channels_1=256, max_batch=1)
    with pytest.raises(ValueError):
        wrong.load_checkpoint(str(tmp_path / "ckpt_ONN.h5"))                      # 8x8 file into a 6x6 network
