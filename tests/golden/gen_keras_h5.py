"""Makes tests/golden/keras_weights_*.h5 with the REAL HDF5 library (libhdf5 1.10.6 found in this image under
/opt/conda/lib; h5py itself is absent), issuing the calls h5py 2.10 makes for Keras 2.4 / TF 2.3.1
`Model.save_weights(path, save_format='h5')` (the reference's Net/NNet.py:90-92; requirements.txt pins
tensorflow==2.3.1, h5py==2.10.0 -- whose wheel bundles HDF5 1.10.x):

  f.attrs['layer_names'] = [fixed-length byte strings]      f.attrs['backend'] = b'tensorflow'
  f.attrs['keras_version'] = b'2.4.0'
  per layer: g = f.create_group(layer.name); g.attrs['weight_names'] = [b'conv2d/kernel:0', ...]
             g.create_dataset('conv2d/kernel:0', shape, dtype=float32)[:] = value      (intermediate group made by the lcpl)

The files are DATA: they pin othellozero_amd/keras_h5.py's from-scratch reader against the genuine library, and the
expected arrays are regenerated from the same seeds by the tests.  Run in this container only:

    python tests/golden/gen_keras_h5.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
LIBHDF5 = os.environ.get("OZ_LIBHDF5", "/opt/conda/lib/libhdf5.so.103")


class H5:
    """the handful of libhdf5 calls needed, through ctypes (hid_t is int64 in 1.10)"""

    def __init__(self, path=LIBHDF5):
        h = self.h = C.CDLL(path)
        h.H5open()
        hid = C.c_int64
        for name, res, args in [
            ("H5Fcreate", hid, [C.c_char_p, C.c_uint, hid, hid]), ("H5Fopen", hid, [C.c_char_p, C.c_uint, hid]),
            ("H5Fclose", C.c_int, [hid]),
            ("H5Gcreate2", hid, [hid, C.c_char_p, hid, hid, hid]), ("H5Gopen2", hid, [hid, C.c_char_p, hid]), ("H5Gclose", C.c_int, [hid]),
            ("H5Screate", hid, [C.c_int]), ("H5Screate_simple", hid, [C.c_int, C.POINTER(C.c_uint64), C.c_void_p]),
            ("H5Sclose", C.c_int, [hid]), ("H5Sget_simple_extent_ndims", C.c_int, [hid]),
            ("H5Sget_simple_extent_dims", C.c_int, [hid, C.POINTER(C.c_uint64), C.c_void_p]),
            ("H5Tcopy", hid, [hid]), ("H5Tset_size", C.c_int, [hid, C.c_size_t]), ("H5Tset_strpad", C.c_int, [hid, C.c_int]),
            ("H5Tclose", C.c_int, [hid]), ("H5Tget_size", C.c_size_t, [hid]),
            ("H5Acreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid]), ("H5Awrite", C.c_int, [hid, hid, C.c_void_p]),
            ("H5Aopen", hid, [hid, C.c_char_p, hid]), ("H5Aread", C.c_int, [hid, hid, C.c_void_p]),
            ("H5Aget_type", hid, [hid]), ("H5Aget_space", hid, [hid]), ("H5Aclose", C.c_int, [hid]),
            ("H5Dcreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), ("H5Dopen2", hid, [hid, C.c_char_p, hid]),
            ("H5Dwrite", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]), ("H5Dread", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
            ("H5Dget_space", hid, [hid]), ("H5Dclose", C.c_int, [hid]),
            ("H5Pcreate", hid, [hid]), ("H5Pset_create_intermediate_group", C.c_int, [hid, C.c_uint]), ("H5Pclose", C.c_int, [hid]),
        ]:
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        g = lambda s: C.c_int64.in_dll(h, s).value
        self.F32LE, self.NATIVE_FLOAT, self.C_S1 = g("H5T_IEEE_F32LE_g"), g("H5T_NATIVE_FLOAT_g"), g("H5T_C_S1_g")
        self.LCPL = g("H5P_CLS_LINK_CREATE_ID_g")

    def check(self, v, what):
        if v < 0:
            raise RuntimeError(f"libhdf5 call failed: {what}")
        return v

    def space(self, shape):
        if len(shape) == 0:
            return self.check(self.h.H5Screate(0), "H5Screate(scalar)")
        dims = (C.c_uint64 * len(shape))(*shape)
        return self.check(self.h.H5Screate_simple(len(shape), dims, None), "H5Screate_simple")

    def set_attr_bytes(self, loc, name, value):
        """h5py semantics: bytes -> scalar fixed-length string; list of bytes -> 1-D array of 'S<max>' (null padded)"""
        arr = np.asarray(value)
        assert arr.dtype.kind == "S"
        t = self.h.H5Tcopy(self.C_S1)
        self.h.H5Tset_size(t, arr.dtype.itemsize)
        self.h.H5Tset_strpad(t, 1)                       # H5T_STR_NULLPAD (h5py's mapping of numpy 'S')
        s = self.space(arr.shape)
        a = self.check(self.h.H5Acreate2(loc, name.encode(), t, s, 0, 0), "H5Acreate2")
        buf = np.ascontiguousarray(arr)
        self.check(self.h.H5Awrite(a, t, buf.ctypes.data_as(C.c_void_p)), "H5Awrite")
        self.h.H5Aclose(a), self.h.H5Sclose(s), self.h.H5Tclose(t)

    def get_attr_bytes(self, loc, name):
        a = self.check(self.h.H5Aopen(loc, name.encode(), 0), "H5Aopen " + name)
        t, s = self.h.H5Aget_type(a), self.h.H5Aget_space(a)
        size, nd = self.h.H5Tget_size(t), self.h.H5Sget_simple_extent_ndims(s)
        dims = (C.c_uint64 * max(nd, 1))()
        if nd:
            self.h.H5Sget_simple_extent_dims(s, dims, None)
        out = np.zeros(tuple(dims[:nd]), dtype=f"S{size}")
        self.check(self.h.H5Aread(a, t, out.ctypes.data_as(C.c_void_p)), "H5Aread")
        self.h.H5Tclose(t), self.h.H5Sclose(s), self.h.H5Aclose(a)
        return out

    def read_dataset_f32(self, loc, name):
        d = self.check(self.h.H5Dopen2(loc, name.encode(), 0), "H5Dopen2 " + name)
        s = self.h.H5Dget_space(d)
        nd = self.h.H5Sget_simple_extent_ndims(s)
        dims = (C.c_uint64 * max(nd, 1))()
        if nd:
            self.h.H5Sget_simple_extent_dims(s, dims, None)
        out = np.zeros(tuple(dims[:nd]), dtype=np.float32)
        self.check(self.h.H5Dread(d, self.NATIVE_FLOAT, 0, 0, 0, out.ctypes.data_as(C.c_void_p)), "H5Dread")
        self.h.H5Sclose(s), self.h.H5Dclose(d)
        return out


def save_weights_like_keras(h5, path, layers, keras_version=b"2.4.0", backend=b"tensorflow"):
    """layers = [(layer_name, [(weight_name, ndarray float32), ...]), ...] in model.layers order (weightless layers too).
    Same call sequence as keras hdf5_format.save_weights_to_hdf5_group under h5py."""
    h = h5.h
    f = h5.check(h.H5Fcreate(path.encode(), 2, 0, 0), "H5Fcreate")             # H5F_ACC_TRUNC, default fcpl/fapl
    lcpl = h.H5Pcreate(h5.LCPL)
    h.H5Pset_create_intermediate_group(lcpl, 1)
    h5.set_attr_bytes(f, "layer_names", [n.encode() for n, _ in layers])
    h5.set_attr_bytes(f, "backend", backend)
    h5.set_attr_bytes(f, "keras_version", keras_version)
    for name, weights in layers:
        g = h5.check(h.H5Gcreate2(f, name.encode(), lcpl, 0, 0), "H5Gcreate2")
        if weights:
            h5.set_attr_bytes(g, "weight_names", [wn.encode() for wn, _ in weights])
        else:                                                                    # np.asarray([]) is float64 under h5py
            s = h5.space((0,))
            a = h.H5Acreate2(g, b"weight_names", C.c_int64.in_dll(h, "H5T_IEEE_F64LE_g").value, s, 0, 0)
            h.H5Aclose(a), h.H5Sclose(s)
        for wn, val in weights:
            val = np.ascontiguousarray(val, dtype=np.float32)
            s = h5.space(val.shape)
            d = h5.check(h.H5Dcreate2(g, wn.encode(), h5.F32LE, s, lcpl, 0, 0), "H5Dcreate2")
            h5.check(h.H5Dwrite(d, h5.NATIVE_FLOAT, 0, 0, 0, val.ctypes.data_as(C.c_void_p)), "H5Dwrite")
            h.H5Dclose(d), h.H5Sclose(s)
        h.H5Gclose(g)
    h.H5Pclose(lcpl)
    h.H5Fclose(f)


def save_small_latest(h5, path, layers):
    """Same layout written with libver='latest' (superblock v3, v2 object headers, compact link messages) plus a
    variable-length UTF-8 string attribute (what h5py stores for a Python str) -- exercises the reader's other branches."""
    h = h5.h
    h.H5Pset_libver_bounds.argtypes = [C.c_int64, C.c_int, C.c_int]
    fapl = h.H5Pcreate(C.c_int64.in_dll(h, "H5P_CLS_FILE_ACCESS_ID_g").value)
    h5.check(h.H5Pset_libver_bounds(fapl, 2, 2), "H5Pset_libver_bounds")        # H5F_LIBVER_V110 = latest in 1.10
    f = h5.check(h.H5Fcreate(path.encode(), 2, 0, fapl), "H5Fcreate")
    lcpl = h.H5Pcreate(h5.LCPL)
    h.H5Pset_create_intermediate_group(lcpl, 1)
    h5.set_attr_bytes(f, "layer_names", [n.encode() for n, _ in layers])
    h5.set_attr_bytes(f, "backend", b"tensorflow")
    t = h.H5Tcopy(h5.C_S1)
    h.H5Tset_size(t, C.c_size_t(-1).value)                                      # H5T_VARIABLE
    h.H5Tset_cset.argtypes = [C.c_int64, C.c_int]
    h.H5Tset_cset(t, 1)                                                          # UTF-8
    s = h5.space(())
    a = h5.check(h.H5Acreate2(f, b"keras_version", t, s, 0, 0), "H5Acreate2 vlen")
    ptr = (C.c_char_p * 1)(b"2.4.0")
    h5.check(h.H5Awrite(a, t, ptr), "H5Awrite vlen")
    h.H5Aclose(a), h.H5Sclose(s), h.H5Tclose(t)
    for name, weights in layers:
        g = h5.check(h.H5Gcreate2(f, name.encode(), lcpl, 0, 0), "H5Gcreate2")
        h5.set_attr_bytes(g, "weight_names", [wn.encode() for wn, _ in weights])
        for wn, val in weights:
            val = np.ascontiguousarray(val, dtype=np.float32)
            s = h5.space(val.shape)
            d = h5.check(h.H5Dcreate2(g, wn.encode(), h5.F32LE, s, lcpl, 0, 0), "H5Dcreate2")
            h5.check(h.H5Dwrite(d, h5.NATIVE_FLOAT, 0, 0, 0, val.ctypes.data_as(C.c_void_p)), "H5Dwrite")
            h.H5Dclose(d), h.H5Sclose(s)
        h.H5Gclose(g)
    h.H5Pclose(lcpl), h.H5Pclose(fapl)
    h.H5Fclose(f)


def small_layers(seed=11):
    rs = np.random.RandomState(seed)
    r = lambda *s: rs.standard_normal(s).astype(np.float32)
    return [("dense", [("dense/kernel:0", r(5, 7)), ("dense/bias:0", r(7))]),
            ("pi", [("pi/kernel:0", r(7, 3)), ("pi/bias:0", r(3))]),
            ("v", [("v/kernel:0", r(7, 1)), ("v/bias:0", r(1))])]


def tiny_weights(n, C, D1, D2, cin, seed):
    """40 arrays with OthelloNN's structure but small dense widths (the file format does not care), all random"""
    rs = np.random.RandomState(seed)
    shapes = []
    for ci in (cin, C, C, C):
        shapes += [(3, 3, ci, C)] + [(C,)] * 5
    shapes += [((n - 4) * (n - 4) * C, D1)] + [(D1,)] * 5 + [(D1, D2)] + [(D2,)] * 5 + [(D2, n * n), (n * n,), (D2, 1), (1,)]
    return [rs.standard_normal(s).astype(np.float32) for s in shapes]


FIXTURES = (("onn6", 6, 8, 24, 12, 2, 5, 0), ("bnn6", 6, 8, 24, 12, 1, 6, 0), ("onn8_third_model", 8, 4, 16, 8, 2, 7, 2))

if __name__ == "__main__":
    from othellozero_amd.keras_h5 import keras_layer_table
    h5 = H5()
    for tag, n, ch, d1, d2, cin, seed, index in FIXTURES:
        layers = keras_layer_table(tiny_weights(n, ch, d1, d2, cin, seed), model_index=index, network="BNN" if cin == 1 else "ONN")
        out = os.path.join(HERE, f"keras_weights_{tag}.h5")
        save_weights_like_keras(h5, out, layers)
        print(out, os.path.getsize(out), "bytes")
    out = os.path.join(HERE, "keras_weights_small_libver_latest.h5")
    save_small_latest(h5, out, small_layers())
    print(out, os.path.getsize(out), "bytes")
