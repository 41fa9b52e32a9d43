"""Test helper: a minimal ctypes binding of the REAL libhdf5 (when the image has one) used to
cross-check a-link_amd/hdf5_lite.py — write files the way h5py does for Keras, and read files back.
Not part of the product; tests skip when no libhdf5 can be loaded."""
import ctypes as C
import glob
import os

import numpy as np

hid_t = C.c_int64
_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib or None
    cands = sorted(glob.glob("/opt/conda/lib/libhdf5.so*")) + sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libhdf5*.so*"))
    for p in cands:
        try:
            L = C.CDLL(p)
            L.H5open()
            _lib = L
            break
        except OSError:
            continue
    if _lib is None:
        _lib = False
        return None
    L = _lib
    for name, res, args in [
            ("H5Fcreate", hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), ("H5Fopen", hid_t, [C.c_char_p, C.c_uint, hid_t]),
            ("H5Fclose", C.c_int, [hid_t]), ("H5Gcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
            ("H5Gclose", C.c_int, [hid_t]), ("H5Gopen2", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Screate_simple", hid_t, [C.c_int, C.c_void_p, C.c_void_p]),
            ("H5Screate", hid_t, [C.c_int]), ("H5Sclose", C.c_int, [hid_t]),
            ("H5Dcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            ("H5Dwrite", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            ("H5Dopen2", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Dclose", C.c_int, [hid_t]),
            ("H5Dget_space", hid_t, [hid_t]), ("H5Dread", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            ("H5Sget_simple_extent_ndims", C.c_int, [hid_t]),
            ("H5Sget_simple_extent_dims", C.c_int, [hid_t, C.c_void_p, C.c_void_p]),
            ("H5Sget_simple_extent_npoints", C.c_int64, [hid_t]),
            ("H5Acreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]),
            ("H5Awrite", C.c_int, [hid_t, hid_t, C.c_void_p]), ("H5Aclose", C.c_int, [hid_t]),
            ("H5Aopen_by_name", hid_t, [hid_t, C.c_char_p, C.c_char_p, hid_t, hid_t]),
            ("H5Aget_type", hid_t, [hid_t]), ("H5Aget_space", hid_t, [hid_t]), ("H5Aread", C.c_int, [hid_t, hid_t, C.c_void_p]),
            ("H5Tcopy", hid_t, [hid_t]), ("H5Tset_size", C.c_int, [hid_t, C.c_size_t]), ("H5Tget_size", C.c_size_t, [hid_t]),
            ("H5Tset_strpad", C.c_int, [hid_t, C.c_int]), ("H5Tclose", C.c_int, [hid_t]),
            ("H5Tget_class", C.c_int, [hid_t])]:
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    return L


def _g(name):
    return hid_t.in_dll(lib(), name).value


def _str_type(n):
    L = lib()
    t = L.H5Tcopy(_g("H5T_C_S1_g"))
    L.H5Tset_size(t, max(1, n))
    L.H5Tset_strpad(t, 1)            # H5T_STR_NULLPAD, what h5py uses for numpy 'S' dtypes
    return t


def _write_attr(obj, name, value):
    L = lib()
    a = np.asarray(value)
    if a.dtype.kind == "S":
        t = _str_type(a.dtype.itemsize)
    elif a.dtype == np.float64:
        t = L.H5Tcopy(_g("H5T_IEEE_F64LE_g"))
    else:
        raise TypeError(a.dtype)
    if a.ndim == 0:
        s = L.H5Screate(0)
    else:
        dims = (C.c_uint64 * a.ndim)(*a.shape)
        s = L.H5Screate_simple(a.ndim, dims, None)
    at = L.H5Acreate2(obj, name.encode(), t, s, 0, 0)
    assert at >= 0
    buf = np.ascontiguousarray(a)
    if buf.size:
        assert L.H5Awrite(at, t, buf.ctypes.data) >= 0
    L.H5Aclose(at); L.H5Sclose(s); L.H5Tclose(t)


def _write_vlen_str_attr(obj, name, text):
    """scalar variable-length UTF-8 string attribute — what h5py >= 3 writes for a Python str"""
    L = lib()
    t = L.H5Tcopy(_g("H5T_C_S1_g"))
    L.H5Tset_size(t, C.c_size_t(-1).value)        # H5T_VARIABLE
    L.H5Tset_cset.restype, L.H5Tset_cset.argtypes = C.c_int, [hid_t, C.c_int]
    L.H5Tset_cset(t, 1)                           # H5T_CSET_UTF8
    s = L.H5Screate(0)
    at = L.H5Acreate2(obj, name.encode(), t, s, 0, 0)
    assert at >= 0
    buf = (C.c_char_p * 1)(text.encode("utf8"))
    assert L.H5Awrite(at, t, buf) >= 0
    L.H5Aclose(at); L.H5Sclose(s); L.H5Tclose(t)


def write_keras_like(path, layers, backend=b"tensorflow", keras_version=b"2.1.2", vlen_scalars=False):
    """What keras.engine.topology.save_weights_to_hdf5_group does through h5py, done through libhdf5."""
    L = lib()
    f = L.H5Fcreate(path.encode(), 2, 0, 0)
    assert f >= 0
    _write_attr(f, "layer_names", np.array([n.encode() for n, _ in layers]))
    if vlen_scalars:
        _write_vlen_str_attr(f, "backend", backend.decode())
        _write_vlen_str_attr(f, "keras_version", keras_version.decode())
    else:
        _write_attr(f, "backend", np.bytes_(backend))
        _write_attr(f, "keras_version", np.bytes_(keras_version))
    made_root = set()
    for lname, weights in layers:
        parts = lname.split("/")
        for i in range(1, len(parts)):                # h5py creates the intermediate groups of "a/b/c"
            sub = "/".join(parts[:i])
            if sub not in made_root and not any(sub == n for n, _ in layers if n in made_root):
                sg = L.H5Gcreate2(f, sub.encode(), 0, 0, 0)
                if sg >= 0:
                    L.H5Gclose(sg)
                made_root.add(sub)
        g = L.H5Gcreate2(f, lname.encode(), 0, 0, 0) if lname not in made_root else L.H5Gopen2(f, lname.encode(), 0)
        made_root.add(lname)
        assert g >= 0
        _write_attr(g, "weight_names", np.array([w.encode() for w, _ in weights]) if weights else np.zeros((0,), np.float64))
        made = set()
        for wname, arr in weights:
            parts = wname.split("/")
            for i in range(1, len(parts)):
                sub = "/".join(parts[:i])
                if sub not in made:
                    sg = L.H5Gcreate2(g, sub.encode(), 0, 0, 0)
                    assert sg >= 0
                    L.H5Gclose(sg)
                    made.add(sub)
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            dims = (C.c_uint64 * arr.ndim)(*arr.shape)
            s = L.H5Screate_simple(arr.ndim, dims, None)
            d = L.H5Dcreate2(g, wname.encode(), _g("H5T_IEEE_F32LE_g"), s, 0, 0, 0)
            assert d >= 0
            assert L.H5Dwrite(d, _g("H5T_NATIVE_FLOAT_g"), 0, 0, 0, arr.ctypes.data) >= 0
            L.H5Dclose(d); L.H5Sclose(s)
        L.H5Gclose(g)
    L.H5Fclose(f)


def read_dataset(path, name):
    L = lib()
    f = L.H5Fopen(path.encode(), 0, 0)
    assert f >= 0, "libhdf5 cannot open %s" % path
    d = L.H5Dopen2(f, name.encode(), 0)
    assert d >= 0, "libhdf5 cannot open dataset %s" % name
    s = L.H5Dget_space(d)
    nd = L.H5Sget_simple_extent_ndims(s)
    dims = (C.c_uint64 * max(nd, 1))()
    L.H5Sget_simple_extent_dims(s, dims, None)
    out = np.empty(tuple(dims[:nd]), np.float32)
    assert L.H5Dread(d, _g("H5T_NATIVE_FLOAT_g"), 0, 0, 0, out.ctypes.data) >= 0
    L.H5Sclose(s); L.H5Dclose(d); L.H5Fclose(f)
    return out


def read_string_attr(path, obj, name):
    """Fixed-length string attribute (scalar or 1-d) of object `obj` -> list of bytes."""
    L = lib()
    f = L.H5Fopen(path.encode(), 0, 0)
    assert f >= 0
    a = L.H5Aopen_by_name(f, obj.encode(), name.encode(), 0, 0)
    assert a >= 0, "libhdf5 cannot open attribute %s of %s" % (name, obj)
    t = L.H5Aget_type(a)
    n = L.H5Tget_size(t)
    s = L.H5Aget_space(a)
    cnt = L.H5Sget_simple_extent_npoints(s)
    buf = C.create_string_buffer(int(n * max(cnt, 1)))
    if cnt:
        assert L.H5Aread(a, t, buf) >= 0
    L.H5Sclose(s); L.H5Tclose(t); L.H5Aclose(a); L.H5Fclose(f)
    raw = buf.raw
    return [raw[i * n:(i + 1) * n].rstrip(b"\0") for i in range(cnt)]
