"""CPU: checkpoint formats either side of the hot path — MXNet .params / -symbol.json
(a-link_amd/mxnet_format.py) and Keras .h5 weight files (a-link_amd/hdf5_lite.py).  The HDF5 code is
cross-checked against the real libhdf5 when the image has one (tests/h5ref.py)."""
import os
import struct

import numpy as np
import pytest

import a_link_amd  # noqa: F401
from a_link_amd import hdf5_lite as H5
from a_link_amd import mxnet_format as MX
from a_link_amd import weights as W

import h5ref


def _head_layers(seed, d=512):
    rng = np.random.RandomState(seed)
    mk = lambda *s: rng.randn(*s).astype(np.float32)
    return [("input_1", []), ("input_2", []), ("lambda_1", []),
            ("dense_1", [("dense_1/kernel:0", mk(d, 512)), ("dense_1/bias:0", mk(512))]),
            ("dense_2", [("dense_2/kernel:0", mk(512, 64)), ("dense_2/bias:0", mk(64))]),
            ("dense_3", [("dense_3/kernel:0", mk(64, 2)), ("dense_3/bias:0", mk(2))]),
            ("activation_1", [])]


def _same(a, b):
    assert [n for n, _ in a] == [n for n, _ in b]
    for (_, wa), (_, wb) in zip(a, b):
        assert [n for n, _ in wa] == [n for n, _ in wb]
        for (_, x), (_, y) in zip(wa, wb):
            assert x.dtype == np.float32 and x.shape == y.shape and np.array_equal(x, y)


def test_h5_roundtrip_own_writer(tmp_path):
    layers = _head_layers(0)
    p = str(tmp_path / "m.h5")
    H5.save_keras_weights(p, layers)
    _same(H5.load_keras_weights(p), layers)
    f = H5.File(p)
    assert f.attrs["backend"] == b"tensorflow" and f.attrs["keras_version"] == b"2.1.2"
    assert sorted(f.keys()) == sorted(n for n, _ in layers)
    assert f["dense_1"]["dense_1"]["kernel:0"].shape == (512, 512)
    assert "nope" not in f and "dense_2/dense_2/bias:0" in f
    with pytest.raises(KeyError):
        f["dense_1/missing"]


def test_h5_many_layers_and_names(tmp_path):
    rng = np.random.RandomState(1)
    layers = [("conv2d_%d" % i, [("conv2d_%d/kernel:0" % i, rng.randn(3, 3, 2, 4).astype(np.float32)),
                                 ("conv2d_%d/bias:0" % i, rng.randn(4).astype(np.float32))]) for i in range(1, 40)]
    layers.insert(3, ("dropout_1", []))
    p = str(tmp_path / "many.h5")
    H5.save_keras_weights(p, layers)
    _same(H5.load_keras_weights(p), layers)


@pytest.mark.skipif(h5ref.lib() is None, reason="no libhdf5 in this image")
def test_h5_reader_on_files_written_by_libhdf5(tmp_path):
    for seed, d in ((2, 512), (3, 2048)):
        layers = _head_layers(seed, d)
        p = str(tmp_path / ("ref%d.h5" % seed))
        h5ref.write_keras_like(p, layers)
        _same(H5.load_keras_weights(p), layers)
        f = H5.File(p)
        assert f.attrs["backend"] == b"tensorflow"
        assert [bytes(x) for x in f.attrs["layer_names"]] == [n.encode() for n, _ in layers]


@pytest.mark.skipif(h5ref.lib() is None, reason="no libhdf5 in this image")
def test_h5_reader_multi_level_group_btree(tmp_path):
    """300 layer groups in the root: libhdf5 splits the symbol table over many leaf nodes and a
    two-level B-tree (a keras-vggface ResNet-50 weight file has ~110 root groups)."""
    rng = np.random.RandomState(0)
    layers = [("layer_%03d" % i, [("layer_%03d/gamma:0" % i, rng.randn(5).astype(np.float32))]) for i in range(300)]
    p = str(tmp_path / "big.h5")
    h5ref.write_keras_like(p, layers)
    _same(H5.load_keras_weights(p), layers)
    assert len(H5.File(p).keys()) == 300
    # nested layer names as keras-vggface has them ("conv1/7x7_s2" and "conv1/7x7_s2/bn" share a prefix)
    nested = [("conv1/7x7_s2", [("conv1/7x7_s2/kernel:0", rng.randn(7, 7, 3, 4).astype(np.float32))]),
              ("conv1/7x7_s2/bn", [("conv1/7x7_s2/bn/gamma:0", rng.randn(4).astype(np.float32)),
                                   ("conv1/7x7_s2/bn/beta:0", rng.randn(4).astype(np.float32))])]
    q = str(tmp_path / "nested.h5")
    h5ref.write_keras_like(q, nested)
    _same(H5.load_keras_weights(q), nested)


@pytest.mark.skipif(h5ref.lib() is None, reason="no libhdf5 in this image")
def test_h5_reader_variable_length_string_attributes(tmp_path):
    """h5py >= 3 stores Python-str attributes (backend, keras_version) as variable-length UTF-8 strings
    in the global heap; layer/weight name arrays stay fixed-length."""
    layers = _head_layers(6)
    p = str(tmp_path / "vlen.h5")
    h5ref.write_keras_like(p, layers, keras_version=b"2.4.3", vlen_scalars=True)
    f = H5.File(p)
    assert f.attrs["backend"] == b"tensorflow" and f.attrs["keras_version"] == b"2.4.3"
    _same(H5.load_keras_weights(p), layers)


@pytest.mark.skipif(h5ref.lib() is None, reason="no libhdf5 in this image")
def test_h5_writer_output_is_readable_by_libhdf5(tmp_path):
    layers = _head_layers(4)
    p = str(tmp_path / "ours.h5")
    H5.save_keras_weights(p, layers)
    assert h5ref.read_string_attr(p, "/", "layer_names") == [n.encode() for n, _ in layers]
    assert h5ref.read_string_attr(p, "/", "backend") == [b"tensorflow"]
    assert h5ref.read_string_attr(p, "/dense_2", "weight_names") == [b"dense_2/kernel:0", b"dense_2/bias:0"]
    for lname, ws in layers:
        for wn, arr in ws:
            assert np.array_equal(h5ref.read_dataset(p, "/%s/%s" % (lname, wn)), arr)


def test_h5_rejects_garbage(tmp_path):
    p = str(tmp_path / "bad.h5")
    open(p, "wb").write(b"not hdf5 at all" * 10)
    with pytest.raises(H5.H5Error):
        H5.File(p)
    q = str(tmp_path / "trunc.h5")
    H5.save_keras_weights(q, _head_layers(5))
    data = open(q, "rb").read()
    open(q, "wb").write(data[:len(data) // 3])
    with pytest.raises((H5.H5Error, KeyError)):
        H5.load_keras_weights(q)


# ---- MXNet ------------------------------------------------------------------------------------------
def test_params_roundtrip_and_legacy_layouts(tmp_path):
    params = W.synthetic_ir_params((1, 1, 1, 1), size=(16, 16), seed=2)
    prefix = str(tmp_path / "model")
    MX.save_checkpoint(prefix, 0, params)
    sym, arg, aux = MX.load_checkpoint(prefix, 0)
    assert set(aux) == {k for k in params if k.endswith("_moving_mean") or k.endswith("_moving_var")}
    for k, v in params.items():
        got = aux[k] if k in aux else arg[k]
        assert got.dtype == np.float32 and np.array_equal(got, v)
    # V1 and pre-1.0 records decode to the same array
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    v1 = struct.pack("<II2q", MX.V1_MAGIC, 2, 2, 3) + struct.pack("<iii", 1, 0, 0) + a.tobytes()
    old = struct.pack("<I2I", 2, 2, 3) + struct.pack("<iii", 1, 0, 0) + a.tobytes()
    for rec in (v1, old):
        p = str(tmp_path / "x.params")
        name = b"arg:w"
        open(p, "wb").write(struct.pack("<QQQ", MX.LIST_MAGIC, 0, 1) + rec + struct.pack("<QQ", 1, len(name)) + name)
        assert np.array_equal(MX.load_ndarray_file(p)["arg:w"], a)
    with pytest.raises(ValueError):
        open(p, "wb").write(b"\0" * 64)
        MX.load_ndarray_file(p)


@pytest.mark.parametrize("arch", ["r100", "r50", "r18"])
def test_symbol_describes_the_architecture(tmp_path, arch):
    units = W.ARCH_UNITS[arch]
    path = str(tmp_path / "m-symbol.json")
    sym = MX.write_ir_symbol(path, units)
    cfg = MX.ir_config_from_symbol(MX.load_symbol(path))
    assert cfg["units"] == tuple(units) and cfg["widths"] == W.WIDTHS and cfg["emb"] == 512
    assert cfg["bn_eps"] == 2e-5 and cfg["fix_gamma"] == ["fc1"]
    # every learnable tensor the C library expects is an argument of the graph, and nothing else
    args = {sym["nodes"][i]["name"] for i in sym["arg_nodes"]} - {"data"}
    assert args == set(W.tensor_shapes(units))


def test_symbol_checks(tmp_path):
    path = str(tmp_path / "m-symbol.json")
    sym = MX.write_ir_symbol(path, (1, 1, 1, 1))
    with pytest.raises(ValueError):
        MX.ir_config_from_symbol(sym, "fc7_output")
    bad = MX.load_symbol(path)
    [n for n in bad["nodes"] if n["name"] == "stage2_unit1_conv2"][0]["attrs"]["stride"] = "(1, 1)"
    with pytest.raises(ValueError):
        MX.ir_config_from_symbol(bad)
    bad = MX.load_symbol(path)
    [n for n in bad["nodes"] if n["name"] == "_mulscalar0"][0]["attrs"]["scalar"] = "1.0"
    with pytest.raises(ValueError):
        MX.ir_config_from_symbol(bad)
    old = MX.load_symbol(path)                      # MXNet < 1.0 called the attribute dict "param"/"attr"
    for n in old["nodes"]:
        if "attrs" in n:
            n["attr"] = n.pop("attrs")
    assert MX.ir_config_from_symbol(old)["units"] == (1, 1, 1, 1)


def test_resolve_model_prefers_mxnet_pair(tmp_path):
    params = W.synthetic_ir_params((1, 1, 1, 1), size=(16, 16), seed=5)
    prefix = str(tmp_path / "model")
    MX.save_checkpoint(prefix, 0, params)
    got, cfg = W.resolve_model_config(prefix + ",0", (16, 16))
    assert cfg["bn_eps"] == 2e-5 and tuple(cfg["widths"]) == W.WIDTHS
    assert set(got) == set(params) and all(np.array_equal(got[k], params[k]) for k in params)


def test_reader_on_independently_assembled_checkpoint():
    """tests/golden/mxnet_tiny-*: a checkpoint assembled byte by byte from the published MXNet layout by
    tests/golden/make_mxnet_fixture.py, which shares no code or constants with mxnet_format.py — so this is a
    check of the READER (the round-trip tests above only compare the module with itself).  The file carries
    what a real export does and the module's own writer does not: arrays saved from gpu(3), a float16 and a
    float64 tensor, a classification head (fc7 + SoftmaxOutput) after the fc1 cut, `workspace` attributes."""
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    prefix = os.path.join(gold, "mxnet_tiny")
    raw = open(prefix + "-0000.params", "rb").read()
    assert raw[:8] == (0x112).to_bytes(8, "little") and raw[24:28] == bytes.fromhex("c9fa93f9")
    sym, arg, aux = MX.load_checkpoint(prefix, 0)
    with np.load(os.path.join(gold, "mxnet_tiny_expected.npz")) as exp:
        assert set(exp.files) == set(arg) | set(aux) and not set(arg) & set(aux)
        for k in exp.files:
            got = aux[k] if k in aux else arg[k]
            assert (k in aux) == (k.endswith("_moving_mean") or k.endswith("_moving_var"))
            assert got.dtype == exp[k].dtype and got.shape == exp[k].shape and np.array_equal(got, exp[k])
        assert arg["stage3_unit1_conv1_weight"].dtype == np.float16 and arg["fc1_beta"].dtype == np.float64
    cfg = MX.ir_config_from_symbol(sym, "fc1_output")              # reference code/face_model.py:35-36
    assert cfg == {"units": (1, 2, 1, 1), "widths": (8, 8, 16, 16, 24), "emb": 12, "bn_eps": 2e-5, "fix_gamma": ["fc1"]}
    params, rcfg = W.resolve_model_config(prefix + ",0", (16, 16))
    want = W.tensor_shapes(cfg["units"], cfg["widths"], (16, 16), cfg["emb"])
    assert set(want) <= set(params) and set(params) - set(want) == {"fc7_weight"}
    assert all(params[k].shape == tuple(s) and params[k].dtype == np.float32 for k, s in want.items())
    # and the forward pass on it runs through the oracle (the graph the file describes is a working network)
    from oracle import ir_resnet
    e = ir_resnet.embed(params, np.random.default_rng(0).integers(0, 256, (2, 16, 16, 3)).astype(np.float32))
    assert e.shape == (2, 12) and np.allclose(np.linalg.norm(e, axis=1), 1, atol=1e-6)
