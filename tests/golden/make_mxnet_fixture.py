#!/usr/bin/env python3
"""Assembles tests/golden/mxnet_tiny-{symbol.json,0000.params} + mxnet_tiny_expected.npz byte by byte from
the PUBLISHED MXNet serialisation rules, WITHOUT importing a-link_amd/mxnet_format.py (neither its writer
nor its constants), so that tests/test_formats.py::test_reader_on_independently_assembled_checkpoint is a
check of the reader and not a round trip through the same author's writer.

What the reference loads (reference code/face_model.py:34, code/arcface_prepreq.sh:13-20) is the pair
`mx.model.save_checkpoint` writes.  Rules restated here (MXNet 1.x sources, not vendored by the reference):

  src/ndarray/ndarray.cc  NDArray::Save(fo, data, names)      uint64 0x112 | uint64 0 | vector<NDArray> | vector<string>
  dmlc-core serializer                                         vector<T>: uint64 n then n x T; string: uint64 len + bytes
  src/ndarray/ndarray.cc  NDArray::Save(strm)                  uint32 0xF993FAC9 | int32 stype(0 = default)
                                                               | TShape | Context | int32 type_flag | raw data
  include/mxnet/tuple.h   TShape::Save                         uint32 ndim | int64 dim[ndim]
  include/mxnet/base.h    Context::Save                        int32 dev_type (1 cpu, 2 gpu) | int32 dev_id
  mshadow type flags                                           0 f32, 1 f64, 2 f16, 3 u8, 4 i32, 5 i8, 6 i64
  python/mxnet/model.py   save_checkpoint                      names "arg:<n>" / "aux:<n>"; symbol.save -> nnvm JSON
  nnvm JSON (MXNet >= 1.0)                                     {"nodes":[{"op","name","attrs"?,"inputs":[[id,idx,ver]]}],
                                                               "arg_nodes","node_row_ptr","heads","attrs":{"mxnet_version":["int",N]}}

The checkpoint is a small LResNet-E-IR (fresnet naming): units (1,2,1,1), widths 8/8/16/16/24, 16x16 input,
embedding 12.  Arrays were "saved from gpu(3)" (the context written is the array's own) and one tensor is float16
and one float64 to exercise the type flags.  Run from the repo root:  python tests/golden/make_mxnet_fixture.py
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
UNITS = (1, 2, 1, 1)
WIDTHS = (8, 8, 16, 16, 24)
EMB = 12
SIZE = 16


def u32(v):
    return int(v).to_bytes(4, "little", signed=False)


def i32(v):
    return int(v).to_bytes(4, "little", signed=True)


def u64(v):
    return int(v).to_bytes(8, "little", signed=False)


def i64(v):
    return int(v).to_bytes(8, "little", signed=True)


TYPE_FLAG = {"float32": 0, "float64": 1, "float16": 2, "uint8": 3, "int32": 4, "int8": 5, "int64": 6}


def ndarray_record(a, dev_type, dev_id):
    out = [u32(0xF993FAC9), i32(0)]                      # V2 magic, kDefaultStorage
    out.append(u32(a.ndim))
    out.extend(i64(d) for d in a.shape)
    out.append(i32(dev_type) + i32(dev_id))
    out.append(i32(TYPE_FLAG[a.dtype.name]))
    out.append(np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<"), copy=False).tobytes(order="C"))
    return b"".join(out)


def build_graph():
    """Returns (nodes, arg_nodes, heads, tensors) where tensors: ordered list of (name, shape, is_aux)."""
    nodes, arg_nodes, tensors = [], [], []

    def variable(name, shape=None, aux=False):
        nodes.append({"op": "null", "name": name, "inputs": []})
        arg_nodes.append(len(nodes) - 1)
        if shape is not None:
            tensors.append((name, tuple(shape), aux))
        return len(nodes) - 1

    def node(op, name, attrs, inputs):
        n = {"op": op, "name": name}
        if attrs:
            n["attrs"] = attrs
        n["inputs"] = [[i, 0, 0] for i in inputs]
        nodes.append(n)
        return len(nodes) - 1

    def batchnorm(name, x, c, fix_gamma):
        g = variable(name + "_gamma", (c,))
        b = variable(name + "_beta", (c,))
        m = variable(name + "_moving_mean", (c,), aux=True)
        v = variable(name + "_moving_var", (c,), aux=True)
        return node("BatchNorm", name, {"eps": "2e-05", "fix_gamma": "True" if fix_gamma else "False", "momentum": "0.9"},
                    [x, g, b, m, v])

    def convolution(name, x, cin, cout, k, stride, pad):
        w = variable(name + "_weight", (cout, cin, k, k))
        return node("Convolution", name, {"kernel": "(%d, %d)" % (k, k), "no_bias": "True", "num_filter": str(cout),
                                          "pad": "(%d, %d)" % (pad, pad), "stride": "(%d, %d)" % (stride, stride),
                                          "workspace": "256"}, [x, w])

    def prelu(name, x, c):
        g = variable(name + "_gamma", (c,))
        return node("LeakyReLU", name, {"act_type": "prelu"}, [x, g])

    x = variable("data")
    x = node("_copy", "id", None, [x])
    x = node("_minus_scalar", "_minusscalar0", {"scalar": "127.5"}, [x])
    x = node("_mul_scalar", "_mulscalar0", {"scalar": "0.0078125"}, [x])
    x = convolution("conv0", x, 3, WIDTHS[0], 3, 1, 1)
    x = batchnorm("bn0", x, WIDTHS[0], False)
    x = prelu("relu0", x, WIDTHS[0])
    plus = 0
    hw = SIZE
    for s in range(4):
        c = WIDTHS[s + 1]
        for u in range(UNITS[s]):
            p = "stage%d_unit%d" % (s + 1, u + 1)
            cin = WIDTHS[s] if u == 0 else c
            stride = 2 if u == 0 else 1
            y = batchnorm(p + "_bn1", x, cin, False)
            y = convolution(p + "_conv1", y, cin, c, 3, 1, 1)
            y = batchnorm(p + "_bn2", y, c, False)
            y = prelu(p + "_relu1", y, c)
            y = convolution(p + "_conv2", y, c, c, 3, stride, 1)
            y = batchnorm(p + "_bn3", y, c, False)
            if u == 0:
                sc = convolution(p + "_conv1sc", x, cin, c, 1, stride, 0)
                sc = batchnorm(p + "_sc", sc, c, False)
            else:
                sc = x
            x = node("elemwise_add", "_plus%d" % plus, None, [y, sc])
            plus += 1
        hw = (hw + 1) // 2
    x = batchnorm("bn1", x, WIDTHS[4], False)
    x = node("Dropout", "dropout0", {"p": "0.4"}, [x])
    w = variable("pre_fc1_weight", (EMB, WIDTHS[4] * hw * hw))
    b = variable("pre_fc1_bias", (EMB,))
    x = node("FullyConnected", "pre_fc1", {"num_hidden": str(EMB)}, [x, w, b])
    x = batchnorm("fc1", x, EMB, True)
    # the saved training graph carries a classification head after fc1; face_model.py cuts at fc1_output
    w7 = variable("fc7_weight", (5, EMB))
    x = node("FullyConnected", "fc7", {"no_bias": "True", "num_hidden": "5"}, [x, w7])
    lab = variable("softmax_label")
    x = node("SoftmaxOutput", "softmax", None, [x, lab])
    return nodes, arg_nodes, [[x, 0, 0]], tensors


def main():
    nodes, arg_nodes, heads, tensors = build_graph()
    sym = {"nodes": nodes, "arg_nodes": arg_nodes, "node_row_ptr": list(range(len(nodes) + 1)), "heads": heads,
           "attrs": {"mxnet_version": ["int", 10300]}}
    with open(os.path.join(HERE, "mxnet_tiny-symbol.json"), "w") as f:
        json.dump(sym, f, indent=2)
        f.write("\n")

    rng = np.random.default_rng(20260104)
    values = {}
    for name, shape, aux in tensors:
        if name.endswith("_moving_var") or (name.endswith("_gamma") and "relu" not in name):
            v = rng.uniform(0.5, 1.5, shape)
        elif "relu" in name:
            v = rng.uniform(0.1, 0.3, shape)
        else:
            v = rng.standard_normal(shape) * (0.2 if len(shape) > 1 else 0.1)
        dt = np.float32
        if name == "stage3_unit1_conv1_weight":
            dt = np.float16                                  # a half-precision export
        if name == "fc1_beta":
            dt = np.float64
        values[name] = np.ascontiguousarray(v).astype(dt)

    # mx.model.save_checkpoint: {'arg:%s': v for arg_params} then {'aux:%s': v for aux_params}
    entries = [("arg:" + n, values[n]) for n, _, aux in tensors if not aux] + \
              [("aux:" + n, values[n]) for n, _, aux in tensors if aux]
    blob = [u64(0x112), u64(0), u64(len(entries))]
    for _, a in entries:
        blob.append(ndarray_record(a, dev_type=2, dev_id=3))
    blob.append(u64(len(entries)))
    for k, _ in entries:
        kb = k.encode("ascii")
        blob.append(u64(len(kb)) + kb)
    with open(os.path.join(HERE, "mxnet_tiny-0000.params"), "wb") as f:
        f.write(b"".join(blob))
    np.savez(os.path.join(HERE, "mxnet_tiny_expected.npz"), **values)
    print("wrote %d tensors, %d bytes" % (len(entries), sum(len(b) for b in blob)))


if __name__ == "__main__":
    main()
