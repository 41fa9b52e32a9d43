"""mxnet_format — read (and write) the two files of an MXNet checkpoint without MXNet.

The reference loads its backbone with `mx.model.load_checkpoint(prefix, epoch)` (code/face_model.py:34)
from `model-r100-ii/model-symbol.json` + `model-0000.params` (code/arcface_prepreq.sh:13-20), takes
the internal output `fc1_output` (code/face_model.py:35-36) and binds it.  MXNet is not installable
here and the checkpoint cannot be downloaded, so this module restates the two published formats:

  * `.params` — `mx.nd.save` of a dict: uint64 0x112, uint64 0, uint64 count, `count` NDArray records,
    uint64 count, `count` names (uint64 length + bytes, prefixed "arg:" / "aux:").  An NDArray
    record is: uint32 magic (0xF993FAC9 "V2", 0xF993FAC8 "V1", 0xF993FACA "V3"; anything else is the
    pre-1.0 layout where that word is already ndim), [V2/V3: int32 storage type, 0 = dense],
    shape (V1/V2: uint32 ndim + int64 dims; V3: int32 ndim + int64 dims; legacy: uint32 ndim +
    uint32 dims), int32 device type, int32 device id, int32 type flag (0 f32, 1 f64, 2 f16, 3 u8,
    4 i32, 5 i8, 6 i64), raw little-endian data.
  * `-symbol.json` — the NNVM graph: "nodes" [{op, name, attrs|attr|param, inputs [[id, out, ver]]}],
    "arg_nodes", "heads".

PARITY UNPINNED: no file written by MXNet exists in this container to read back; tests round-trip
this module's own writer and check the parsed architecture against the tensor name table
(weights.tensor_shapes).  The reader is driven by what the files say (names, shapes, attrs), not by
hard-coded offsets, and refuses what it does not understand.
"""
import json
import struct

import numpy as np

LIST_MAGIC = 0x112
V1_MAGIC, V2_MAGIC, V3_MAGIC = 0xF993FAC8, 0xF993FAC9, 0xF993FACA
_DTYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
_FLAGS = {np.dtype(v): k for k, v in _DTYPES.items()}


class _Reader(object):
    def __init__(self, buf):
        self.buf, self.pos = buf, 0

    def take(self, fmt):
        size = struct.calcsize(fmt)
        if self.pos + size > len(self.buf):
            raise ValueError("truncated .params file")
        v = struct.unpack_from(fmt, self.buf, self.pos)
        self.pos += size
        return v if len(v) > 1 else v[0]

    def raw(self, n):
        if self.pos + n > len(self.buf):
            raise ValueError("truncated .params file")
        b = self.buf[self.pos:self.pos + n]
        self.pos += n
        return b


def _read_ndarray(r):
    magic = r.take("<I")
    if magic in (V2_MAGIC, V3_MAGIC):
        stype = r.take("<i")
        if stype != 0:
            raise NotImplementedError("sparse NDArray (storage type %d) in checkpoint" % stype)
        ndim = r.take("<i") if magic == V3_MAGIC else r.take("<I")
        if ndim < 0:
            raise ValueError("NDArray of unknown shape")
        shape = tuple(r.take("<%dq" % ndim)) if ndim > 1 else ((r.take("<q"),) if ndim == 1 else ())
    elif magic == V1_MAGIC:
        ndim = r.take("<I")
        shape = tuple(r.take("<%dq" % ndim)) if ndim > 1 else ((r.take("<q"),) if ndim == 1 else ())
    else:                                        # pre-1.0: the word just read is ndim, dims are uint32
        ndim = magic
        if ndim > 32:
            raise ValueError("not an NDArray record (leading word 0x%08X)" % magic)
        shape = tuple(r.take("<%dI" % ndim)) if ndim > 1 else ((r.take("<I"),) if ndim == 1 else ())
    if ndim == 0:
        return np.zeros((), np.float32)          # MXNet writes nothing more for a none array
    r.take("<ii")                                # context the array was saved from
    flag = r.take("<i")
    if flag not in _DTYPES:
        raise NotImplementedError("NDArray type flag %d" % flag)
    dt = np.dtype(_DTYPES[flag]).newbyteorder("<")
    count = int(np.prod(shape, dtype=np.int64))
    return np.frombuffer(r.raw(count * dt.itemsize), dtype=dt).reshape(shape).copy()


def load_ndarray_file(path):
    """-> dict name -> ndarray (names as stored, e.g. "arg:conv0_weight"), or a list if unnamed."""
    with open(path, "rb") as f:
        r = _Reader(f.read())
    header, _reserved = r.take("<QQ")
    if header != LIST_MAGIC:
        raise ValueError("%s is not an MXNet NDArray file (header 0x%X)" % (path, header))
    n = r.take("<Q")
    arrays = [_read_ndarray(r) for _ in range(n)]
    nn = r.take("<Q")
    names = [r.raw(r.take("<Q")).decode("utf8") for _ in range(nn)]
    if nn == 0:
        return arrays
    if nn != n:
        raise ValueError("%d arrays but %d names" % (n, nn))
    return dict(zip(names, arrays))


def save_ndarray_file(path, named):
    """mx.nd.save(path, dict) in the V2 layout (what MXNet 1.x writes)."""
    out = [struct.pack("<QQQ", LIST_MAGIC, 0, len(named))]
    for a in named.values():
        a = np.ascontiguousarray(a)
        if a.dtype not in _FLAGS:
            raise TypeError("dtype %s not storable" % a.dtype)
        out.append(struct.pack("<IiI", V2_MAGIC, 0, a.ndim))
        out.append(struct.pack("<%dq" % a.ndim, *a.shape))
        out.append(struct.pack("<iii", 1, 0, _FLAGS[a.dtype]))          # cpu(0)
        out.append(a.astype(a.dtype.newbyteorder("<")).tobytes())
    out.append(struct.pack("<Q", len(named)))
    for k in named:
        b = k.encode("utf8")
        out.append(struct.pack("<Q", len(b)) + b)
    with open(path, "wb") as f:
        f.write(b"".join(out))


def split_arg_aux(named):
    """save_dict -> (arg_params, aux_params) as mx.model.load_checkpoint does."""
    arg, aux = {}, {}
    for k, v in named.items():
        tp, _, name = k.partition(":")
        if tp == "arg":
            arg[name] = v
        elif tp == "aux":
            aux[name] = v
        else:
            raise ValueError("checkpoint entry %r is neither arg: nor aux:" % k)
    return arg, aux


# ---------------------------------------------------------------------------------------------------
# symbol
# ---------------------------------------------------------------------------------------------------
def _attrs(node):
    for key in ("attrs", "attr", "param"):                      # the key changed across MXNet versions
        if key in node and isinstance(node[key], dict):
            return node[key]
    return {}


def _tuple_attr(s):
    return tuple(int(v) for v in str(s).strip("()[] ").replace(" ", "").split(",") if v != "")


def _bool_attr(s):
    return str(s).strip().lower() in ("true", "1")


def load_symbol(path):
    with open(path) as f:
        return json.load(f)


def ir_config_from_symbol(sym, output="fc1_output"):
    """Architecture the backbone needs, read from the graph of an insightface LResNet-E-IR symbol:
    dict(units, widths, emb, bn_eps, fix_gamma (names of BatchNorms with fix_gamma=True)).
    Checks every layer the C library will run has the operator and hyper-parameters it assumes and
    that `output` (code/face_model.py:35-36) exists; raises ValueError otherwise."""
    nodes = sym["nodes"]
    by_name = {n["name"]: n for n in nodes}
    layer = output[:-len("_output")] if output.endswith("_output") else output
    if layer not in by_name or by_name[layer]["op"] == "null":
        raise ValueError("symbol has no internal output %r" % output)

    def need(name, op):
        if name not in by_name:
            raise ValueError("symbol has no node %r" % name)
        if by_name[name]["op"] != op:
            raise ValueError("node %r is %s, expected %s" % (name, by_name[name]["op"], op))
        return _attrs(by_name[name])

    def conv(name, kernel, stride, pad):
        a = need(name, "Convolution")
        got = (_tuple_attr(a.get("kernel", "")), _tuple_attr(a.get("stride", "(1,1)")) or (1, 1),
               _tuple_attr(a.get("pad", "(0,0)")) or (0, 0))
        if got != (kernel, stride, pad) or not _bool_attr(a.get("no_bias", "False")):
            raise ValueError("convolution %s has kernel/stride/pad %s no_bias=%s; expected %s without bias"
                             % (name, got, a.get("no_bias"), (kernel, stride, pad)))
        return int(a["num_filter"])

    fix_gamma, eps = set(), set()

    def bn(name):
        a = need(name, "BatchNorm")
        # MXNet's BatchNorm defaults: fix_gamma=True, eps=1e-3
        if _bool_attr(a.get("fix_gamma", "True")):
            fix_gamma.add(name)
        eps.add(float(a.get("eps", 1e-3)))

    def prelu(name):
        a = need(name, "LeakyReLU")
        if a.get("act_type", "leaky") != "prelu":
            raise ValueError("activation %s is %s, expected prelu" % (name, a.get("act_type")))

    # input normalisation (x - 127.5) * 0.0078125 baked into the graph
    scal = [(n["op"], float(_attrs(n).get("scalar", "nan"))) for n in nodes if n["op"] in ("_minus_scalar", "_mul_scalar")]
    if scal[:2] != [("_minus_scalar", 127.5), ("_mul_scalar", 0.0078125)]:
        raise ValueError("symbol does not start with (data - 127.5) * 0.0078125: %s" % (scal[:2],))
    widths = [conv("conv0", (3, 3), (1, 1), (1, 1))]
    bn("bn0")
    prelu("relu0")
    units = []
    for s in range(1, 5):
        u = 0
        while "stage%d_unit%d_conv1" % (s, u + 1) in by_name:
            u += 1
            p = "stage%d_unit%d" % (s, u)
            bn(p + "_bn1")
            c1 = conv(p + "_conv1", (3, 3), (1, 1), (1, 1))
            bn(p + "_bn2")
            prelu(p + "_relu1")
            stride = (2, 2) if u == 1 else (1, 1)
            c2 = conv(p + "_conv2", (3, 3), stride, (1, 1))
            bn(p + "_bn3")
            if u == 1:
                sc = conv(p + "_conv1sc", (1, 1), stride, (0, 0))
                bn(p + "_sc")
                if sc != c2:
                    raise ValueError("%s: shortcut width %d != %d" % (p, sc, c2))
            if c1 != c2 or (u > 1 and c2 != widths[-1]):
                raise ValueError("%s: widths %d/%d do not form an IR unit" % (p, c1, c2))
            if u == 1:
                widths.append(c2)
        if u == 0:
            raise ValueError("symbol has no stage %d" % s)
        units.append(u)
    bn("bn1")
    emb = int(need("pre_fc1", "FullyConnected")["num_hidden"])
    bn("fc1")
    if len(eps) != 1:
        raise ValueError("BatchNorm layers disagree on eps: %s" % sorted(eps))
    extra = fix_gamma - {"fc1"}
    if extra or "fc1" not in fix_gamma:
        raise ValueError("fix_gamma set on %s; the backbone folds gamma for every BatchNorm except fc1"
                         % sorted(fix_gamma))
    return {"units": tuple(units), "widths": tuple(widths), "emb": emb, "bn_eps": eps.pop(),
            "fix_gamma": sorted(fix_gamma)}


def write_ir_symbol(path, units, widths=(64, 64, 128, 256, 512), emb=512, bn_eps=2e-5):
    """An LResNet-E-IR graph in MXNet's JSON layout (insightface fresnet.py naming), so that synthetic
    or re-exported checkpoints carry the same two files the reference expects."""
    nodes, arg_nodes = [], []

    def var(name):
        nodes.append({"op": "null", "name": name, "inputs": []})
        arg_nodes.append(len(nodes) - 1)
        return len(nodes) - 1

    def op(kind, name, inputs, **attrs):
        nodes.append({"op": kind, "name": name, "attrs": {k: str(v) for k, v in attrs.items()},
                      "inputs": [[i, 0, 0] for i in inputs]})
        return len(nodes) - 1

    def bn(name, x, fix_gamma=False):
        ins = [x] + [var(name + s) for s in ("_gamma", "_beta", "_moving_mean", "_moving_var")]
        return op("BatchNorm", name, ins, eps=bn_eps, fix_gamma=fix_gamma, momentum=0.9)

    def conv(name, x, nf, k, stride, pad):
        return op("Convolution", name, [x, var(name + "_weight")], kernel="(%d, %d)" % (k, k), no_bias="True",
                  num_filter=nf, pad="(%d, %d)" % (pad, pad), stride="(%d, %d)" % (stride, stride))

    def prelu(name, x):
        return op("LeakyReLU", name, [x, var(name + "_gamma")], act_type="prelu")

    x = var("data")
    x = op("_copy", "id", [x])
    x = op("_minus_scalar", "_minusscalar0", [x], scalar=127.5)
    x = op("_mul_scalar", "_mulscalar0", [x], scalar=0.0078125)
    x = prelu("relu0", bn("bn0", conv("conv0", x, widths[0], 3, 1, 1)))
    for s in range(4):
        for u in range(units[s]):
            p = "stage%d_unit%d" % (s + 1, u + 1)
            stride = 2 if u == 0 else 1
            y = bn(p + "_bn1", x)
            y = conv(p + "_conv1", y, widths[s + 1], 3, 1, 1)
            y = prelu(p + "_relu1", bn(p + "_bn2", y))
            y = bn(p + "_bn3", conv(p + "_conv2", y, widths[s + 1], 3, stride, 1))
            sc = bn(p + "_sc", conv(p + "_conv1sc", x, widths[s + 1], 1, stride, 0)) if u == 0 else x
            x = op("elemwise_add", "_plus%d" % len(nodes), [y, sc])
    x = bn("bn1", x)
    x = op("Dropout", "dropout0", [x], p=0.4)
    x = op("FullyConnected", "pre_fc1", [x, var("pre_fc1_weight"), var("pre_fc1_bias")], num_hidden=emb)
    x = bn("fc1", x, fix_gamma=True)
    sym = {"nodes": nodes, "arg_nodes": arg_nodes, "node_row_ptr": list(range(len(nodes) + 1)), "heads": [[x, 0, 0]],
           "attrs": {"mxnet_version": ["int", 10200]}}
    with open(path, "w") as f:
        json.dump(sym, f)
    return sym


def load_checkpoint(prefix, epoch):
    """mx.model.load_checkpoint(prefix, epoch) -> (symbol dict, arg_params, aux_params)."""
    sym = load_symbol("%s-symbol.json" % prefix)
    arg, aux = split_arg_aux(load_ndarray_file("%s-%04d.params" % (prefix, epoch)))
    return sym, arg, aux


def save_checkpoint(prefix, epoch, params, units=None, widths=(64, 64, 128, 256, 512), emb=512, bn_eps=2e-5):
    """Write `params` (flat dict with MXNet tensor names) as prefix-symbol.json + prefix-%04d.params."""
    from . import weights as W
    units = units or W.infer_units(params)
    write_ir_symbol("%s-symbol.json" % prefix, units, widths, emb, bn_eps)
    named = {}
    for k, v in params.items():
        aux = k.endswith("_moving_mean") or k.endswith("_moving_var")
        named[("aux:" if aux else "arg:") + k] = np.asarray(v, dtype=np.float32)
    save_ndarray_file("%s-%04d.params" % (prefix, epoch), named)
