"""hdf5_lite — the subset of HDF5 that Keras weight files use, in pure Python (no h5py / libhdf5).

The reference persists its pair scorers with Keras `save_weights` / `load_weights` on `<name>.h5`
(code/siamese.py:114-125) and ships pretrained `disguisedModel.h5`, `ensemble1.h5`
(code/arcface_prepreq.sh:7-12).  Keras 2.1.2 writes through h5py with libhdf5 defaults:
superblock version 0, "old style" groups (symbol-table message -> v1 B-tree -> symbol-table nodes,
names in a local heap), version-1 object headers, contiguous little-endian float32 datasets, and
attributes holding fixed-length byte strings (`layer_names`, `weight_names`, `backend`,
`keras_version`; keras/engine/topology.py save_weights_to_hdf5_group).

Reader: superblock 0/1 (and 2/3 with compact link-message groups), object headers v1/v2 with
continuation blocks, dataspace v1/v2, datatypes fixed-point / IEEE float / fixed string /
variable-length string (global heap), data layout v1-v4 contiguous + compact (chunked raises),
attribute messages v1-v3.  Writer: exactly the h5py-default structures listed above.
Cross-checked in tests against the real libhdf5 (ctypes on /opt/conda/lib/libhdf5.so, when present):
files written here are read back by libhdf5, files created by libhdf5 are read here.

Format source: "HDF5 File Format Specification Version 2.0" (The HDF Group), restated from memory.
"""
import struct

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(IOError):
    pass


# =====================================================================================================
# reader
# =====================================================================================================
class _Type(object):
    """Decoded datatype message: numpy dtype for numeric / fixed strings; vlen strings flagged."""

    def __init__(self, dtype=None, size=0, vlen_str=False):
        self.dtype, self.size, self.vlen_str = dtype, size, vlen_str


class Dataset(object):
    def __init__(self, f, name, shape, typ, layout, attrs):
        self._f, self.name, self.shape, self._typ, self._layout, self.attrs = f, name, shape, typ, layout, attrs
        self.dtype = typ.dtype

    def read(self):
        kind, a, b = self._layout
        count = int(np.prod(self.shape, dtype=np.int64)) if len(self.shape) else 1
        nbytes = count * self._typ.size
        if kind == "contiguous":
            if a == UNDEF:                            # never written: HDF5 returns the fill value (zeros)
                raw = b"\0" * nbytes
            else:
                raw = self._f._bytes(a, nbytes)
        elif kind == "compact":
            raw = a[:nbytes]
        else:
            raise H5Error("dataset %s uses chunked storage, which Keras weight files do not" % self.name)
        return self._f._decode(raw, self._typ, self.shape)

    def __getitem__(self, key):
        return self.read()[key]


class Group(object):
    def __init__(self, f, name, links, attrs):
        self._f, self.name, self._links, self.attrs = f, name, links, attrs

    def keys(self):
        return list(self._links)

    def __contains__(self, key):
        try:
            self[key]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group) or part not in node._links:
                raise KeyError(path)
            base = node.name.rstrip("/") + "/" + part
            node = node._f._object(node._links[part], base)
        return node

    def visit_datasets(self, prefix=""):
        out = []
        for k in self.keys():
            o = self[k]
            if isinstance(o, Group):
                out += o.visit_datasets(prefix + k + "/")
            else:
                out.append((prefix + k, o))
        return out


class File(Group):
    def __init__(self, path):
        with open(path, "rb") as fh:
            self._buf = fh.read()
        self._f = self
        self._cache = {}
        root = self._superblock()
        g = self._object(root, "/")
        if not isinstance(g, Group):
            raise H5Error("root object is not a group")
        Group.__init__(self, self, "/", g._links, g.attrs)

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    # ---- primitives ---------------------------------------------------------------------------------
    def _bytes(self, off, n):
        off += self._base
        if off < 0 or off + n > len(self._buf):
            raise H5Error("read of %d bytes at %d beyond the end of the file" % (n, off))
        return self._buf[off:off + n]

    def _u(self, off, n):
        return int.from_bytes(self._bytes(off, n), "little")

    def _superblock(self):
        pos = -1
        for cand in [0] + [512 << i for i in range(12)]:
            if self._buf[cand:cand + 8] == SIGNATURE:
                pos = cand
                break
        if pos < 0:
            raise H5Error("not an HDF5 file (signature not found)")
        self._base = 0
        b = self._buf
        ver = b[pos + 8]
        if ver in (0, 1):
            self.O, self.L = b[pos + 13], b[pos + 14]
            p = pos + 24 + (4 if ver == 1 else 0)
            base = int.from_bytes(b[p:p + self.O], "little")
            p += 4 * self.O                           # base, free-space info, end of file, driver info
            root_hdr = int.from_bytes(b[p + self.O:p + 2 * self.O], "little")     # symbol table entry: name off, header
        elif ver in (2, 3):
            self.O, self.L = b[pos + 9], b[pos + 10]
            p = pos + 12
            base = int.from_bytes(b[p:p + self.O], "little")
            root_hdr = int.from_bytes(b[p + 3 * self.O:p + 4 * self.O], "little")
        else:
            raise H5Error("superblock version %d not supported" % ver)
        if self.O != 8 or self.L != 8:
            raise H5Error("only 8-byte offsets/lengths are supported (file has %d/%d)" % (self.O, self.L))
        self._base = base if base != UNDEF else 0
        return root_hdr

    # ---- object headers -------------------------------------------------------------------------------
    def _messages(self, addr):
        """-> list of (type, flags, data bytes) of the object header at addr (v1 or v2)."""
        msgs = []
        if self._bytes(addr, 4) == b"OHDR":
            ver, flags = self._u(addr + 4, 1), self._u(addr + 5, 1)
            if ver != 2:
                raise H5Error("object header version %d" % ver)
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            szlen = 1 << (flags & 3)
            chunk0 = self._u(p, szlen)
            p += szlen
            blocks = [(p, chunk0)]
            track = bool(flags & 0x04)
            while blocks:
                start, size = blocks.pop(0)
                q, end = start, start + size
                while q + 4 <= end:
                    mtype, msize, mflags = self._u(q, 1), self._u(q + 1, 2), self._u(q + 3, 1)
                    q += 4 + (2 if track else 0)
                    data = self._bytes(q, msize)
                    q += msize
                    if mtype == 0x10:
                        o, ln = struct.unpack_from("<QQ", data)
                        blocks.append((o + 4, ln - 8))          # skip 'OCHK', drop the checksum
                    elif mtype != 0:
                        msgs.append((mtype, mflags, data))
            return msgs
        ver = self._u(addr, 1)
        if ver != 1:
            raise H5Error("object header version %d at %d" % (ver, addr))
        nmsg, size = self._u(addr + 2, 2), self._u(addr + 8, 4)
        blocks = [(addr + 16, size)]
        while blocks and len(msgs) < nmsg + 64:
            start, size = blocks.pop(0)
            q, end = start, start + size
            while q + 8 <= end:
                mtype, msize, mflags = self._u(q, 2), self._u(q + 2, 2), self._u(q + 4, 1)
                data = self._bytes(q + 8, msize)
                q += 8 + msize
                if mtype == 0x10:
                    o, ln = struct.unpack_from("<QQ", data)
                    blocks.append((o, ln))
                elif mtype != 0:
                    msgs.append((mtype, mflags, data))
        return msgs

    def _object(self, addr, name):
        if addr in self._cache:
            return self._cache[addr]
        msgs = self._messages(addr)
        attrs, links = {}, None
        shape = typ = layout = None
        for mtype, mflags, d in msgs:
            if mtype == 0x0C:
                k, v = self._attribute(d)
                attrs[k] = v
            elif mtype == 0x11:
                btree, heap = struct.unpack_from("<QQ", d)
                links = self._symbol_table(btree, heap)
            elif mtype == 0x06:
                links = {} if links is None else links
                k, a = self._link(d)
                links[k] = a
            elif mtype == 0x02:
                links = {} if links is None else links
                ver, fl = d[0], d[1]
                p = 2 + (8 if fl & 1 else 0)
                fheap = struct.unpack_from("<Q", d, p)[0]
                if fheap != UNDEF:
                    raise H5Error("group %s stores its links densely (fractal heap): not supported" % name)
            elif mtype == 0x01:
                shape = self._dataspace(d)
            elif mtype == 0x03:
                typ = self._datatype(d)
            elif mtype == 0x08:
                layout = self._layout(d)
        if typ is not None and shape is not None and layout is not None:
            obj = Dataset(self, name, shape, typ, layout, attrs)
        else:
            obj = Group(self, name, links or {}, attrs)
        self._cache[addr] = obj
        return obj

    # ---- groups -----------------------------------------------------------------------------------------
    def _heap_name(self, heap_data, off):
        raw = self._bytes(heap_data + off, min(1024, len(self._buf) - self._base - heap_data - off))
        return raw.split(b"\0", 1)[0].decode("utf8")

    def _symbol_table(self, btree, heap):
        if self._bytes(heap, 4) != b"HEAP":
            raise H5Error("bad local heap signature")
        heap_data = self._u(heap + 24, 8)
        links = {}

        def walk(node):
            if self._bytes(node, 4) != b"TREE":
                raise H5Error("bad B-tree signature")
            ntype, level, used = self._u(node + 4, 1), self._u(node + 5, 1), self._u(node + 6, 2)
            if ntype != 0:
                raise H5Error("B-tree node type %d in a group" % ntype)
            p = node + 24
            for i in range(used):
                child = self._u(p + 8 + 16 * i, 8)
                if level > 0:
                    walk(child)
                else:
                    if self._bytes(child, 4) != b"SNOD":
                        raise H5Error("bad symbol table node signature")
                    n = self._u(child + 6, 2)
                    for e in range(n):
                        q = child + 8 + 40 * e
                        links[self._heap_name(heap_data, self._u(q, 8))] = self._u(q + 8, 8)
        walk(btree)
        return links

    def _link(self, d):
        ver, fl = d[0], d[1]
        p = 2
        ltype = 0
        if fl & 0x08:
            ltype = d[p]
            p += 1
        if fl & 0x04:
            p += 8
        if fl & 0x10:
            p += 1
        ln = 1 << (fl & 3)
        n = int.from_bytes(d[p:p + ln], "little")
        p += ln
        name = d[p:p + n].decode("utf8")
        p += n
        if ltype != 0:
            raise H5Error("soft/external link %r not supported" % name)
        return name, struct.unpack_from("<Q", d, p)[0]

    # ---- messages ---------------------------------------------------------------------------------------
    @staticmethod
    def _dataspace(d):
        ver, rank, fl = d[0], d[1], d[2]
        if ver == 1:
            p = 8
        elif ver == 2:
            if d[3] == 2:
                return None
            p = 4
        else:
            raise H5Error("dataspace message version %d" % ver)
        return tuple(struct.unpack_from("<%dQ" % rank, d, p)) if rank else ()

    def _datatype(self, d):
        cls, ver = d[0] & 0x0F, d[0] >> 4
        bits = d[1] | (d[2] << 8) | (d[3] << 16)
        size = struct.unpack_from("<I", d, 4)[0]
        order = ">" if bits & 1 else "<"
        if cls == 0:
            return _Type(np.dtype("%s%s%d" % (order, "i" if bits & 8 else "u", size)), size)
        if cls == 1:
            if size not in (2, 4, 8):
                raise H5Error("float of %d bytes" % size)
            return _Type(np.dtype("%sf%d" % (order, size)), size)
        if cls == 3:
            return _Type(np.dtype("S%d" % size), size)
        if cls == 9:
            if (bits & 0x0F) != 1:
                raise H5Error("variable-length sequences are not supported")
            return _Type(np.dtype(object), 16, vlen_str=True)
        raise H5Error("datatype class %d not supported" % cls)

    def _layout(self, d):
        ver = d[0]
        if ver in (1, 2):
            rank, cls = d[1], d[2]
            p = 8
            addr = None
            if cls != 0:
                addr = struct.unpack_from("<Q", d, p)[0]
                p += 8
            p += 4 * rank
            if cls == 1:
                return ("contiguous", addr, None)
            if cls == 0:
                n = struct.unpack_from("<I", d, p)[0]
                return ("compact", bytes(d[p + 4:p + 4 + n]), None)
            return ("chunked", addr, None)
        if ver in (3, 4):
            cls = d[1]
            if cls == 0:
                n = struct.unpack_from("<H", d, 2)[0]
                return ("compact", bytes(d[4:4 + n]), None)
            if cls == 1:
                addr, size = struct.unpack_from("<QQ", d, 2)
                return ("contiguous", addr, size)
            return ("chunked", None, None)
        raise H5Error("data layout message version %d" % ver)

    def _decode(self, raw, typ, shape):
        if typ.vlen_str:
            count = int(np.prod(shape, dtype=np.int64)) if shape else 1
            vals = []
            for i in range(count):
                ln, addr, idx = struct.unpack_from("<IQI", raw, 16 * i)
                vals.append(self._global_heap_object(addr, idx)[:ln] if ln else b"")
            arr = np.empty(count, dtype=object)
            arr[:] = vals
            return arr.reshape(shape) if shape else arr[0]
        arr = np.frombuffer(raw, dtype=typ.dtype, count=int(np.prod(shape, dtype=np.int64)) if shape else 1)
        arr = arr.reshape(shape) if shape else arr[0]
        if typ.dtype.kind in "fiu" and isinstance(arr, np.ndarray):
            arr = arr.astype(typ.dtype.newbyteorder("="))
        return arr

    def _global_heap_object(self, addr, idx):
        if self._bytes(addr, 4) != b"GCOL":
            raise H5Error("bad global heap signature")
        size = self._u(addr + 8, 8)
        p, end = addr + 16, addr + size
        while p + 16 <= end:
            oid, osz = self._u(p, 2), self._u(p + 8, 8)
            if oid == idx:
                return self._bytes(p + 16, osz)
            if oid == 0:
                break
            p += 16 + ((osz + 7) & ~7)
        raise H5Error("global heap object %d not found" % idx)

    def _attribute(self, d):
        ver = d[0]
        nsz, tsz, ssz = struct.unpack_from("<HHH", d, 2)
        if ver == 1:
            p = 8
            pad = lambda n: (n + 7) & ~7
        elif ver in (2, 3):
            p = 8 + (1 if ver == 3 else 0)
            pad = lambda n: n
        else:
            raise H5Error("attribute message version %d" % ver)
        name = bytes(d[p:p + nsz]).split(b"\0", 1)[0].decode("utf8")
        p += pad(nsz)
        typ = self._datatype(d[p:p + tsz])
        p += pad(tsz)
        shape = self._dataspace(d[p:p + ssz])
        p += pad(ssz)
        if shape is None:
            return name, None
        return name, self._decode(bytes(d[p:]), typ, shape)


# =====================================================================================================
# writer
# =====================================================================================================
def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _dtype_message(dt):
    dt = np.dtype(dt)
    if dt.kind == "f" and dt.itemsize in (4, 8):
        if dt.itemsize == 4:
            bits, props = (0x20, 31, 0), struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        else:
            bits, props = (0x20, 63, 0), struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        return bytes([0x11, bits[0], bits[1], bits[2]]) + struct.pack("<I", dt.itemsize) + props
    if dt.kind in "iu":
        return bytes([0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0]) + struct.pack("<IHH", dt.itemsize, 0, 8 * dt.itemsize)
    if dt.kind == "S":
        return bytes([0x13, 0x01, 0, 0]) + struct.pack("<I", max(1, dt.itemsize))        # null-padded ASCII
    raise TypeError("cannot store dtype %s" % dt)


def _dataspace_message(shape):
    return struct.pack("<BBBBI", 1, len(shape), 0, 0, 0) + b"".join(struct.pack("<Q", s) for s in shape)


def _message(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHBBBB", mtype, len(data), flags, 0, 0, 0) + data


def _attr_message(name, value):
    a = np.asarray(value)
    if a.dtype.kind == "U":
        a = np.char.encode(a, "utf8")
    if a.dtype.kind == "O":
        raise TypeError("attribute %s: object arrays are not storable" % name)
    if a.dtype.kind == "S" and a.dtype.itemsize == 0:
        a = a.astype("S1")
    if a.dtype.kind in "fiu":
        a = a.astype(a.dtype.newbyteorder("<"))
    nm = name.encode("utf8") + b"\0"
    dt, ds = _dtype_message(a.dtype), _dataspace_message(a.shape)
    body = struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + a.tobytes()
    if len(body) > 0xFFF8:
        raise ValueError("attribute %s too large for an object-header message" % name)
    return _message(0x0C, body)


def _object_header(messages):
    body = b"".join(messages)
    return struct.pack("<BBHII", 1, 0, len(messages), 1, len(body)) + b"\0" * 4 + body


class _WGroup(object):
    def __init__(self):
        self.children, self.attrs = {}, {}


class _WDataset(object):
    def __init__(self, array):
        self.array, self.attrs = array, {}


class Writer(object):
    """Build a tree with create_group / create_dataset / attrs, then save(path)."""

    def __init__(self):
        self.root = _WGroup()

    def _walk(self, path, create=True):
        node = self.root
        for part in [p for p in path.split("/") if p]:
            if part not in node.children:
                if not create:
                    raise KeyError(path)
                node.children[part] = _WGroup()
            node = node.children[part]
            if not isinstance(node, _WGroup):
                raise ValueError("%s is a dataset" % part)
        return node

    def create_group(self, path):
        return self._walk(path)

    def create_dataset(self, path, array):
        parts = [p for p in path.split("/") if p]
        parent = self._walk("/".join(parts[:-1]))
        a = np.ascontiguousarray(array)
        if a.dtype.kind in "fiu":
            a = a.astype(a.dtype.newbyteorder("<"))
        parent.children[parts[-1]] = d = _WDataset(a)
        return d

    def set_attr(self, path, name, value):
        node = self.root
        for part in [p for p in path.split("/") if p]:
            node = node.children[part]
        node.attrs[name] = value

    def save(self, path):
        max_children = [1]

        def scan(g):
            max_children[0] = max(max_children[0], len(g.children))
            for c in g.children.values():
                if isinstance(c, _WGroup):
                    scan(c)
        scan(self.root)
        leaf_k = max(4, (max_children[0] + 1) // 2)          # one symbol-table node (2K entries) per group
        internal_k = 16
        if leaf_k > 0x7FFF:
            raise ValueError("too many objects in one group")
        chunks, pos = [], [96]                                # superblock v0 occupies [0, 96)

        def alloc(data):
            data = _pad8(data)
            addr = pos[0]
            chunks.append(data)
            pos[0] += len(data)
            return addr

        def reserve(n):
            addr = pos[0]
            chunks.append(None)
            pos[0] += (n + 7) & ~7
            return addr, len(chunks) - 1

        def emit_dataset(d):
            data_addr = alloc(d.array.tobytes()) if d.array.size else UNDEF
            msgs = [_message(0x01, _dataspace_message(d.array.shape)),
                    _message(0x03, _dtype_message(d.array.dtype), flags=1),
                    _message(0x05, struct.pack("<BBBB", 2, 2, 2, 0)),
                    _message(0x08, struct.pack("<BBQQ", 3, 1, data_addr, d.array.nbytes))]
            msgs += [_attr_message(k, v) for k, v in d.attrs.items()]
            return alloc(_object_header(msgs))

        def emit_group(g):
            names = sorted(g.children, key=lambda s: s.encode("utf8"))
            child_addr = {}
            for n in names:
                c = g.children[n]
                child_addr[n] = emit_group(c)[0] if isinstance(c, _WGroup) else emit_dataset(c)
            # local heap data segment: "" at offset 0, the names, one free block at the end
            seg, offs = bytearray(b"\0" * 8), {}
            for n in names:
                offs[n] = len(seg)
                seg += _pad8(n.encode("utf8") + b"\0")
            free_off = len(seg)
            seg += struct.pack("<QQ", 1, 32) + b"\0" * 16      # free block: next = 1 (none), size 32
            seg_addr = alloc(bytes(seg))
            heap_addr = alloc(b"HEAP" + struct.pack("<BBBBQQQ", 0, 0, 0, 0, len(seg), free_off, seg_addr))
            snod = bytearray(b"SNOD" + struct.pack("<BBH", 1, 0, len(names)))
            for n in names:
                snod += struct.pack("<QQII", offs[n], child_addr[n], 0, 0) + b"\0" * 16
            snod += b"\0" * (8 + 2 * leaf_k * 40 - len(snod))
            snod_addr = alloc(bytes(snod)) if names else None
            tree = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, UNDEF, UNDEF))
            tree += struct.pack("<Q", 0)
            if names:
                tree += struct.pack("<QQ", snod_addr, offs[names[-1]])
            tree += b"\0" * (24 + (2 * internal_k + 1) * 8 + 2 * internal_k * 8 - len(tree))
            tree_addr = alloc(bytes(tree))
            msgs = [_message(0x11, struct.pack("<QQ", tree_addr, heap_addr))]
            msgs += [_attr_message(k, v) for k, v in g.attrs.items()]
            return alloc(_object_header(msgs)), tree_addr, heap_addr

        root_hdr, root_tree, root_heap = emit_group(self.root)
        eof = pos[0]
        sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, leaf_k, internal_k, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", root_tree, root_heap)
        assert len(sb) == 96
        with open(path, "wb") as f:
            f.write(sb)
            for c in chunks:
                f.write(c)


# =====================================================================================================
# Keras weight files
# =====================================================================================================
def save_keras_weights(path, layers, backend="tensorflow", keras_version="2.1.2"):
    """keras.engine.topology.save_weights_to_hdf5_group: `layers` = [(layer_name, [(weight_name, array)])]
    in model.layers order (layers without weights included, with an empty list)."""
    w = Writer()
    w.root.attrs["layer_names"] = np.array([n.encode("utf8") for n, _ in layers])
    w.root.attrs["backend"] = np.bytes_(backend.encode("utf8"))
    w.root.attrs["keras_version"] = np.bytes_(keras_version.encode("utf8"))
    for lname, weights in layers:
        g = w.create_group(lname)
        if weights:
            g.attrs["weight_names"] = np.array([wn.encode("utf8") for wn, _ in weights])
        else:
            g.attrs["weight_names"] = np.zeros((0,), np.float64)     # h5py stores [] as an empty float64 array
        for wn, arr in weights:
            w.create_dataset(lname + "/" + wn, np.asarray(arr))
    w.save(path)


def load_keras_weights(path):
    """-> [(layer_name, [(weight_name, array)])] in file order; accepts weight files and full-model
    files (weights under /model_weights, keras/models.py save_model)."""
    f = File(path)
    g = f["model_weights"] if ("layer_names" not in f.attrs and "model_weights" in f) else f
    if "layer_names" not in g.attrs:
        raise H5Error("%s has no layer_names attribute: not a Keras weight file" % path)

    def text(v):
        return v.decode("utf8") if isinstance(v, (bytes, np.bytes_)) else str(v)
    out = []
    for lname in np.atleast_1d(g.attrs["layer_names"]):
        lname = text(lname)
        lg = g[lname]
        wn = lg.attrs.get("weight_names")
        names = [] if wn is None or np.asarray(wn).dtype.kind == "f" else [text(n) for n in np.atleast_1d(wn)]
        out.append((lname, [(n, np.asarray(lg[n].read())) for n in names]))
    return out
