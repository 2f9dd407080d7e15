"""A small HDF5 reader / writer in NumPy for the one shape of file this path meets: Keras `.h5` checkpoints
(`model.save(path)` at SRCNN_model.py:249-260, ESRGAN_model.py:981-995; `load_model` at ESRGAN_model.py:143-149,
SRCNN_model.py:35).  h5py is not installed in this image, so `sr355.wrappers.load_pretrained` reads `.h5` through this
module when h5py is missing (and through h5py when it is there).

What such a file contains (HDF5 File Format Specification 3.0; h5py's default `libver="earliest"`):
  * superblock version 0 (or 1), 8-byte offsets and lengths;
  * version-1 object headers (message continuation blocks included);
  * "old-style" groups: a symbol-table message -> a version-1 B-tree of symbol-table nodes (SNOD) + a local heap of names;
  * datasets with a simple dataspace, an IEEE little-endian float (or fixed-point / fixed-length string) datatype and a
    CONTIGUOUS (or compact) layout, no filters -- Keras never chunks or compresses its weights;
  * attributes (`layer_names`, `weight_names`, `keras_version`, `backend`) as version-1 attribute messages.
The reader handles exactly that and says what it met when a file goes beyond it (chunked / filtered datasets, version-2 object
headers of `libver="latest"`, variable-length strings: NotImplementedError with the feature's name).  The writer emits the
same structures, so that a fixture written here exercises the same parser paths as a file from h5py."""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIG = b"\x89HDF\r\n\x1a\n"


# ------------------------------------------------------------------------------------------------------------------ reader
class _Reader:
    def __init__(self, buf):
        self.b = buf
        at = 0
        while self.b[at:at + 8] != SIG:          # the superblock may sit at 0, 512, 1024, ... (user block)
            at = 512 if at == 0 else at * 2
            if at + 8 > len(self.b):
                raise ValueError("not an HDF5 file (no superblock signature)")
        ver = self.b[at + 8]
        if ver not in (0, 1):
            raise NotImplementedError(f"HDF5 superblock version {ver} (libver='latest' files): only versions 0 and 1, which h5py writes by default, are read")
        so, sl = self.b[at + 13], self.b[at + 14]
        if (so, sl) != (8, 8):
            raise NotImplementedError(f"HDF5 offsets/lengths of {so}/{sl} bytes: only 8/8 is read")
        p = at + 24 + (4 if ver == 1 else 0)
        self.base = struct.unpack_from("<Q", self.b, p)[0]
        root_entry = p + 32                    # base, free-space, end-of-file, driver-info addresses, then the root symbol-table entry
        self.root = struct.unpack_from("<Q", self.b, root_entry + 8)[0]

    def u(self, fmt, at):
        return struct.unpack_from("<" + fmt, self.b, self.base + at)

    # -- object header (version 1) -> [(type, flags, payload bytes)]
    def messages(self, addr):
        a = self.base + addr
        if self.b[a:a + 4] == b"OHDR":
            raise NotImplementedError("version-2 object headers (a file written with libver='latest'): only version-1 headers are read")
        ver, _, nmsg, _refs, hsize = struct.unpack_from("<BBHII", self.b, a)
        if ver != 1:
            raise ValueError(f"object header version {ver} at {addr}")
        out, blocks, left = [], [(a + 16, hsize)], nmsg
        while blocks and left > 0:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and left > 0:
                mtype, msize, flags = struct.unpack_from("<HHB", self.b, p)
                body = self.b[p + 8:p + 8 + msize]
                p += 8 + msize
                left -= 1
                if mtype == 0x0010:            # continuation: (address, length) of another block of messages
                    ca, cl = struct.unpack_from("<QQ", body, 0)
                    blocks.append((self.base + ca, cl))
                else:
                    out.append((mtype, flags, body))
        return out

    # -- group traversal
    def _heap_name(self, heap_addr, off):
        a = self.base + heap_addr
        if self.b[a:a + 4] != b"HEAP":
            raise ValueError("local heap signature missing")
        data = struct.unpack_from("<Q", self.b, a + 24)[0]
        s = self.base + data + off
        e = self.b.index(b"\0", s)
        return self.b[s:e].decode("utf-8")

    def _btree_entries(self, addr, heap):
        a = self.base + addr
        sig = self.b[a:a + 4]
        if sig == b"SNOD":
            n = struct.unpack_from("<H", self.b, a + 6)[0]
            for i in range(n):
                e = a + 8 + 40 * i
                noff, oaddr = struct.unpack_from("<QQ", self.b, e)
                yield self._heap_name(heap, noff), oaddr
            return
        if sig != b"TREE":
            raise ValueError("B-tree / symbol-table node signature missing")
        ntype, _level, used = struct.unpack_from("<BBH", self.b, a + 4)
        if ntype != 0:
            raise ValueError("a group B-tree node was expected")
        p = a + 24                              # signature, type, level, entries used, left / right sibling
        for i in range(used):
            child = struct.unpack_from("<Q", self.b, p + 8 + 16 * i)[0]     # key0, child0, key1, child1, ...
            yield from self._btree_entries(child, heap)

    def children(self, addr):
        """{name: object header address} of a group; None for a dataset."""
        for mtype, _f, body in self.messages(addr):
            if mtype == 0x0011:
                bt, heap = struct.unpack_from("<QQ", body, 0)
                return dict(self._btree_entries(bt, heap))
            if mtype in (0x0002, 0x0006):
                raise NotImplementedError("new-style (link message / fractal heap) groups: written only with libver='latest'")
        return None

    # -- datatype / dataspace / data
    @staticmethod
    def _dtype(body):
        cls, ver = body[0] & 0x0F, body[0] >> 4
        bits0 = body[1]
        size = struct.unpack_from("<I", body, 4)[0]
        if ver not in (1, 2, 3):
            raise NotImplementedError(f"datatype message version {ver}")
        order = ">" if (bits0 & 1) else "<"
        if cls == 1:
            if size not in (2, 4, 8):
                raise NotImplementedError(f"{size}-byte floating point")
            return np.dtype(f"{order}f{size}")
        if cls == 0:
            signed = (bits0 >> 3) & 1
            return np.dtype(f"{order}{'i' if signed else 'u'}{size}")
        if cls == 3:
            return np.dtype(f"S{size}")
        if cls == 9:
            raise NotImplementedError("variable-length datatypes (vlen strings): Keras writes fixed-length byte strings")
        raise NotImplementedError(f"HDF5 datatype class {cls}")

    @staticmethod
    def _shape(body):
        ver, rank = body[0], body[1]
        if ver == 1:
            off = 8
        elif ver == 2:
            if body[3] == 2:                    # null dataspace
                return (0,)
            off = 4
        else:
            raise NotImplementedError(f"dataspace message version {ver}")
        return tuple(struct.unpack_from("<" + "Q" * rank, body, off)) if rank else ()

    def dataset(self, addr):
        dt = shape = None
        data = None
        for mtype, _f, body in self.messages(addr):
            if mtype == 0x0001:
                shape = self._shape(body)
            elif mtype == 0x0003:
                dt = self._dtype(body)
            elif mtype == 0x000B:
                raise NotImplementedError("filtered (compressed) datasets: Keras checkpoints are written without filters")
            elif mtype == 0x0008:
                ver = body[0]
                if ver != 3:
                    raise NotImplementedError(f"data layout message version {ver} (only version 3, what HDF5 >= 1.6.3 writes)")
                cls = body[1]
                if cls == 1:
                    a, n = struct.unpack_from("<QQ", body, 2)
                    data = None if a == UNDEF else (self.base + a, n)
                    if a == UNDEF:
                        data = (0, 0)
                elif cls == 0:
                    n = struct.unpack_from("<H", body, 2)[0]
                    data = bytes(body[4:4 + n])
                else:
                    raise NotImplementedError("chunked datasets: Keras checkpoints are contiguous")
        if dt is None or shape is None or data is None:
            raise ValueError("dataset without datatype / dataspace / layout")
        count = int(np.prod(shape)) if shape else 1
        if isinstance(data, tuple):
            a, n = data
            if n == 0:
                return np.zeros(shape, dt.newbyteorder("=") if dt.kind in "fiu" else dt)
            arr = np.frombuffer(self.b, dt, count, a)
        else:
            arr = np.frombuffer(data, dt, count)
        arr = arr.reshape(shape)
        return arr.astype(dt.newbyteorder("=")) if dt.kind in "fiu" else arr.copy()

    def attributes(self, addr):
        out = {}
        for mtype, _f, body in self.messages(addr):
            if mtype != 0x000C:
                continue
            ver = body[0]
            if ver not in (1, 2, 3):
                raise NotImplementedError(f"attribute message version {ver}")
            nsz, dsz, ssz = struct.unpack_from("<HHH", body, 2)
            p = 8 + (1 if ver == 3 else 0)
            pad = (lambda n: (n + 7) & ~7) if ver == 1 else (lambda n: n)
            name = bytes(body[p:p + nsz]).split(b"\0")[0].decode("utf-8")
            p += pad(nsz)
            dtb = body[p:p + dsz]
            p += pad(dsz)
            spb = body[p:p + ssz]
            p += pad(ssz)
            try:
                dt, shape = self._dtype(dtb), self._shape(spb)
            except NotImplementedError:
                out[name] = None                # e.g. a vlen string written by a newer Keras: not needed to find the weights
                continue
            count = int(np.prod(shape)) if shape else 1
            arr = np.frombuffer(bytes(body[p:p + count * dt.itemsize]), dt, count).reshape(shape)
            out[name] = arr.astype(dt.newbyteorder("=")) if dt.kind in "fiu" else arr.copy()
        return out


def read_h5(path):
    """-> ({'/group/.../dataset': ndarray}, {'/group' or '/group/dataset': {attribute: ndarray}}) of an HDF5 file."""
    with open(path, "rb") as f:
        r = _Reader(f.read())
    data, attrs = {}, {}

    def walk(addr, prefix, seen):
        if addr in seen:
            return
        seen = seen | {addr}
        a = r.attributes(addr)
        if a:
            attrs[prefix or "/"] = a
        kids = r.children(addr)
        if kids is None:
            data[prefix] = r.dataset(addr)
            return
        for name, child in kids.items():
            walk(child, prefix + "/" + name, seen)

    walk(r.root, "", frozenset())
    return data, attrs


def load_keras_weights(path):
    """Keras checkpoint -> {layer: (kernel, bias)} as sr355.wrappers._load_h5 returns it: every dataset called kernel* / bias*
    (`kernel:0`, `bias:0`) under `model_weights` (a `model.save` file) or under the root (a `save_weights` file), keyed by the group
    that holds it -- which is the layer's name also for nested models (`model_weights/vgg16/block1_conv1/kernel:0`)."""
    data, _ = read_h5(path)
    keys = [k for k in data if k.startswith("/model_weights/")] or list(data)
    out = {}
    for k in keys:
        parts = k.strip("/").split("/")
        leaf = parts[-1].split(":")[0]
        if leaf not in ("kernel", "bias") or len(parts) < 2:
            continue
        out.setdefault(parts[-2], [None, None])[0 if leaf == "kernel" else 1] = np.asarray(data[k], np.float32)
    missing = [n for n, (k, b) in out.items() if k is None]
    if missing:
        raise ValueError(f"{path}: layers without a kernel dataset: {missing[:5]}")
    return {n: (k, b if b is not None else np.zeros(k.shape[-1], np.float32)) for n, (k, b) in out.items()}


# ------------------------------------------------------------------------------------------------------------------ writer
def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == "f":
        exp, man = {2: (5, 10), 4: (8, 23), 8: (11, 52)}[dt.itemsize]
        bits = dt.itemsize * 8
        return struct.pack("<BBBBI", 0x11, 0x20, bits - 1, 0, dt.itemsize) + struct.pack("<HHBBBBI", 0, bits, man, exp, 0, man, (1 << (exp - 1)) - 1)
    if dt.kind in "iu":
        return struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0, 0, 0, dt.itemsize) + struct.pack("<HH", 0, dt.itemsize * 8)
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, dt.itemsize)          # null-padded ASCII, as h5py writes numpy 'S' arrays
    raise TypeError(f"dtype {dt} is not written")


def _space_msg(shape):
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)


def _attr_msg(name, value):
    a = np.asarray(value)
    if a.dtype.kind == "U":
        a = np.char.encode(a, "utf-8")
    if a.dtype.kind == "O":
        raise TypeError("object arrays are not written")
    nm = name.encode("utf-8") + b"\0"
    dtb, spb = _dtype_msg(a.dtype), _space_msg(a.shape)
    body = struct.pack("<BxHHH", 1, len(nm), len(dtb), len(spb)) + _pad8(nm) + _pad8(dtb) + _pad8(spb) + np.asarray(a, order="C").tobytes()
    return _msg(0x000C, body)


class _Writer:
    def __init__(self, leaf_k):
        self.buf = bytearray(96)               # superblock, filled in at the end
        self.leaf_k = leaf_k

    def put(self, b):
        self.buf += b"\0" * (-len(self.buf) % 8)
        at = len(self.buf)
        self.buf += b
        return at

    def header(self, msgs):
        body = b"".join(msgs)
        return self.put(struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body)

    def dataset(self, arr, attrs):
        arr = np.asarray(arr, order="C")          # (np.ascontiguousarray would turn a scalar into a 1-vector)
        if arr.dtype.kind in "fiu":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
        at = self.put(arr.tobytes()) if arr.size else UNDEF
        msgs = [_msg(0x0001, _space_msg(arr.shape)), _msg(0x0003, _dtype_msg(arr.dtype), flags=1),
                _msg(0x0005, struct.pack("<BBBBI", 2, 2, 2, 1, 0)),                         # fill value: version 2, late allocation, default value
                _msg(0x0008, struct.pack("<BBQQ", 3, 1, at, arr.nbytes))]
        msgs += [_attr_msg(k, v) for k, v in (attrs or {}).items()]
        return self.header(msgs)

    def group(self, entries, attrs):
        """entries: {name: object header address}.  One local heap, symbol-table nodes of up to 2 * leaf_k sorted entries, one B-tree level."""
        names = sorted(entries, key=lambda s: s.encode("utf-8"))
        heap, offs = bytearray(8), {}
        for n in names:
            offs[n] = len(heap)
            heap += _pad8(n.encode("utf-8") + b"\0")
        heap_data = self.put(bytes(heap))
        # free-list head 1 = "no free block" (libhdf5's H5HL_FREE_NULL; the library rejects any other value >= the segment size)
        heap_at = self.put(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), 1, heap_data))
        per = 2 * self.leaf_k
        chunks = [names[i:i + per] for i in range(0, len(names), per)] or [[]]
        if len(chunks) > 32:
            raise ValueError("too many entries for a one-level B-tree: raise leaf_k")
        snods, keys = [], [0]
        for ch in chunks:
            body = b"SNOD" + struct.pack("<BBH", 1, 0, len(ch))
            for n in ch:
                body += struct.pack("<QQII16x", offs[n], entries[n], 0, 0)
            body += b"\0" * (40 * (per - len(ch)))
            snods.append(self.put(body))
            keys.append(offs[ch[-1]] if ch else 0)
        tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods), UNDEF, UNDEF) + struct.pack("<Q", keys[0])
        for s_at, k in zip(snods, keys[1:]):
            tree += struct.pack("<QQ", s_at, k)
        tree += b"\0" * (16 * (32 - len(snods)))
        bt = self.put(tree)
        msgs = [_msg(0x0011, struct.pack("<QQ", bt, heap_at))] + [_attr_msg(k, v) for k, v in (attrs or {}).items()]
        return self.header(msgs), bt, heap_at


def write_h5(path, datasets, attrs=None, leaf_k=64):
    """datasets: {'/a/b/name': ndarray}; attrs: {'/a' | '/a/b/name' | '/': {attribute: array-like}}.  Version-0 superblock, version-1 object
    headers, symbol-table groups, contiguous little-endian datasets -- the structures h5py's defaults produce (module docstring)."""
    attrs = attrs or {}
    tree = {}
    for k, v in datasets.items():
        node = tree
        parts = k.strip("/").split("/")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
            if not isinstance(node, dict):
                raise ValueError(f"{k}: a dataset is in the way")
        node[parts[-1]] = np.asarray(v)
    w = _Writer(leaf_k)

    def emit(node, prefix):
        entries = {}
        for name, child in node.items():
            full = prefix + "/" + name
            entries[name] = emit(child, full)[0] if isinstance(child, dict) else w.dataset(child, attrs.get(full))
        return w.group(entries, attrs.get(prefix or "/"))

    root, bt, heap = emit(tree, "")
    eof = len(w.buf) + (-len(w.buf) % 8)
    w.buf += b"\0" * (eof - len(w.buf))
    sb = SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, leaf_k, 16, 0) + struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", bt, heap)
    assert len(sb) == 96
    w.buf[:96] = sb
    with open(path, "wb") as f:
        f.write(bytes(w.buf))


def save_keras_weights(path, weights, model_name=None):
    """{layer: (kernel, bias)} -> a file in Keras' `model.save` weight layout: /model_weights/<layer>/<layer>/kernel:0 and bias:0 with the
    `layer_names` / `weight_names` / `backend` / `keras_version` attributes Keras' own loader walks (hdf5_format.load_weights_from_hdf5_group).
    WEIGHTS ONLY, FLAT NAMES: the file holds no `model_config` / `training_config`, so the reference's `load_model(path)` (ESRGAN_model.py:143-149,
    SRCNN_model.py:35) cannot open it; it is read by this package's loader and by Keras' `model.load_weights(path)` on a model whose layers carry
    these flat names.  The reference's discriminator wraps its layers in tfa SpectralNormalization (groups `spectral_normalization*/...` with a
    `sn_u` vector) and its classifier nests the base under `vgg16/`: neither nesting is emitted here."""
    ds, at = {}, {"/model_weights": {"layer_names": np.array([n.encode() for n in weights], dtype="S"), "backend": np.array(b"tensorflow"),
                                     "keras_version": np.array(b"2.10.0")}}
    for n, (k, b) in weights.items():
        ds[f"/model_weights/{n}/{n}/kernel:0"] = np.asarray(k, np.float32)
        ds[f"/model_weights/{n}/{n}/bias:0"] = np.asarray(b, np.float32)
        at[f"/model_weights/{n}"] = {"weight_names": np.array([f"{n}/kernel:0".encode(), f"{n}/bias:0".encode()], dtype="S")}
    if model_name:
        at["/"] = {"keras_version": np.array(b"2.10.0"), "backend": np.array(b"tensorflow")}
    write_h5(path, ds, at)
