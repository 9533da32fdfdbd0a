"""A small, dependency-free reader and writer for the subset of HDF5 that Keras weight files use.

Why: the reference checkpoints its two MLPs with `tf.keras.Model.save_weights('coarse.h5')` / `load_weights`
(reference keras_nerf/model/nerf/nerf.py:63-64, 132-136), i.e. Keras' HDF5 weight format, and neither `h5py` nor TensorFlow is
a dependency of this package.  What such a file contains (Keras `save_weights_to_hdf5_group`, h5py with its default
"earliest" library version bounds):

  superblock version 0, 8-byte offsets/lengths                      (HDF5 File Format Specification, section II.A)
  old-style groups: symbol-table message -> v1 B-tree + local heap + SNOD nodes            (III.A, III.B, III.D)
  version-1 object headers with continuation blocks                                         (IV.A.1, message 0x0010)
  datasets: simple dataspace, IEEE float datatype, contiguous (or compact) layout           (IV.A.2.b/d/i)
  attributes `layer_names` / `weight_names` (fixed-length string arrays), `backend`, `keras_version`   (IV.A.2.m)

The reader walks groups and reads float datasets (attributes are not needed to locate Dense kernels and biases and are
skipped, so variable-length string attributes and the global heap never come into play).  It also accepts superblock
versions 2 and 3, version-2 object headers and link-message ("compact") groups, so files written with `libver='latest'`
load too; chunked, filtered or virtual datasets and dense (fractal-heap) groups raise `Hdf5FormatError` with the feature
named.  The writer emits the "earliest" format above, byte-compatible with what libhdf5 itself reads (validated against
libhdf5 1.10's `h5dump`/`h5ls`, tests/test_hdf5_min.py); it writes fixed-length string attributes, which Keras' loader
decodes the same way as the ones h5py writes.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5FormatError(ValueError):
    pass


def is_hdf5(path: str) -> bool:
    """True when the file starts with the HDF5 signature (at offset 0, the only place h5py puts it)"""
    with open(path, "rb") as f:
        return f.read(8) == SIGNATURE


# =====================================================================================================================
# reader
# =====================================================================================================================
class _Dataset:
    def __init__(self, shape, dtype, data):
        self.shape, self.dtype, self._data = tuple(shape), dtype, data

    def __array__(self, dtype=None, copy=None):
        a = np.frombuffer(self._data, dtype=self.dtype, count=int(np.prod(self.shape, dtype=np.int64))).reshape(self.shape)
        return a.astype(dtype) if dtype is not None else a


class _Group(dict):
    """name -> _Group | _Dataset"""

    def visit_datasets(self, prefix=""):
        for k in sorted(self):
            v = self[k]
            if isinstance(v, _Group):
                yield from v.visit_datasets(prefix + k + "/")
            else:
                yield prefix + k, v


class Hdf5Reader:
    def __init__(self, path: str):
        with open(path, "rb") as f:
            self.buf = f.read()
        if self.buf[:8] != SIGNATURE:
            raise Hdf5FormatError(f"{path}: not an HDF5 file (signature missing)")
        # The file is untrusted input: every address is bounds-checked (_u), object / B-tree / continuation references that
        # loop back are refused (_object's path set, _MAX_DEPTH, _MAX_BLOCKS), and whatever a damaged structure still trips
        # (a bad length in a message, a name without terminator, ...) surfaces as Hdf5FormatError, never as a bare IndexError,
        # a short silent read or an endless recursion.
        self._path, self._done = set(), {}
        try:
            self._superblock()
            self.root = self._object(self.root_addr)
        except Hdf5FormatError:
            raise
        except (IndexError, ValueError, struct.error, RecursionError, UnicodeDecodeError, OverflowError, MemoryError) as e:
            raise Hdf5FormatError(f"{path}: damaged or unsupported HDF5 structure ({type(e).__name__}: {e})") from e
        if not isinstance(self.root, _Group):
            raise Hdf5FormatError("root object is not a group")

    _MAX_DEPTH = 64            # nesting of groups (a Keras weight file has 4)
    _MAX_BLOCKS = 4096         # object-header continuation blocks / B-tree nodes followed per object

    # ---- primitives
    def _u(self, off, n):
        if off < 0 or off + n > len(self.buf):
            raise Hdf5FormatError(f"address {off} (+{n}) lies outside the file ({len(self.buf)} bytes)")
        return int.from_bytes(self.buf[off:off + n], "little")

    def _addr(self, off):
        return self._u(off, self.so)

    def _superblock(self):
        b, ver = self.buf, self.buf[8]
        if ver in (0, 1):
            self.so, self.sl = b[13], b[14]
            self.leaf_k = self._u(16, 2)
            off = 24 + (4 if ver == 1 else 0)
            self.base = self._u(off, self.so)
            entry = off + 4 * self.so                       # base, free-space, eof, driver -> root symbol-table entry
            self.root_addr = self._u(entry + self.so, self.so)
        elif ver in (2, 3):
            self.so, self.sl = b[9], b[10]
            self.base = self._u(12, self.so)
            self.root_addr = self._u(12 + 3 * self.so, self.so)
        else:
            raise Hdf5FormatError(f"superblock version {ver} is not supported")
        if self.so != 8 or self.sl != 8:
            raise Hdf5FormatError(f"{self.so}-byte offsets / {self.sl}-byte lengths are not supported (expected 8/8)")
        if self.base != 0:
            raise Hdf5FormatError("non-zero base address (user block) is not supported")

    # ---- object headers
    def _messages(self, addr):
        """list of (type, flags, data bytes) of the object header at addr, following continuation blocks"""
        b = self.buf
        msgs = []
        if b[addr:addr + 4] == b"OHDR":                      # version 2
            flags = b[addr + 5]
            p = addr + 6
            if flags & 0x20:
                p += 16                                      # access / modification / change / birth times
            if flags & 0x10:
                p += 4                                       # max compact / min dense attributes
            csz = 1 << (flags & 3)
            chunk0 = self._u(p, csz); p += csz
            tracked = bool(flags & 0x04)
            blocks = [(p, chunk0)]
            seen = 0
            while blocks:
                p, size = blocks.pop(0)
                seen += 1
                if seen > self._MAX_BLOCKS or p < 0 or p + size > len(b):
                    raise Hdf5FormatError("object header continuation chain is cyclic or leaves the file")
                end = p + size
                while p + 4 <= end - 0:
                    t = b[p]; sz = self._u(p + 1, 2); fl = b[p + 3]; p += 4
                    if tracked:
                        p += 2
                    if p + sz > end:
                        break
                    data = b[p:p + sz]; p += sz
                    if t == 0x10:
                        caddr, clen = struct.unpack("<QQ", data[:16])
                        if b[caddr:caddr + 4] != b"OCHK":
                            raise Hdf5FormatError("bad object header continuation block")
                        blocks.append((caddr + 4, clen - 8))         # signature in front, checksum behind
                    elif t != 0:
                        msgs.append((t, fl, data))
            return msgs
        if b[addr] != 1:
            raise Hdf5FormatError(f"object header version {b[addr]} at {addr} is not supported")
        n_msgs = self._u(addr + 2, 2)
        size = self._u(addr + 8, 4)
        blocks = [(addr + 16, size)]
        seen = 0
        while blocks and len(msgs) < n_msgs + 64:
            p, size = blocks.pop(0)
            seen += 1
            if seen > self._MAX_BLOCKS or p < 0 or p + size > len(b):
                raise Hdf5FormatError("object header continuation chain is cyclic or leaves the file")
            end = p + size
            while p + 8 <= end:
                t, sz, fl = self._u(p, 2), self._u(p + 2, 2), b[p + 4]
                data = b[p + 8:p + 8 + sz]
                p += 8 + sz
                if t == 0x10:
                    caddr, clen = struct.unpack("<QQ", data[:16])
                    blocks.append((caddr, clen))
                elif t != 0:
                    msgs.append((t, fl, data))
        return msgs

    def _object(self, addr):
        """the group or dataset whose object header sits at addr; an address that is already on the path from the root (a
        group that contains itself) is refused, one that was reached by another path (a hard link) is returned again"""
        if addr in self._done:
            return self._done[addr]
        if addr in self._path or len(self._path) >= self._MAX_DEPTH:
            raise Hdf5FormatError(f"group structure is cyclic or nested deeper than {self._MAX_DEPTH} (object header at {addr})")
        if addr < 0 or addr >= len(self.buf):
            raise Hdf5FormatError(f"object header address {addr} lies outside the file")
        self._path.add(addr)
        try:
            obj = self._done[addr] = self._object_at(addr)
        finally:
            self._path.discard(addr)
        return obj

    def _object_at(self, addr):
        msgs = self._messages(addr)
        types = {t for t, _, _ in msgs}
        if 0x11 in types:                                    # symbol table: old-style group
            data = next(d for t, _, d in msgs if t == 0x11)
            btree, heap = struct.unpack("<QQ", data[:16])
            return self._old_group(btree, heap)
        if 0x08 in types:
            return self._dataset(msgs)
        if 0x02 in types or 0x06 in types or not (types - {0x0C, 0x12, 0x0A, 0x15, 0x16}):   # link info / link messages / empty group
            g = _Group()
            for t, _, d in msgs:
                if t == 0x02 and len(d) >= 2:
                    fl = d[1]
                    p = 2 + (8 if fl & 1 else 0)
                    fheap = int.from_bytes(d[p:p + 8], "little")
                    if fheap != UNDEF:
                        raise Hdf5FormatError("dense link storage (fractal heap) is not supported; re-save the file with fewer links per group")
                if t == 0x06:
                    name, target = self._link(d)
                    if target is not None:
                        g[name] = self._object(target)
            return g
        raise Hdf5FormatError(f"object at {addr}: unsupported message set {sorted(types)}")

    def _link(self, d):
        fl = d[1]
        p = 2
        ltype = 0
        if fl & 0x08:
            ltype = d[p]; p += 1
        if fl & 0x04:
            p += 8
        if fl & 0x10:
            p += 1
        nsz = 1 << (fl & 3)
        nlen = int.from_bytes(d[p:p + nsz], "little"); p += nsz
        name = d[p:p + nlen].decode("utf8"); p += nlen
        if ltype != 0:
            return name, None                                # soft / external links: ignored
        return name, int.from_bytes(d[p:p + 8], "little")

    # ---- old-style groups
    def _heap_name(self, heap_data_addr, off):
        if heap_data_addr + off >= len(self.buf):
            raise Hdf5FormatError("link name offset lies outside the file")
        end = self.buf.index(b"\x00", heap_data_addr + off)
        return self.buf[heap_data_addr + off:end].decode("utf8")

    def _old_group(self, btree, heap):
        b = self.buf
        if b[heap:heap + 4] != b"HEAP":
            raise Hdf5FormatError("bad local heap signature")
        heap_data = self._u(heap + 8 + 2 * self.sl, self.so)
        g = _Group()
        for snod in self._btree_leaves(btree):
            if b[snod:snod + 4] != b"SNOD":
                raise Hdf5FormatError("bad symbol table node signature")
            n = self._u(snod + 6, 2)
            p = snod + 8
            for _ in range(n):
                name_off, ohdr, cache = self._u(p, 8), self._u(p + 8, 8), self._u(p + 16, 4)
                if cache != 2:                               # cache type 2 = symbolic link: its "object header address" is not one
                    g[self._heap_name(heap_data, name_off)] = self._object(ohdr)
                p += 40
        return g

    def _btree_leaves(self, addr, above=256, budget=None):
        b = self.buf
        if addr == UNDEF:
            return
        budget = budget if budget is not None else [self._MAX_BLOCKS]
        budget[0] -= 1
        if budget[0] < 0 or addr < 0 or addr + 8 > len(b):
            raise Hdf5FormatError("group B-tree is cyclic or leaves the file")
        if b[addr:addr + 4] != b"TREE":
            raise Hdf5FormatError("bad B-tree node signature")
        if b[addr + 4] != 0:
            raise Hdf5FormatError("chunked-dataset B-tree where a group B-tree was expected")
        level, used = b[addr + 5], self._u(addr + 6, 2)
        if level >= above:
            raise Hdf5FormatError("group B-tree levels do not decrease towards the leaves")
        p = addr + 8 + 2 * self.so
        for i in range(used):
            child = self._u(p + self.sl, self.so)           # key_i, child_i, key_i+1, ...
            p += self.sl + self.so
            if level == 0:
                yield child
            else:
                yield from self._btree_leaves(child, level, budget)

    # ---- datasets
    def _dataset(self, msgs) -> _Dataset:
        shape = dtype = None
        layout = None
        for t, _, d in msgs:
            if t == 0x01:                                    # dataspace
                ver, rank = d[0], d[1]
                if ver == 1:
                    shape = struct.unpack(f"<{rank}Q", d[8:8 + 8 * rank])
                elif ver == 2:
                    if d[3] == 2:
                        raise Hdf5FormatError("null dataspace")
                    shape = struct.unpack(f"<{rank}Q", d[4:4 + 8 * rank])
                else:
                    raise Hdf5FormatError(f"dataspace message version {ver}")
            elif t == 0x03:                                  # datatype
                cls, size = d[0] & 0x0F, struct.unpack("<I", d[4:8])[0]
                big = bool(d[1] & 1)
                if cls == 1 and size in (2, 4, 8):
                    dtype = np.dtype((">" if big else "<") + "f" + str(size))
                elif cls == 0 and size in (1, 2, 4, 8):
                    dtype = np.dtype((">" if big else "<") + ("i" if d[1] & 8 else "u") + str(size))
                else:
                    raise Hdf5FormatError(f"datatype class {cls} size {size} is not supported (float / integer datasets only)")
            elif t == 0x08:
                layout = d
            elif t == 0x0B:
                raise Hdf5FormatError("filtered (compressed) datasets are not supported; Keras writes weights uncompressed")
        if shape is None or dtype is None or layout is None:
            raise Hdf5FormatError("dataset without dataspace / datatype / layout message")
        n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        ver = layout[0]
        if ver == 3 or ver == 4:
            cls = layout[1]
            if cls == 1:
                addr, size = struct.unpack("<QQ", layout[2:18])
                data = b"\x00" * n if addr == UNDEF else self.buf[addr:addr + n]
            elif cls == 0:
                size = struct.unpack("<H", layout[2:4])[0]
                data = layout[4:4 + size]
            else:
                raise Hdf5FormatError("chunked / virtual dataset layout is not supported; Keras writes weights contiguous")
        elif ver in (1, 2):
            rank, cls = layout[1], layout[2]
            if cls != 1:
                raise Hdf5FormatError("only contiguous datasets are supported in layout versions 1/2")
            addr = struct.unpack("<Q", layout[8:16])[0]
            data = self.buf[addr:addr + n]
        else:
            raise Hdf5FormatError(f"data layout message version {ver}")
        if len(data) < n:
            raise Hdf5FormatError("dataset extends past the end of the file")
        return _Dataset(shape, dtype, data)

    # ---- convenience
    def datasets(self) -> Dict[str, np.ndarray]:
        return {k: np.asarray(v) for k, v in self.root.visit_datasets()}


def read_keras_weights(path: str, layer_names: Sequence[str]) -> List[np.ndarray]:
    """[kernel, bias] per layer in `layer_names` order from a Keras HDF5 weight file: /<layer>/.../kernel:0 and bias:0
    (the intermediate groups carry the variable's name scope, e.g. coarse_nerf/layer_0, which differs between Keras
    versions and between subclassed and functional models; any nesting is accepted)."""
    root = Hdf5Reader(path).root
    if "model_weights" in root and isinstance(root["model_weights"], _Group):      # a full-model .h5 (model.save)
        root = root["model_weights"]
    out = []
    for name in layer_names:
        g = root.get(name)
        if not isinstance(g, _Group):
            raise KeyError(f"{path}: no group '{name}' (found {sorted(root)})")
        found = dict(g.visit_datasets())
        k = [v for n, v in found.items() if n.split("/")[-1].startswith("kernel")]
        b = [v for n, v in found.items() if n.split("/")[-1].startswith("bias")]
        if len(k) != 1 or len(b) != 1:
            raise KeyError(f"{path}: group '{name}' should hold one kernel and one bias, found {sorted(found)}")
        out += [np.asarray(k[0], dtype=np.float32), np.asarray(b[0], dtype=np.float32)]
    return out


# =====================================================================================================================
# writer ("earliest" format: superblock 0, v1 object headers, symbol-table groups, contiguous datasets)
# =====================================================================================================================
_LEAF_K = 16           # symbol-table nodes hold up to 2K = 32 entries: every group of a weight file fits one node


def _pad8(b: bytes) -> bytes:
    return b + b"\x00" * (-len(b) % 8)


def _msg(mtype: int, data: bytes, flags: int = 0) -> bytes:
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _dataspace(shape) -> bytes:
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(s)) for s in shape)


def _f32_type() -> bytes:
    return struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)


def _str_type(n: int) -> bytes:
    return struct.pack("<BBBBI", 0x13, 0x01, 0x00, 0x00, n)          # fixed length, null padded, ASCII


def _attr(name: str, values: Sequence[bytes], scalar: bool = False) -> bytes:
    """attribute message (version 1) holding fixed-length strings: a 1-D array, or a scalar"""
    width = max([len(v) for v in values] + [1])
    nm = name.encode() + b"\x00"
    dt, ds = _str_type(width), (struct.pack("<BBB5x", 1, 0, 0) if scalar else _dataspace([len(values)]))
    body = struct.pack("<BxHHH", 1, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds)
    body += b"".join(v.ljust(width, b"\x00") for v in values)
    return _msg(0x0C, body)


def _ohdr(messages: Sequence[bytes]) -> bytes:
    body = b"".join(messages)
    return struct.pack("<BxHII4x", 1, len(messages), 1, len(body)) + body


class _Node:
    def __init__(self):
        self.children: Dict[str, Union["_Node", np.ndarray]] = {}
        self.attrs: List[bytes] = []


class Hdf5Writer:
    """Build a tree of groups / float32 datasets / string attributes, then write() it.

        w = Hdf5Writer(); w.dataset("layer_0/coarse_nerf/layer_0/kernel:0", array); w.attr("", "layer_names", [b"layer_0"])
    """

    def __init__(self):
        self.root = _Node()

    def _node(self, path: str, create=True) -> _Node:
        n = self.root
        for part in [p for p in path.split("/") if p]:
            if part not in n.children:
                n.children[part] = _Node()
            n = n.children[part]
            if not isinstance(n, _Node):
                raise ValueError(f"{path}: a dataset is in the way")
        return n

    def group(self, path: str):
        self._node(path)

    def dataset(self, path: str, array):
        parts = [p for p in path.split("/") if p]
        self._node("/".join(parts[:-1])).children[parts[-1]] = np.ascontiguousarray(array, dtype="<f4")

    def attr(self, path: str, name: str, values, scalar: bool = False):
        vals = [values] if scalar else list(values)
        self._node(path).attrs.append(_attr(name, [v if isinstance(v, bytes) else str(v).encode() for v in vals], scalar))

    def write(self, path: str):
        chunks: List[bytes] = []
        pos = [96]                                           # behind the superblock

        def alloc(data: bytes) -> int:
            a = pos[0]
            data = _pad8(data)
            chunks.append(data); pos[0] += len(data)
            return a

        def emit_dataset(arr: np.ndarray) -> int:
            raw = arr.tobytes()
            data_addr = alloc(raw) if raw else UNDEF
            msgs = [_msg(0x01, _dataspace(arr.shape)), _msg(0x03, _f32_type(), flags=1),
                    _msg(0x05, struct.pack("<BBBB", 2, 2, 2, 0)),                   # fill value v2: late alloc, never written, undefined
                    _msg(0x08, struct.pack("<BBQQ", 3, 1, data_addr, len(raw)))]
            return alloc(_ohdr(msgs))

        def emit_group(node: _Node) -> Tuple[int, int, int]:
            """returns (object header address, b-tree address, heap address)"""
            names = sorted(node.children, key=lambda s: s.encode())
            if len(names) > 2 * _LEAF_K:
                raise ValueError(f"more than {2 * _LEAF_K} links in one group")
            entries = []
            for nm in names:
                ch = node.children[nm]
                if isinstance(ch, _Node):
                    oh, bt, hp = emit_group(ch)
                    entries.append((nm, oh, 1, struct.pack("<QQ", bt, hp)))
                else:
                    entries.append((nm, emit_dataset(ch), 0, b"\x00" * 16))
            # local heap: offset 0 = "" (8 bytes), then the names, each padded to 8
            heap_data, offs = bytearray(b"\x00" * 8), {}
            for nm in names:
                offs[nm] = len(heap_data)
                heap_data += _pad8(nm.encode() + b"\x00")
            free_off = len(heap_data)
            heap_data += struct.pack("<QQ", 1, 16)               # one free block: next = 1 (none), size 16
            seg_addr = alloc(bytes(heap_data))
            heap_addr = alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free_off, seg_addr))
            snod = b"SNOD" + struct.pack("<BxH", 1, len(entries))
            for nm, oh, cache, scratch in entries:
                snod += struct.pack("<QQI4x", offs[nm], oh, cache) + scratch
            snod += b"\x00" * (40 * (2 * _LEAF_K - len(entries)))
            if entries:
                snod_addr = alloc(snod)
                tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod_addr, offs[names[-1]])
            else:
                tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 0, UNDEF, UNDEF)
            tree += b"\x00" * (24 + (2 * 16 + 1) * 8 + 2 * 16 * 8 - len(tree))   # node sized for internal K = 16: 33 keys, 32 children
            bt_addr = alloc(tree)
            oh = alloc(_ohdr([_msg(0x11, struct.pack("<QQ", bt_addr, heap_addr))] + node.attrs))
            return oh, bt_addr, heap_addr

        root_oh, root_bt, root_heap = emit_group(self.root)
        eof = pos[0]
        sb = SIGNATURE + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", _LEAF_K, 16, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQI4x", 0, root_oh, 1) + struct.pack("<QQ", root_bt, root_heap)
        assert len(sb) == 96
        with open(path, "wb") as f:
            f.write(sb)
            for c in chunks:
                f.write(c)


def write_keras_weights(path: str, model_name: str, layer_names: Sequence[str], weights: Sequence[np.ndarray],
                        keras_version: str = "2.9.0"):
    """Keras `save_weights` layout for a subclassed model of Dense layers: root attributes layer_names / backend /
    keras_version; one group per layer with attribute weight_names = [<model>/<layer>/kernel:0, <model>/<layer>/bias:0] and the
    two float32 datasets under those paths; an empty top_level_model_weights group (Keras >= 2.8)."""
    assert len(weights) == 2 * len(layer_names)
    w = Hdf5Writer()
    w.attr("", "layer_names", [n.encode() for n in layer_names])
    w.attr("", "backend", b"tensorflow", scalar=True)
    w.attr("", "keras_version", keras_version.encode(), scalar=True)
    for i, name in enumerate(layer_names):
        kn, bn = f"{model_name}/{name}/kernel:0", f"{model_name}/{name}/bias:0"
        w.attr(name, "weight_names", [kn.encode(), bn.encode()])
        w.dataset(f"{name}/{kn}", weights[2 * i])
        w.dataset(f"{name}/{bn}", weights[2 * i + 1])
    w.group("top_level_model_weights")
    w.write(path)
