"""Minimal HDF5 reader / writer for the files of the reference's I/O contract.

The reference reads its spectra and writes ``params.h5`` / ``flux_model.h5`` with h5py
(vpspectrum.py:260-266, 528-538); h5py is not available here, and these files need very little of
HDF5: one root group holding contiguous, uncompressed numeric datasets.  That subset is what this
module speaks, in the same on-disk form h5py's defaults produce (checked against the reference's
own ``simba_*.h5`` / ``ramses_ray.h5``, which the reader parses structure by structure):

  superblock version 0 -> root symbol-table entry -> version-1 B-tree of the group ("TREE") ->
  symbol-table node ("SNOD") + local heap ("HEAP") for the names -> version-1 object headers with
  dataspace (v1), datatype (IEEE float / fixed point), fill value, contiguous layout (v3) messages.

``read(path)`` returns {name: ndarray} for the datasets of the root group (sub-groups are
followed, names joined by '/').  ``write(path, {name: array})`` creates such a file: float64,
float32, int64, int32 and bool (stored as int8) arrays of any rank, including scalars.
Chunked / compressed / compact layouts, attributes and other datatype classes are outside the
subset: the reader raises on them.
"""
from __future__ import annotations

import struct
import time

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5FormatError(RuntimeError):
    pass


# ---------------------------------------------------------------------------------------------
# reader
# ---------------------------------------------------------------------------------------------
class _Reader:
    def __init__(self, buf):
        self.b = buf
        if buf[:8] != SIGNATURE:
            raise H5FormatError("not an HDF5 file")
        ver = buf[8]
        if ver not in (0, 1):
            raise H5FormatError("superblock version %d is outside the supported subset (0, 1)" % ver)
        if buf[13] != 8 or buf[14] != 8:
            raise H5FormatError("only 8-byte offsets and lengths are supported")
        self.leaf_k, self.int_k = struct.unpack_from("<HH", buf, 16)
        p = 24 + (4 if ver == 1 else 0)
        self.base = struct.unpack_from("<Q", buf, p)[0]
        self.root_entry = p + 32           # after base, free-space, end-of-file, driver-info addresses

    # -- groups --------------------------------------------------------------------------------
    def symbol_entry(self, p):
        name_off, ohdr, cache = struct.unpack_from("<QQI", self.b, p)
        scratch = self.b[p + 24:p + 40]
        return name_off, ohdr, cache, scratch

    def group_members(self, btree, heap):
        if self.b[heap:heap + 4] != b"HEAP":
            raise H5FormatError("local heap signature missing")
        data_addr = struct.unpack_from("<Q", self.b, heap + 24)[0]
        out = []

        def walk(node):
            if self.b[node:node + 4] != b"TREE":
                raise H5FormatError("B-tree signature missing")
            ntype, level, used = struct.unpack_from("<BBH", self.b, node + 4)
            if ntype != 0:
                raise H5FormatError("not a group B-tree")
            p = node + 24
            for i in range(used):
                child = struct.unpack_from("<Q", self.b, p + 8)[0]
                if level > 0:
                    walk(child)
                else:
                    if self.b[child:child + 4] != b"SNOD":
                        raise H5FormatError("symbol-table node signature missing")
                    n = struct.unpack_from("<H", self.b, child + 6)[0]
                    for j in range(n):
                        name_off, ohdr, cache, scratch = self.symbol_entry(child + 8 + 40 * j)
                        s = data_addr + name_off
                        name = self.b[s:self.b.index(b"\0", s)].decode("utf-8")
                        out.append((name, ohdr, cache, scratch))
                p += 16

        walk(btree)
        return out

    # -- object headers ---------------------------------------------------------------------------
    def messages(self, addr):
        ver, _, nmsg, _, hsize = struct.unpack_from("<BBHII", self.b, addr)
        if ver != 1:
            raise H5FormatError("object header version %d is outside the supported subset (1)" % ver)
        blocks = [(addr + 16, addr + 16 + hsize)]
        out = []
        while blocks and len(out) < nmsg:
            p, end = blocks.pop(0)
            while p + 8 <= end and len(out) < nmsg:
                mtype, size, flags = struct.unpack_from("<HHB", self.b, p)
                data = self.b[p + 8:p + 8 + size]
                out.append((mtype, flags, data))
                if mtype == 0x10:                                   # continuation
                    off, length = struct.unpack_from("<QQ", data)
                    blocks.append((off, off + length))
                p += 8 + size
        return out

    def dataset(self, msgs):
        shape = dtype = None
        addr = size = None
        for mtype, _, d in msgs:
            if mtype == 0x01:
                ver, rank, flags = d[0], d[1], d[2]
                off = 8 if ver == 1 else 4
                shape = struct.unpack_from("<%dQ" % rank, d, off) if rank else ()
            elif mtype == 0x03:
                cls, bits0, nbytes = d[0] & 0x0F, d[1], struct.unpack_from("<I", d, 4)[0]
                order = ">" if bits0 & 1 else "<"
                if cls == 1:
                    dtype = np.dtype(order + "f%d" % nbytes)
                elif cls == 0:
                    dtype = np.dtype(order + ("i" if bits0 & 8 else "u") + "%d" % nbytes)
                else:
                    raise H5FormatError("datatype class %d is outside the supported subset (integers, floats)" % cls)
            elif mtype == 0x08:
                ver = d[0]
                if ver == 3:
                    if d[1] != 1:
                        raise H5FormatError("only contiguous datasets are supported (layout class %d)" % d[1])
                    addr, size = struct.unpack_from("<QQ", d, 2)
                elif ver in (1, 2):
                    rank, cls = d[1], d[2]
                    if cls != 1:
                        raise H5FormatError("only contiguous datasets are supported (layout class %d)" % cls)
                    addr = struct.unpack_from("<Q", d, 8)[0]
                else:
                    raise H5FormatError("layout message version %d is not supported" % ver)
            elif mtype == 0x0B:
                raise H5FormatError("filtered (compressed) datasets are not supported")
        if shape is None or dtype is None or addr is None:
            return None
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if addr == UNDEF:
            return np.zeros(shape, dtype=dtype.newbyteorder("="))   # never written: fill value
        a = np.frombuffer(self.b, dtype, count, self.base + addr).reshape(shape)
        return a.astype(dtype.newbyteorder("="))

    def read_group(self, btree, heap, prefix, out):
        for name, ohdr, cache, scratch in self.group_members(btree, heap):
            if cache == 1:                                          # a group, addresses cached in the entry
                bt, hp = struct.unpack_from("<QQ", scratch)
                self.read_group(bt, hp, prefix + name + "/", out)
                continue
            msgs = self.messages(self.base + ohdr)
            stab = [d for t, _, d in msgs if t == 0x11]
            if stab:
                bt, hp = struct.unpack_from("<QQ", stab[0])
                self.read_group(bt, hp, prefix + name + "/", out)
                continue
            arr = self.dataset(msgs)
            if arr is not None:
                out[prefix + name] = arr

    def read(self):
        name_off, ohdr, cache, scratch = self.symbol_entry(self.root_entry)
        if cache == 1:
            bt, hp = struct.unpack_from("<QQ", scratch)
        else:
            stab = [d for t, _, d in self.messages(self.base + ohdr) if t == 0x11]
            if not stab:
                raise H5FormatError("the root group has no symbol table (new-style groups are not supported)")
            bt, hp = struct.unpack_from("<QQ", stab[0])
        out = {}
        self.read_group(bt, hp, "", out)
        return out


def read(path):
    """{name: ndarray} of every dataset in the file (see the module docstring for the subset)."""
    with open(path, "rb") as fh:
        return _Reader(fh.read()).read()


# ---------------------------------------------------------------------------------------------
# writer
# ---------------------------------------------------------------------------------------------
def _pad8(n):
    return (n + 7) & ~7


def _datatype_message(dt):
    if dt.kind == "f":
        if dt.itemsize == 8:      # IEEE double, little endian: exponent at bit 52 (11 bits), bias 1023
            body = struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 0x3F, 0x00, 8, 0, 64, 52, 11, 0, 52, 1023)
        elif dt.itemsize == 4:
            body = struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 0x1F, 0x00, 4, 0, 32, 23, 8, 0, 23, 127)
        else:
            raise TypeError("unsupported float size")
    elif dt.kind in "iu":
        body = struct.pack("<BBBBIHH", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize, 0, 8 * dt.itemsize)
    else:
        raise TypeError("unsupported dtype %s" % dt)
    return body + b"\0" * (_pad8(len(body)) - len(body))


def _message(mtype, body, flags=0):
    body = body + b"\0" * (_pad8(len(body)) - len(body))
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _dataset_header(arr, data_addr, mtime):
    rank = arr.ndim
    dims = struct.pack("<%dQ" % rank, *arr.shape) if rank else b""
    space = struct.pack("<BBB5x", 1, rank, 1 if rank else 0) + dims + (dims if rank else b"")
    msgs = _message(0x01, space)
    msgs += _message(0x03, _datatype_message(arr.dtype), flags=1)
    msgs += _message(0x05, struct.pack("<BBBBI", 2, 2, 2, 1, 0), flags=1)       # fill value: version 2, none defined
    msgs += _message(0x08, struct.pack("<BBQQ", 3, 1, data_addr if arr.nbytes else UNDEF, arr.nbytes))
    msgs += _message(0x12, struct.pack("<B3xI", 1, mtime))
    nmsg = 5
    size = _pad8(len(msgs))
    if size < 256:                                                              # room to grow, as h5py leaves it
        msgs += _message(0x00, b"\0" * (256 - len(msgs) - 8))
        nmsg += 1
        size = 256
    return struct.pack("<BBHII4x", 1, 0, nmsg, 1, size) + msgs


def write(path, datasets, mtime=None):
    """Create an HDF5 file with one root group holding ``datasets`` ({name: array-like}) as
    contiguous datasets.  bool arrays are stored as int8."""
    mtime = int(time.time()) if mtime is None else int(mtime)
    items = []
    for name, value in datasets.items():
        a = np.asarray(value)
        if a.dtype == np.bool_:
            a = a.astype(np.int8)
        if a.dtype.kind == "f" and a.dtype.itemsize not in (4, 8):
            a = a.astype(np.float64)
        if a.dtype.kind not in "fiu":
            raise TypeError("dataset %r: dtype %s is outside the supported subset" % (name, a.dtype))
        a = np.array(a, dtype=a.dtype.newbyteorder("<"), order="C")      # (ascontiguousarray would make scalars 1-d)
        nm = str(name).encode("utf-8")
        if not nm or b"/" in nm or b"\0" in nm:
            raise ValueError("dataset name %r is not a plain link name" % (name,))
        items.append((nm, a))
    items.sort(key=lambda t: t[0])                     # symbol-table entries are ordered by name
    n = len(items)
    leaf_k = max(4, (n + 1) // 2)
    int_k = 16
    # local heap: the empty name at offset 0, the link names, one free block
    heap_data = bytearray(8)
    name_off = []
    for nm, _ in items:
        name_off.append(len(heap_data))
        heap_data += nm + b"\0" * (_pad8(len(nm) + 1) - len(nm))
    free_off = len(heap_data)
    heap_data += struct.pack("<QQ", 1, 32) + b"\0" * 16          # free block: next = 1 (end of list), size 32
    # addresses
    root_ohdr = 96
    btree = root_ohdr + 16 + 24
    btree_size = 24 + (2 * int_k + 1) * 8 + 2 * int_k * 8
    heap = btree + btree_size
    heap_data_addr = heap + 32
    snod = _pad8(heap_data_addr + len(heap_data))
    snod_size = 8 + 2 * leaf_k * 40
    p = snod + snod_size
    hdr_addr, data_addr = [], []
    for _, a in items:
        hdr_addr.append(p)
        p += 16 + max(256, _pad8(len(_dataset_header(a, 0, 0)) - 16))
    for _, a in items:
        data_addr.append(p)
        p += _pad8(a.nbytes)
    eof = p
    out = bytearray(eof)
    # superblock (version 0) + root symbol-table entry with the group's B-tree / heap cached
    out[0:8] = SIGNATURE
    struct.pack_into("<BBBBBBBBHHI", out, 8, 0, 0, 0, 0, 0, 8, 8, 0, leaf_k, int_k, 0)
    struct.pack_into("<QQQQ", out, 24, 0, UNDEF, eof, UNDEF)
    struct.pack_into("<QQII", out, 56, 0, root_ohdr, 1, 0)
    struct.pack_into("<QQ", out, 80, btree, heap)
    # root object header: one symbol-table message
    out[root_ohdr:root_ohdr + 40] = struct.pack("<BBHII4x", 1, 0, 1, 1, 24) + _message(0x11, struct.pack("<QQ", btree, heap))
    # B-tree node: one child (the symbol-table node); keys = heap offsets of the empty and the last name
    struct.pack_into("<4sBBHQQ", out, btree, b"TREE", 0, 0, 1 if n else 0, UNDEF, UNDEF)
    struct.pack_into("<QQQ", out, btree + 24, 0, snod, name_off[-1] if n else 0)
    # local heap
    struct.pack_into("<4sB3xQQQ", out, heap, b"HEAP", 0, len(heap_data), free_off, heap_data_addr)
    out[heap_data_addr:heap_data_addr + len(heap_data)] = heap_data
    # symbol-table node
    struct.pack_into("<4sBBH", out, snod, b"SNOD", 1, 0, n)
    for i in range(n):
        struct.pack_into("<QQII", out, snod + 8 + 40 * i, name_off[i], hdr_addr[i], 0, 0)
    # datasets
    for (nm, a), h, d in zip(items, hdr_addr, data_addr):
        hb = _dataset_header(a, d, mtime)
        out[h:h + len(hb)] = hb
        out[d:d + a.nbytes] = a.tobytes()
    with open(path, "wb") as fh:
        fh.write(out)
