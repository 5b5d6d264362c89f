"""A small HDF5 reader / writer for the wire formats either side of the hot path.

The reference stores models, k-point lists and eigenvalue tables in HDF5 through ``h5py`` +
``fsc.hdf5_io`` (`src/tbmodels/_tb_model.py:1000-1058`, `src/tbmodels/_cli.py:227-262`; the
``bands_inspect`` containers ``kpoints_explicit`` / ``eigenvals_data``).  The target image has no
``h5py``, so this module implements the subset of the HDF5 file format those writers emit -- and nothing
else -- on NumPy and ``struct``:

* superblock version 0 / 1, object headers version 1 (with continuation blocks),
* "old style" groups: symbol-table message -> v1 B-tree -> symbol-table nodes -> local heap,
* datasets with contiguous or compact layout (no chunking, no filters), scalar or simple dataspaces,
* datatypes: fixed-point, IEEE float, compound (used for complex: ``{r, i}``), enum over int8 (h5py's
  bool), fixed-length strings and variable-length strings (global heap).

``read(path)`` returns the file as a nested ``dict`` (groups) of NumPy arrays / scalars / ``str``.
``write(path, tree)`` stores such a tree so that libhdf5 / h5py read it back (checked in
``tests/test_hdf5_lite.py`` against the reference's own sample files, and against real h5py where one is
installed).  Anything outside the subset raises :class:`HDF5FormatError` naming the feature.

Format reference: "HDF5 File Format Specification Version 2.0" (the public spec; section numbers in the
comments below).
"""

import struct

import numpy as np

__all__ = ("read", "write", "HDF5FormatError")

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class HDF5FormatError(ValueError):
    """The file uses an HDF5 feature outside the subset this module implements (or is not HDF5)."""


# =================================================================================================
# reading
# =================================================================================================
class _Reader:
    def __init__(self, buf):
        self.b = buf
        if len(buf) < 64 or buf[:8] != SIGNATURE:
            raise HDF5FormatError("not an HDF5 file (signature missing at offset 0)")
        version = buf[8]
        if version not in (0, 1):
            raise HDF5FormatError("superblock version %d is not supported (only 0 and 1)" % version)
        self.size_off, self.size_len = buf[13], buf[14]
        if (self.size_off, self.size_len) != (8, 8):
            raise HDF5FormatError("only 8-byte offsets / lengths are supported")
        pos = 24 + (4 if version == 1 else 0)  # group K values, flags [, indexed storage K]
        self.base, _free, self.eof, _driver = struct.unpack_from("<4Q", buf, pos)
        pos += 32
        # root group symbol table entry (spec III.C)
        _name_off, self.root_header = struct.unpack_from("<QQ", buf, pos)

    # ---- low level ------------------------------------------------------------------------------
    def u(self, pos, n):
        return int.from_bytes(self.b[pos : pos + n], "little")

    def messages(self, addr):
        """All (type, flags, payload position, size) of a version-1 object header (spec IV.A.1.a)."""
        b = self.b
        if b[addr] != 1:
            if b[addr : addr + 4] == b"OHDR":
                raise HDF5FormatError("version-2 object headers (libver='latest' files) are not supported")
            raise HDF5FormatError("unknown object header version %d" % b[addr])
        n_msg, _refs, hdr_size = struct.unpack_from("<HIi", b, addr + 2)
        blocks = [(addr + 16, hdr_size)]
        out = []
        while blocks and len(out) < n_msg:
            pos, size = blocks.pop(0)
            end = pos + size
            while pos + 8 <= end and len(out) < n_msg:
                mtype, msize, flags = struct.unpack_from("<HHB", b, pos)
                body = pos + 8
                if mtype == 0x0010:  # continuation
                    c_off, c_len = struct.unpack_from("<QQ", b, body)
                    blocks.append((c_off + self.base, c_len))
                out.append((mtype, flags, body, msize))
                pos = body + msize
        return out

    # ---- datatypes (spec IV.A.2.d) ----------------------------------------------------------------
    def datatype(self, pos):
        """-> (numpy dtype or a tag tuple, bytes consumed)."""
        b = self.b
        cls, version = b[pos] & 0x0F, b[pos] >> 4
        bits = b[pos + 1] | (b[pos + 2] << 8) | (b[pos + 3] << 16)
        size = self.u(pos + 4, 4)
        p = pos + 8
        if cls == 0:  # fixed point
            order = ">" if bits & 1 else "<"
            kind = "i" if bits & 8 else "u"
            return np.dtype("%s%s%d" % (order, kind, size)), p + 4 - pos
        if cls == 1:  # floating point
            order = ">" if bits & 1 else "<"
            if size not in (2, 4, 8):
                raise HDF5FormatError("%d-byte floating point type is not supported" % size)
            return np.dtype("%sf%d" % (order, size)), p + 12 - pos
        if cls == 3:  # fixed-length string
            return np.dtype("S%d" % size), p - pos
        if cls == 6:  # compound
            n_members = bits & 0xFFFF
            names, offsets, types = [], [], []
            for _ in range(n_members):
                end = b.index(b"\0", p)
                name = b[p:end].decode()
                if version < 3:
                    p += (end - p + 8) // 8 * 8
                else:
                    p = end + 1
                if version == 1:
                    offset = self.u(p, 4)
                    if b[p + 4] != 0:
                        raise HDF5FormatError("array members of compound types are not supported")
                    p += 4 + 1 + 3 + 4 + 4 + 16
                elif version == 2:
                    offset = self.u(p, 4)
                    p += 4
                else:
                    nbytes = 1 if size < 256 else 2 if size < 65536 else 4
                    offset = self.u(p, nbytes)
                    p += nbytes
                mtype, used = self.datatype(p)
                if not isinstance(mtype, np.dtype):
                    raise HDF5FormatError("compound member '%s' has an unsupported type" % name)
                p += used
                names.append(name)
                offsets.append(offset)
                types.append(mtype)
            return np.dtype({"names": names, "formats": types, "offsets": offsets, "itemsize": size}), p - pos
        if cls == 8:  # enumeration
            n_members = bits & 0xFFFF
            base, used = self.datatype(p)
            p += used
            names = []
            for _ in range(n_members):
                end = b.index(b"\0", p)
                names.append(b[p:end].decode())
                p = p + (end - p + 8) // 8 * 8 if version < 3 else end + 1
            values = np.frombuffer(b, dtype=base, count=n_members, offset=p)
            p += n_members * base.itemsize
            mapping = dict(zip(names, values.tolist()))
            if sorted(mapping) == ["FALSE", "TRUE"] and mapping["FALSE"] == 0 and mapping["TRUE"] == 1:
                return ("bool", base), p - pos
            return base, p - pos  # other enums: hand out the integer codes
        if cls == 9:  # variable length
            if bits & 0x0F != 1:
                raise HDF5FormatError("variable-length sequences are not supported (only strings)")
            _base, used = self.datatype(p)
            return ("vlen_str", size), p + used - pos
        names = {2: "time", 4: "bit field", 5: "opaque", 7: "reference", 10: "array"}
        raise HDF5FormatError("datatype class %s is not supported" % names.get(cls, cls))

    def global_heap_object(self, addr, index):
        """(spec III.E)"""
        b = self.b
        if b[addr : addr + 4] != b"GCOL":
            raise HDF5FormatError("global heap collection signature missing")
        size = self.u(addr + 8, 8)
        pos, end = addr + 16, addr + size
        while pos + 16 <= end:
            idx, _refs, _res, osize = struct.unpack_from("<HHIQ", b, pos)
            if idx == index:
                return bytes(b[pos + 16 : pos + 16 + osize])
            if idx == 0:
                break
            pos += 16 + (osize + 7) // 8 * 8
        raise HDF5FormatError("global heap object %d not found" % index)

    # ---- objects ----------------------------------------------------------------------------------
    def node(self, addr):
        msgs = self.messages(addr)
        by_type = {}
        for mtype, _flags, body, size in msgs:
            by_type.setdefault(mtype, (body, size))
        if 0x0011 in by_type:
            btree, heap = struct.unpack_from("<QQ", self.b, by_type[0x0011][0])
            return self.group(btree + self.base, heap + self.base)
        if 0x0002 in by_type or 0x0006 in by_type:
            raise HDF5FormatError("new-style groups (link messages; libver='latest') are not supported")
        if 0x0001 in by_type and 0x0003 in by_type and 0x0008 in by_type:
            if 0x000B in by_type:
                raise HDF5FormatError("filtered (compressed) datasets are not supported")
            return self.dataset(by_type[0x0001][0], by_type[0x0003][0], by_type[0x0008][0])
        if 0x0003 in by_type:
            raise HDF5FormatError("committed datatypes are not supported")
        return {}  # a group without a symbol table: empty

    def group(self, btree, heap):
        b = self.b
        if b[heap : heap + 4] != b"HEAP":
            raise HDF5FormatError("local heap signature missing")
        heap_data = self.u(heap + 24, 8) + self.base
        out = {}

        def walk(addr):
            if b[addr : addr + 4] == b"SNOD":
                n = self.u(addr + 6, 2)
                for i in range(n):
                    e = addr + 8 + 40 * i
                    name_off, header = struct.unpack_from("<QQ", b, e)
                    start = heap_data + name_off
                    name = b[start : b.index(b"\0", start)].decode()
                    out[name] = self.node(header + self.base)
                return
            if b[addr : addr + 4] != b"TREE" or b[addr + 4] != 0:
                raise HDF5FormatError("group B-tree node signature missing")
            n = self.u(addr + 6, 2)
            for i in range(n):
                child = self.u(addr + 24 + 8 + 16 * i, 8)  # key_i (8), child_i (8), ...
                walk(child + self.base)

        if btree != UNDEF + self.base and btree != UNDEF:
            walk(btree)
        return out

    def dataset(self, p_space, p_type, p_layout):
        b = self.b
        # dataspace (spec IV.A.2.b)
        version, rank, flags = b[p_space], b[p_space + 1], b[p_space + 2]
        if version == 1:
            dims_at = p_space + 8
        elif version == 2:
            dims_at = p_space + 4
            if b[p_space + 3] == 2:
                return None  # null dataspace
        else:
            raise HDF5FormatError("dataspace message version %d is not supported" % version)
        shape = tuple(self.u(dims_at + 8 * i, 8) for i in range(rank))
        count = int(np.prod(shape, dtype=np.int64)) if rank else 1
        dtype, _ = self.datatype(p_type)
        # layout (spec IV.A.2.i)
        lversion = b[p_layout]
        if lversion == 3:
            lclass = b[p_layout + 1]
            if lclass == 0:
                size = self.u(p_layout + 2, 2)
                data_at = p_layout + 4
            elif lclass == 1:
                addr, size = struct.unpack_from("<QQ", b, p_layout + 2)
                data_at = None if addr == UNDEF else addr + self.base
            else:
                raise HDF5FormatError("chunked datasets are not supported")
        elif lversion in (1, 2):
            ndim, lclass = b[p_layout + 1], b[p_layout + 2]
            if lclass == 2:
                raise HDF5FormatError("chunked datasets are not supported")
            p = p_layout + 8
            if lclass == 1:
                addr = self.u(p, 8)
                data_at = None if addr == UNDEF else addr + self.base
                p += 8
            p += 4 * ndim
            if lclass == 0:
                data_at = p + 4
        else:
            raise HDF5FormatError("data layout message version %d is not supported" % lversion)

        if isinstance(dtype, tuple) and dtype[0] == "vlen_str":
            strings = []
            for i in range(count):
                if data_at is None:
                    strings.append("")
                    continue
                e = data_at + 16 * i
                length, heap_addr, index = struct.unpack_from("<IQI", b, e)
                raw = self.global_heap_object(heap_addr + self.base, index)[:length] if length else b""
                strings.append(raw.decode("utf-8"))
            if not rank:
                return strings[0]
            return np.array(strings, dtype=object).reshape(shape)
        is_bool = isinstance(dtype, tuple) and dtype[0] == "bool"
        np_dtype = dtype[1] if is_bool else dtype
        if data_at is None:  # never written: fill value (zero)
            arr = np.zeros(count, dtype=np_dtype)
        else:
            arr = np.frombuffer(b, dtype=np_dtype, count=count, offset=data_at)
        if np_dtype.names == ("r", "i") and np_dtype[0] == np_dtype[1] and np_dtype[0].kind == "f":
            arr = arr["r"] + 1j * arr["i"]  # h5py's complex convention
        elif np_dtype.kind == "S":
            arr = np.array([s.decode("utf-8") for s in arr.tolist()], dtype=object)
        else:
            arr = arr.astype(np_dtype.newbyteorder("="), copy=True)
        if is_bool:
            arr = arr.astype(bool)
        arr = arr.reshape(shape)
        return arr[()] if not rank else arr


def read(path):
    """Read a whole HDF5 file into a nested dict (groups) of arrays / NumPy scalars / ``str``."""
    with open(path, "rb") as handle:
        buf = handle.read()
    reader = _Reader(buf)
    return reader.node(reader.root_header + reader.base)


# =================================================================================================
# writing
# =================================================================================================
_LEAF_K = 256      # symbol-table node holds up to 2 K entries
_INTERNAL_K = 64   # B-tree node holds up to 2 K children


def _pad8(raw):
    return raw + b"\0" * (-len(raw) % 8)


def _dt_fixed(size, signed=True):
    return struct.pack("<BBBBI", 0x10 | 0, 0x08 if signed else 0x00, 0, 0, size) + struct.pack("<HH", 0, size * 8)


def _dt_float(size):
    # IEEE little endian: (bit offset, precision, exponent location, exponent size, mantissa location, size, bias)
    if size == 8:
        props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        sign_loc = 63
    else:
        props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        sign_loc = 31
    # bits: byte order 0, padding 0, mantissa normalisation = 2 (implied msb) at bits 4-5, sign location at 8-15
    return struct.pack("<BBBBI", 0x10 | 1, 0x20, sign_loc, 0, size) + props


def _dt_complex(fsize):
    members = b""
    for name, offset in ((b"r", 0), (b"i", fsize)):
        members += _pad8(name + b"\0") + struct.pack("<IB3xII4I", offset, 0, 0, 0, 0, 0, 0, 0) + _dt_float(fsize)
    return struct.pack("<BBBBI", 0x10 | 6, 2, 0, 0, 2 * fsize) + members


def _dt_bool():
    names = _pad8(b"FALSE\0") + _pad8(b"TRUE\0")
    return struct.pack("<BBBBI", 0x10 | 8, 2, 0, 0, 1) + _dt_fixed(1) + names + bytes([0, 1])


def _dt_vlen_str():
    # class 9, type = string (1), padding null-terminate (0), character set: UTF-8 (1 at bits 8-11)
    base = struct.pack("<BBBBI", 0x10 | 3, 0x10, 0, 0, 1)  # 1-byte UTF-8 string base type
    return struct.pack("<BBBBI", 0x10 | 9, 0x01, 0x01, 0, 16) + base


class _Writer:
    def __init__(self, first_free):
        self.chunks = []          # (address, bytes)
        self.pos = first_free     # next unallocated file offset
        self.gheap_addr = None    # the global heap collection holding every variable-length string
        self.n_strings = 0

    def alloc(self, raw):
        self.pos += -self.pos % 8
        addr = self.pos
        self.chunks.append((addr, raw))
        self.pos += len(raw)
        return addr

    def reserve(self, size):
        self.pos += -self.pos % 8
        addr = self.pos
        self.pos += size
        return addr

    def put(self, addr, raw):
        self.chunks.append((addr, raw))

    # ---- object headers ---------------------------------------------------------------------------
    @staticmethod
    def message(mtype, body, flags=0):
        body = _pad8(body)
        return struct.pack("<HHB3x", mtype, len(body), flags) + body

    def object_header(self, messages):
        body = b"".join(messages)
        head = struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body))
        return self.alloc(head + body)

    def dataset(self, value):
        dt, arr, raw = self.encode(value)
        shape = arr.shape if arr is not None else ()
        space = struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", d) for d in shape)
        if raw:
            data_addr = self.alloc(raw)
        else:
            data_addr = UNDEF
        layout = struct.pack("<BBQQ", 3, 1, data_addr, len(raw))
        fill = struct.pack("<BBBB", 2, 2, 0, 0)  # version 2, late allocation, write at allocation, undefined value
        msgs = [
            self.message(0x0001, space),
            self.message(0x0003, dt, flags=1),  # constant message
            self.message(0x0005, fill),
            self.message(0x0008, layout),
        ]
        return self.object_header(msgs)

    def encode(self, value):
        """-> (datatype message body, array for the shape, raw little-endian data)."""
        if isinstance(value, (str, bytes)):
            payload = value.encode("utf-8") if isinstance(value, str) else value
            self.n_strings += 1  # objects of the collection are numbered in write order (see `write`)
            return _dt_vlen_str(), None, struct.pack("<IQI", len(payload), self.gheap_addr, self.n_strings)
        arr = np.asarray(value)
        if arr.dtype == bool:
            return _dt_bool(), arr, arr.astype("<i1").tobytes()
        if arr.dtype.kind in "iu":
            return _dt_fixed(arr.dtype.itemsize, arr.dtype.kind == "i"), arr, arr.astype(arr.dtype.newbyteorder("<")).tobytes()
        if arr.dtype.kind == "f":
            if arr.dtype.itemsize not in (4, 8):
                arr = arr.astype(np.float64)
            return _dt_float(arr.dtype.itemsize), arr, arr.astype(arr.dtype.newbyteorder("<")).tobytes()
        if arr.dtype.kind == "c":
            if arr.dtype.itemsize not in (8, 16):
                arr = arr.astype(np.complex128)
            return _dt_complex(arr.dtype.itemsize // 2), arr, arr.astype(arr.dtype.newbyteorder("<")).tobytes()
        raise TypeError("cannot store values of dtype %s in HDF5 (hdf5_lite)" % arr.dtype)

    # ---- groups -----------------------------------------------------------------------------------
    def group(self, tree):
        """Writes the children, then heap + symbol-table nodes + B-tree + header; returns
        (header address, B-tree address, heap address)."""
        names = _sorted_names(tree)
        headers = []
        for name in names:
            child = tree[name]
            if isinstance(child, dict):
                headers.append(self.group(child))
            else:
                headers.append((self.dataset(child), None, None))

        # local heap data segment: offset 0 holds the empty string
        data = bytearray(8)
        name_off = []
        for name in names:
            name_off.append(len(data))
            data += _pad8(name.encode("utf-8") + b"\0")
        free_at = len(data)
        data += struct.pack("<QQ", 1, 16)  # one free block at the tail: (next = H5HL_FREE_NULL, size)
        data_addr = self.alloc(bytes(data))
        heap_addr = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(data), free_at, data_addr))

        # symbol-table nodes (leaves) of up to 2 * _LEAF_K entries
        def entry(i):
            header, btree, heap = headers[i]
            if btree is None:
                return struct.pack("<QQII16x", name_off[i], header, 0, 0)
            return struct.pack("<QQIIQQ", name_off[i], header, 1, 0, btree, heap)

        per_leaf = 2 * _LEAF_K
        leaves = []  # (address, heap offset of the last name in the node)
        for start in range(0, len(names), per_leaf):
            idx = range(start, min(start + per_leaf, len(names)))
            raw = b"SNOD" + struct.pack("<BBH", 1, 0, len(idx)) + b"".join(entry(i) for i in idx)
            raw += b"\0" * (8 + per_leaf * 40 - len(raw))
            leaves.append((self.alloc(raw), name_off[idx[-1]]))

        # B-tree over the leaves (spec III.A.1); keys are heap offsets of names, key_0 = 0 (the empty string)
        def tree_nodes(children, level):
            per_node = 2 * _INTERNAL_K
            nodes = []
            for start in range(0, len(children), per_node):
                part = children[start : start + per_node]
                raw = b"TREE" + struct.pack("<BBHQQ", 0, level, len(part), UNDEF, UNDEF)
                raw += struct.pack("<Q", 0 if start == 0 else children[start - 1][1])
                for addr, last in part:
                    raw += struct.pack("<QQ", addr, last)
                raw += b"\0" * (24 + 8 + per_node * 16 - len(raw))
                nodes.append([self.reserve(len(raw)), part[-1][1], raw])
            # sibling links
            for i, nd in enumerate(nodes):
                left = nodes[i - 1][0] if i > 0 else UNDEF
                right = nodes[i + 1][0] if i + 1 < len(nodes) else UNDEF
                raw = nd[2][:8] + struct.pack("<QQ", left, right) + nd[2][24:]
                self.put(nd[0], raw)
            out = [(nd[0], nd[1]) for nd in nodes]
            return out if len(out) == 1 else tree_nodes(out, level + 1)

        if leaves:
            btree_addr = tree_nodes(leaves, 0)[0][0]
        else:
            raw = b"TREE" + struct.pack("<BBHQQ", 0, 0, 0, UNDEF, UNDEF) + b"\0" * (8 + 2 * _INTERNAL_K * 16)
            btree_addr = self.alloc(raw)
        header = self.object_header([self.message(0x0011, struct.pack("<QQ", btree_addr, heap_addr))])
        return header, btree_addr, heap_addr


def _sorted_names(tree):
    return sorted(tree, key=lambda name: name.encode("utf-8"))  # the order libhdf5 keeps symbol tables in


def write(path, tree):
    """Store a nested dict of arrays / scalars / strings as an HDF5 file (old-style groups, contiguous data)."""
    if not isinstance(tree, dict):
        raise TypeError("hdf5_lite.write expects a dict (the root group)")
    superblock_size = 8 + 8 + 8 + 32 + 40  # signature, versions / sizes, K values + flags, addresses, root entry
    w = _Writer(superblock_size)

    # Variable-length strings live in ONE global heap collection (spec III.E), placed first so that its
    # address is known while the datasets that point into it are encoded.  Objects are numbered in the
    # order `_Writer.group` visits datasets: name-sorted, depth first.
    strings = []

    def collect(t):
        for name in _sorted_names(t):
            v = t[name]
            if isinstance(v, dict):
                collect(v)
            elif isinstance(v, (str, bytes)):
                strings.append(v.encode("utf-8") if isinstance(v, str) else v)

    collect(tree)
    if strings:
        body = b""
        for i, payload in enumerate(strings):
            body += struct.pack("<HHIQ", i + 1, 1, 0, len(payload)) + _pad8(payload)
        size = max(4096, 16 + len(body) + 16)
        body += struct.pack("<HHIQ", 0, 0, 0, size - 16 - len(body))  # object 0: the free space
        raw = b"GCOL" + struct.pack("<B3xQ", 1, size) + body
        w.gheap_addr = w.alloc(raw + b"\0" * (size - len(raw)))

    root_header, root_btree, root_heap = w.group(tree)
    eof = w.pos + (-w.pos % 8)

    sb = SIGNATURE + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0)
    sb += struct.pack("<HHI", _LEAF_K, _INTERNAL_K, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQIIQQ", 0, root_header, 1, 0, root_btree, root_heap)
    assert len(sb) == superblock_size
    out = bytearray(eof)
    out[: len(sb)] = sb
    for addr, raw in w.chunks:
        out[addr : addr + len(raw)] = raw
    with open(path, "wb") as handle:
        handle.write(bytes(out))
