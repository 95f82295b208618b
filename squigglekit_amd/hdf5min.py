"""A small read-only HDF5 reader: what it takes to pull raw signals out of fast5 files without h5py.

The reference opens fast5 with h5py (segmenter.py:321-355 single-read files, :358-396 multi-read files;
MotifSeq.py:327-350) and touches very little of HDF5: groups by name, `list(group.keys())`, one 1-D integer
dataset read whole (`Signal[()]`), and a few scalar attributes (`read_id`, `digitisation`, `offset`, `range`,
`sampling_rate`).  This module covers exactly that, from the published HDF5 file format (version 3.0 of the
specification), in pure Python + numpy + zlib:

  superblock 0/1 (and 2/3), object headers 1 and 2, old-style groups (v1 B-tree + local heap + symbol nodes),
  new-style groups with compact link messages, dataspace / datatype (integers, floats, fixed and variable-length
  strings) / layout 3 (compact, contiguous, chunked through a v1 chunk B-tree) / filter pipeline (deflate,
  shuffle, fletcher32) / attribute messages 1-3, global heap for variable-length strings.

Anything else (dense link or attribute storage in fractal heaps, layout version 4, third-party filters such as
ONT's VBZ id 32020, virtual datasets) raises `Hdf5Unsupported` naming the feature; callers report it and move on
to the next file, like the reference does when h5py throws.

    with hdf5min.File(path) as f:
        name = list(f["Raw/Reads"].keys())[0]
        sig = f["Raw/Reads"][name]["Signal"][()]          # numpy array
        rid = f["Raw/Reads"][name].attrs["read_id"]       # bytes, like h5py
"""
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Unsupported(Exception):
    pass


class Hdf5Error(Exception):
    pass


def _pad8(n):
    return (n + 7) & ~7


class File:
    def __init__(self, path):
        with open(path, "rb") as fh:
            self._b = fh.read()
        b = self._b
        sig = b"\x89HDF\r\n\x1a\n"
        base = 0
        while b[base:base + 8] != sig:                      # a user block may precede the superblock
            base = 512 if base == 0 else base * 2
            if base + 8 > len(b):
                raise Hdf5Error("not an HDF5 file: %s" % path)
        self._base = base
        ver = b[base + 8]
        if ver in (0, 1):
            self.O, self.L = b[base + 13], b[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            p += 4 * self.O                                 # base address, free space, end of file, driver info
            self._root_addr = self._uo(p + self.O)          # root symbol-table entry: object header address
        elif ver in (2, 3):
            self.O, self.L = b[base + 9], b[base + 10]
            p = base + 12 + 3 * self.O                      # base address, superblock extension, end of file
            self._root_addr = self._uo(p)
        else:
            raise Hdf5Unsupported("superblock version %d" % ver)
        self._gheaps = {}
        self.root = Group(self, self._root_addr, "/")

    # -- context manager / h5py-shaped access --------------------------------------------------
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def close(self):
        pass

    def __getitem__(self, path):
        return self.root[path]

    def keys(self):
        return self.root.keys()

    @property
    def attrs(self):
        return self.root.attrs

    # -- primitives ---------------------------------------------------------------------------------
    def _u(self, off, size):
        return int.from_bytes(self._b[off:off + size], "little")

    def _uo(self, off):
        return self._u(off, self.O)

    def _ul(self, off):
        return self._u(off, self.L)

    def _addr(self, a):
        return a + self._base

    # -- object headers -----------------------------------------------------------------------------
    def messages(self, addr):
        """[(type, flags, payload bytes)] of the object header at file address `addr`."""
        b = self._b
        a = self._addr(addr)
        out = []
        if b[a:a + 4] == b"OHDR":
            if b[a + 4] != 2:
                raise Hdf5Unsupported("object header version %d" % b[a + 4])
            flags = b[a + 5]
            p = a + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            szsz = 1 << (flags & 3)
            csize = self._u(p, szsz)
            p += szsz
            blocks = [(p, p + csize)]
            corder = 2 if flags & 0x04 else 0
            while blocks:
                p, end = blocks.pop(0)
                while p + 4 + corder <= end:
                    t, sz, fl = b[p], self._u(p + 1, 2), b[p + 3]
                    q = p + 4 + corder
                    if q + sz > end:
                        break
                    data = b[q:q + sz]
                    if t == 0x10:
                        co, cl = self._uo(q), self._ul(q + self.O)
                        ca = self._addr(co)
                        if b[ca:ca + 4] != b"OCHK":
                            raise Hdf5Error("bad object header continuation")
                        blocks.append((ca + 4, ca + cl - 4))
                    elif t != 0:
                        out.append((t, fl, data))
                    p = q + sz
            return out
        ver = b[a]
        if ver != 1:
            raise Hdf5Unsupported("object header version %d" % ver)
        nmsg = self._u(a + 2, 2)
        hsize = self._u(a + 8, 4)
        blocks = [(a + 16, a + 16 + hsize)]
        seen = 0
        while blocks and seen < nmsg:
            p, end = blocks.pop(0)
            while p + 8 <= end and seen < nmsg:
                t, sz, fl = self._u(p, 2), self._u(p + 2, 2), b[p + 4]
                data = b[p + 8:p + 8 + sz]
                seen += 1
                if t == 0x10:
                    co, cl = self._uo(p + 8), self._ul(p + 8 + self.O)
                    blocks.append((self._addr(co), self._addr(co) + cl))
                elif t != 0:
                    out.append((t, fl, data))
                p += 8 + sz
        return out

    def global_heap_object(self, coll_addr, index):
        if coll_addr not in self._gheaps:
            b = self._b
            a = self._addr(coll_addr)
            if b[a:a + 4] != b"GCOL":
                raise Hdf5Error("bad global heap collection")
            size = self._ul(a + 8)
            objs = {}
            p = a + 8 + self.L
            end = a + size
            while p + 8 + self.L <= end:
                idx = self._u(p, 2)
                osz = self._ul(p + 8)
                if idx == 0:
                    break
                objs[idx] = b[p + 8 + self.L:p + 8 + self.L + osz]
                p += 8 + self.L + _pad8(osz)
            self._gheaps[coll_addr] = objs
        return self._gheaps[coll_addr].get(index, b"")


# ------------------------------------------------------------------------------------------------
# datatypes
# ------------------------------------------------------------------------------------------------
class _Type:
    def __init__(self, f, data):
        self.size = int.from_bytes(data[4:8], "little")
        cls = data[0] & 0x0F
        bits = data[1] | (data[2] << 8) | (data[3] << 16)
        self.kind, self.np = None, None
        self.end = 8
        if cls == 0:                                        # fixed point
            order = ">" if bits & 1 else "<"
            self.kind = "num"
            self.np = np.dtype("%s%s%d" % (order, "i" if bits & 8 else "u", self.size))
            self.end = 12
        elif cls == 1:                                      # floating point
            order = ">" if bits & 1 else "<"
            if self.size not in (2, 4, 8):
                raise Hdf5Unsupported("%d-byte float" % self.size)
            self.kind = "num"
            self.np = np.dtype("%sf%d" % (order, self.size))
            self.end = 20
        elif cls == 3:                                      # fixed-length string
            self.kind = "str"
            self.np = np.dtype("S%d" % self.size)
        elif cls == 9:                                      # variable length
            if (bits & 0x0F) != 1:
                raise Hdf5Unsupported("variable-length sequence datatype")
            self.kind = "vstr"
        elif cls == 8:                                      # enumeration over an integer base (e.g. booleans)
            base = _Type(f, data[8:])
            self.kind, self.np = base.kind, base.np
        else:
            raise Hdf5Unsupported("datatype class %d" % cls)


def _dataspace(f, data):
    ver, rank, flags = data[0], data[1], data[2]
    if ver == 1:
        p = 8
    elif ver == 2:
        if data[3] == 2:                                    # null dataspace
            return None
        p = 4
    else:
        raise Hdf5Unsupported("dataspace version %d" % ver)
    return tuple(int.from_bytes(data[p + i * f.L:p + (i + 1) * f.L], "little") for i in range(rank))


def _decode(f, typ, shape, raw):
    count = 1
    for d in (shape or ()):
        count *= d
    if typ.kind == "num":
        a = np.frombuffer(raw, dtype=typ.np, count=count).reshape(shape or ())
        return a[()] if not shape else a.copy()
    if typ.kind == "str":
        a = np.frombuffer(raw, dtype=typ.np, count=count)
        vals = [bytes(x).split(b"\0", 1)[0] for x in a]
        return vals[0] if not shape else np.array(vals, dtype=object).reshape(shape)
    if typ.kind == "vstr":
        step = 4 + f.O + 4
        vals = []
        for i in range(count):
            rec = raw[i * step:(i + 1) * step]
            coll = int.from_bytes(rec[4:4 + f.O], "little")
            idx = int.from_bytes(rec[4 + f.O:8 + f.O], "little")
            n = int.from_bytes(rec[:4], "little")
            vals.append(f.global_heap_object(coll, idx)[:n] if coll not in (0, UNDEF) else b"")
        return vals[0] if not shape else np.array(vals, dtype=object).reshape(shape)
    raise Hdf5Unsupported("datatype")


def _attribute(f, data):
    ver = data[0]
    nsz, tsz, ssz = (int.from_bytes(data[2:4], "little"), int.from_bytes(data[4:6], "little"),
                     int.from_bytes(data[6:8], "little"))
    if ver == 1:
        p = 8
        name = data[p:p + nsz].split(b"\0", 1)[0].decode("utf-8", "replace")
        p += _pad8(nsz)
        tdata = data[p:p + tsz]
        p += _pad8(tsz)
        sdata = data[p:p + ssz]
        p += _pad8(ssz)
    elif ver in (2, 3):
        if data[1] & 3:
            raise Hdf5Unsupported("shared attribute datatype / dataspace")
        p = 8 + (1 if ver == 3 else 0)
        name = data[p:p + nsz].split(b"\0", 1)[0].decode("utf-8", "replace")
        p += nsz
        tdata = data[p:p + tsz]
        p += tsz
        sdata = data[p:p + ssz]
        p += ssz
    else:
        raise Hdf5Unsupported("attribute message version %d" % ver)
    shape = _dataspace(f, sdata)
    if shape is None:
        return name, None
    return name, _decode(f, _Type(f, tdata), shape, data[p:])


# ------------------------------------------------------------------------------------------------
# objects
# ------------------------------------------------------------------------------------------------
class _Object:
    def __init__(self, f, addr, name):
        self._f, self._addr_, self.name = f, addr, name
        self._msgs = f.messages(addr)
        self._attrs = None

    @property
    def attrs(self):
        if self._attrs is None:
            out = {}
            for t, _, data in self._msgs:
                if t == 0x0C:
                    k, v = _attribute(self._f, data)
                    out[k] = v
                elif t == 0x15:                             # attribute info: dense storage?
                    p = 2 + (2 if data[1] & 1 else 0)
                    if int.from_bytes(data[p:p + self._f.O], "little") != UNDEF:
                        raise Hdf5Unsupported("attributes in dense (fractal heap) storage")
            self._attrs = out
        return self._attrs


def _open(f, addr, name):
    types = {t for t, _, _ in f.messages(addr)}
    return Dataset(f, addr, name) if 0x08 in types else Group(f, addr, name)


class Group(_Object):
    def _links(self):
        f, b = self._f, self._f._b
        links = {}
        for t, _, data in self._msgs:
            if t == 0x11:                                   # symbol table: v1 B-tree + local heap
                btree, heap = int.from_bytes(data[:f.O], "little"), int.from_bytes(data[f.O:2 * f.O], "little")
                h = f._addr(heap)
                if b[h:h + 4] != b"HEAP":
                    raise Hdf5Error("bad local heap")
                hdata = f._addr(f._uo(h + 8 + 2 * f.L))
                stack = [btree]
                while stack:
                    n = f._addr(stack.pop())
                    if b[n:n + 4] == b"TREE":
                        level, nent = b[n + 5], f._u(n + 6, 2)
                        p = n + 8 + 2 * f.O
                        kids = []
                        for i in range(nent):
                            p += f.L                        # key i
                            kids.append(f._uo(p))
                            p += f.O
                        stack.extend(reversed(kids))
                    elif b[n:n + 4] == b"SNOD":
                        nsym = f._u(n + 6, 2)
                        p = n + 8
                        for i in range(nsym):
                            noff, oaddr = f._uo(p), f._uo(p + f.O)
                            s = hdata + noff
                            nm = b[s:b.index(b"\0", s)].decode("utf-8", "replace")
                            links[nm] = oaddr
                            p += 2 * f.O + 24
                    else:
                        raise Hdf5Error("bad group B-tree node")
            elif t == 0x06:                                 # link message (compact new-style group)
                fl = data[1]
                p = 2
                ltype = 0
                if fl & 0x08:
                    ltype = data[p]
                    p += 1
                if fl & 0x04:
                    p += 8
                if fl & 0x10:
                    p += 1
                lsz = 1 << (fl & 3)
                nlen = int.from_bytes(data[p:p + lsz], "little")
                p += lsz
                nm = data[p:p + nlen].decode("utf-8", "replace")
                p += nlen
                if ltype == 0:
                    links[nm] = int.from_bytes(data[p:p + f.O], "little")
            elif t == 0x02:                                 # link info: dense storage?
                p = 2 + (8 if data[1] & 1 else 0)
                if int.from_bytes(data[p:p + f.O], "little") != UNDEF:
                    raise Hdf5Unsupported("group links in dense (fractal heap) storage")
        return links

    def keys(self):
        if not hasattr(self, "_lk"):
            self._lk = self._links()
        return list(self._lk.keys())

    def __iter__(self):
        return iter(self.keys())

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [x for x in path.split("/") if x]:
            if not isinstance(node, Group):
                raise KeyError(path)
            node.keys()
            if part not in node._lk:
                raise KeyError("%s (no '%s' in %s)" % (path, part, node.name))
            node = _open(self._f, node._lk[part], node.name.rstrip("/") + "/" + part)
        return node


class Dataset(_Object):
    def _describe(self):
        f = self._f
        self.shape = self.dtype = None
        typ = layout = None
        filters = []
        for t, _, data in self._msgs:
            if t == 0x01:
                self.shape = _dataspace(f, data)
            elif t == 0x03:
                typ = _Type(f, data)
            elif t == 0x08:
                layout = data
            elif t == 0x0B:
                ver, nf = data[0], data[1]
                p = 8 if ver == 1 else 2
                for _ in range(nf):
                    fid = int.from_bytes(data[p:p + 2], "little")
                    p += 2
                    nlen = 0
                    if ver == 1 or fid >= 256:
                        nlen = int.from_bytes(data[p:p + 2], "little")
                        p += 2
                    p += 2                                  # flags
                    ncd = int.from_bytes(data[p:p + 2], "little")
                    p += 2
                    p += _pad8(nlen) if ver == 1 else nlen
                    cd = [int.from_bytes(data[p + 4 * i:p + 4 * i + 4], "little") for i in range(ncd)]
                    p += 4 * ncd
                    if ver == 1 and ncd % 2:
                        p += 4
                    filters.append((fid, cd))
        if typ is None or layout is None or self.shape is None:
            raise Hdf5Error("incomplete dataset header: %s" % self.name)
        return typ, layout, filters

    def _unfilter(self, raw, filters, mask, itemsize):
        for k in range(len(filters) - 1, -1, -1):
            if mask & (1 << k):
                continue
            fid, cd = filters[k]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:                                  # shuffle
                n = len(raw) // itemsize
                raw = np.frombuffer(raw[:n * itemsize], dtype=np.uint8).reshape(itemsize, n).T.tobytes() + raw[n * itemsize:]
            elif fid == 3:                                  # fletcher32: checksum at the end
                raw = raw[:-4]
            else:
                raise Hdf5Unsupported("filter id %d%s" % (fid, " (ONT VBZ compression)" if fid == 32020 else ""))
        return raw

    def __getitem__(self, key):
        if key != () and key is not Ellipsis and key != slice(None):
            return self[()][key]
        f, b = self._f, self._f._b
        typ, layout, filters = self._describe()
        shape = self.shape
        count = 1
        for d in shape:
            count *= d
        if typ.kind != "num":
            raise Hdf5Unsupported("non-numeric dataset")
        item = typ.np.itemsize
        ver, cls = layout[0], layout[1]
        if ver != 3:
            raise Hdf5Unsupported("data layout message version %d" % ver)
        if cls == 0:
            size = int.from_bytes(layout[2:4], "little")
            raw = layout[4:4 + size]
        elif cls == 1:
            addr = int.from_bytes(layout[2:2 + f.O], "little")
            raw = b"\0" * (count * item) if addr == UNDEF else b[f._addr(addr):f._addr(addr) + count * item]
        elif cls == 2:
            nd = layout[2]
            btree = int.from_bytes(layout[3:3 + f.O], "little")
            cdims = [int.from_bytes(layout[3 + f.O + 4 * i:7 + f.O + 4 * i], "little") for i in range(nd)]
            cshape = tuple(cdims[:-1])
            out = np.zeros(shape, dtype=typ.np)
            if btree != UNDEF:
                stack = [btree]
                while stack:
                    n = f._addr(stack.pop())
                    if b[n:n + 4] != b"TREE" or b[n + 4] != 1:
                        raise Hdf5Error("bad chunk B-tree node")
                    level, nent = b[n + 5], f._u(n + 6, 2)
                    p = n + 8 + 2 * f.O
                    for _ in range(nent):
                        csize, mask = f._u(p, 4), f._u(p + 4, 4)
                        offs = [f._u(p + 8 + 8 * i, 8) for i in range(nd - 1)]
                        child = f._uo(p + 8 + 8 * nd)
                        p += 8 + 8 * nd + f.O
                        if level > 0:
                            stack.append(child)
                            continue
                        ca = f._addr(child)
                        raw = self._unfilter(b[ca:ca + csize], filters, mask, item)
                        if len(cshape) == 1:
                            # (writers that size the chunk beyond the dataset store only the part that exists:
                            # example/test.fast5 has a 201 536-sample chunk holding 36 978 samples)
                            have = min(len(raw) // item, cshape[0], shape[0] - offs[0])
                            out[offs[0]:offs[0] + have] = np.frombuffer(raw, dtype=typ.np, count=have)
                            continue
                        chunk = np.frombuffer(raw, dtype=typ.np, count=int(np.prod(cshape))).reshape(cshape)
                        sel_o = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cshape, shape))
                        sel_c = tuple(slice(0, s.stop - s.start) for s in sel_o)
                        out[sel_o] = chunk[sel_c]
            return out
        else:
            raise Hdf5Unsupported("data layout class %d" % cls)
        return np.frombuffer(raw, dtype=typ.np, count=count).reshape(shape).copy()

    @property
    def value(self):
        return self[()]
