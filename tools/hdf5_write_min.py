#!/usr/bin/env python3
"""A minimal HDF5 WRITER, for test fixtures only: what it takes to lay out a multi-read fast5 the way classic
(libver "earliest") HDF5 does -- superblock 0, version-1 object headers, old-style groups (symbol-table message -> v1
B-tree -> one symbol node + local heap), 1-D integer datasets stored chunked (one chunk, v1 chunk B-tree) behind the
deflate filter, version-1 attribute messages (float64 scalars, fixed-length strings).  Written from the published file
format specification (version 3.0); h5py is not installable here, and the reference ships no multi-read file.

    tree = {"read_abc": {"Raw": {"@read_id": b"abc", "Signal": np.int16 array}, "channel_id": {"@offset": 12.0, ...}}}
    write_hdf5(path, tree)

Keys starting with "@" are attributes of the enclosing group; dict values are groups; numpy arrays are datasets.
tools/gen_golden_fast5.py builds tests/golden/multi_two_reads.fast5 with it; squigglekit_amd/hdf5min.py (validated
against the reference's own example/test.fast5) reads it back."""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


def _pad8(b):
    return b + b"\0" * ((-len(b)) % 8)


class _Out:
    def __init__(self):
        self.b = bytearray(96)                               # the superblock is written last

    def put(self, data):
        while len(self.b) % 8:
            self.b.append(0)
        addr = len(self.b)
        self.b += data
        return addr


def _msg(mtype, data):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), 0) + data


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind in "iu":
        bits = 0x08 if dt.kind == "i" else 0x00
        return struct.pack("<BBBBI", 0x10, bits, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
    if dt == np.float64:
        return (struct.pack("<BBBBI", 0x11, 0x20, 0x3F, 0, 8) +
                struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023))
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0, 0, 0, dt.itemsize)
    raise ValueError(dt)


def _space_msg(shape):
    out = struct.pack("<BBB5x", 1, len(shape), 0)
    for d in shape:
        out += struct.pack("<Q", d)
    return out


def _attr(name, value):
    if isinstance(value, bytes):
        arr = np.array(value, dtype="S%d" % max(1, len(value)))
    else:
        arr = np.array(float(value), dtype=np.float64)
    nm = name.encode() + b"\0"
    t, s = _dtype_msg(arr.dtype), _space_msg(())
    return _msg(0x0C, struct.pack("<BxHHH", 1, len(nm), len(t), len(s)) + _pad8(nm) + _pad8(t) + _pad8(s) + arr.tobytes())


def _header(out, msgs):
    body = b"".join(msgs)
    return out.put(struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body)


def _dataset(out, arr, attrs):
    arr = np.ascontiguousarray(arr)
    assert arr.ndim == 1
    comp = zlib.compress(arr.tobytes(), 4)
    chunk_addr = out.put(comp)
    n, item = arr.shape[0], arr.dtype.itemsize
    # chunk B-tree: one leaf entry; keys are {chunk bytes, filter mask, offsets (rank + 1)}
    key = struct.pack("<IIQQ", len(comp), 0, 0, 0)
    last = struct.pack("<IIQQ", 0, 0, n, 0)
    btree = out.put(b"TREE" + struct.pack("<BBHQQ", 1, 0, 1, UNDEF, UNDEF) + key + struct.pack("<Q", chunk_addr) + last)
    layout = struct.pack("<BBB", 3, 2, 2) + struct.pack("<Q", btree) + struct.pack("<II", n, item)
    filt = struct.pack("<BB6x", 1, 1) + struct.pack("<HHHH", 1, 0, 1, 1) + struct.pack("<I", 4) + b"\0\0\0\0"
    msgs = [_msg(0x01, _space_msg((n,))), _msg(0x03, _dtype_msg(arr.dtype)), _msg(0x0B, filt), _msg(0x08, layout)]
    msgs += [_attr(k, v) for k, v in attrs]
    return _header(out, msgs)


def _group(out, tree):
    attrs = [(k[1:], v) for k, v in tree.items() if k.startswith("@")]
    kids = {}
    for k, v in tree.items():
        if k.startswith("@"):
            continue
        if isinstance(v, dict):
            kids[k] = _group(out, v)
        else:
            kids[k] = _dataset(out, v, [])
    names = sorted(kids, key=lambda s: s.encode())
    assert len(names) <= 8, "one symbol node holds 2 K = 8 entries"
    heap = bytearray(8)                                      # offset 0: the empty name
    offs = {}
    for nm in names:
        offs[nm] = len(heap)
        heap += _pad8(nm.encode() + b"\0")
    heap_data = out.put(bytes(heap))
    heap_addr = out.put(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), UNDEF, heap_data))
    snod = b"SNOD" + struct.pack("<BxH", 1, len(names))
    for nm in names:
        snod += struct.pack("<QQII16x", offs[nm], kids[nm], 0, 0)
    snod += b"\0" * (40 * (8 - len(names)))
    snod_addr = out.put(snod)
    btree = out.put(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) +
                    struct.pack("<QQQ", 0, snod_addr, offs[names[-1]] if names else 0))
    msgs = [_msg(0x11, struct.pack("<QQ", btree, heap_addr))] + [_attr(k, v) for k, v in attrs]
    return _header(out, msgs)


def write_hdf5(path, tree):
    out = _Out()
    root = _group(out, tree)
    eof = len(out.b)
    sb = (b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0) +
          struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF) + struct.pack("<QQII16x", 0, root, 0, 0))
    assert len(sb) == 96
    out.b[:96] = sb
    with open(path, "wb") as fh:
        fh.write(bytes(out.b))


if __name__ == "__main__":
    import sys
    write_hdf5(sys.argv[1], {"read_x": {"Raw": {"@read_id": b"x", "Signal": np.arange(100, dtype=np.int16)},
                                        "channel_id": {"@digitisation": 8192.0, "@offset": 3.0, "@range": 1467.61,
                                                       "@sampling_rate": 4000.0}}})
