"""Minimal BLOW5 reader (v0.1.0, zlib record compression, no signal compression).

Only what is needed to pull raw int16 squiggles out of files such as the
reference's example/slow5/0.blow5 without pyslow5/slow5lib (absent here).
Layout: 64-byte file header (magic "BLOW5\\x01", version, record compression),
uint32 ASCII-header length, ASCII header, then records
    uint64 size | zlib{ uint16 idlen | id | uint32 read_group | f64 digitisation |
                        f64 offset | f64 range | f64 sampling_rate | uint64 n |
                        int16[n] | aux... }
The reference reaches the same data through pyslow5 (SquigglePlot.py:229-263,
dRNA_segmenter.py:85-100).
"""
import struct
import zlib

import numpy as np

MAGIC = b"BLOW5\x01"


def read_blow5(path):
    """Yield dicts {read_id, digitisation, offset, range, sampling_rate, signal(int16)}."""
    with open(path, "rb") as fh:
        buf = fh.read()
    if buf[:6] != MAGIC:
        raise ValueError("not a BLOW5 file: %s" % path)
    major, minor, patch, comp = struct.unpack_from("<BBBB", buf, 6)
    if comp not in (0, 1):
        raise ValueError("unsupported BLOW5 record compression %d (only none/zlib)" % comp)
    # byte 10 is the signal-compression method in >= 0.2.0; 0.1.0 has none
    if (major, minor) >= (0, 2) and buf[10] != 0:
        raise ValueError("unsupported BLOW5 signal compression %d" % buf[10])
    (hlen,) = struct.unpack_from("<I", buf, 64)
    pos = 68 + hlen
    eof_marker = b"5WOLB"
    while pos + 8 <= len(buf):
        if buf[pos:pos + 5] == eof_marker:
            break
        (size,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        rec = buf[pos:pos + size]
        pos += size
        if comp == 1:
            rec = zlib.decompress(rec)
        (idlen,) = struct.unpack_from("<H", rec, 0)
        rid = rec[2:2 + idlen].decode().rstrip("\x00")
        o = 2 + idlen
        rg, dig, off, rng_, rate, n = struct.unpack_from("<IddddQ", rec, o)
        o += struct.calcsize("<IddddQ")
        sig = np.frombuffer(rec, dtype="<i2", count=n, offset=o).copy()
        yield {"read_id": rid, "read_group": rg, "digitisation": dig, "offset": off,
               "range": rng_, "sampling_rate": rate, "signal": sig}


def to_pA(raw, digitisation, offset, range_):
    """SquigglePull's pA conversion (SquigglePull.py:183-189,238-240):
    round((raw + offset) * (round(range, 2) / digitisation), 2)."""
    range2 = float("{0:.2f}".format(range_))
    raw_unit = range2 / digitisation
    return np.round((np.asarray(raw, dtype=np.int64) + offset) * raw_unit, 2)


def read_slow5_ascii(path):
    """ASCII SLOW5: '#'/'@' header lines, a '#read_id<TAB>...' column line, then one read per
    line with the raw signal as a comma separated list in the `raw_signal` column."""
    cols = None
    with open(path, "rt") as fh:
        for line in fh:
            line = line.rstrip("\n")
            if not line:
                continue
            if line.startswith("#read_id"):
                cols = line[1:].split("\t")
                continue
            if line[0] in "#@":
                continue
            if cols is None:
                raise ValueError("SLOW5 column header missing: %s" % path)
            f = dict(zip(cols, line.split("\t")))
            sig = np.array(f["raw_signal"].split(","), dtype=np.int64).astype(np.int16)
            yield {"read_id": f["read_id"], "read_group": int(f.get("read_group", 0)),
                   "digitisation": float(f["digitisation"]), "offset": float(f["offset"]),
                   "range": float(f["range"]), "sampling_rate": float(f["sampling_rate"]), "signal": sig}


def read_slow5(path):
    """BLOW5 (binary) or SLOW5 (ASCII), chosen by the file's magic."""
    with open(path, "rb") as fh:
        head = fh.read(6)
    return read_blow5(path) if head == MAGIC else read_slow5_ascii(path)
