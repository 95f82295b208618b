"""Host-side fast paths of the command-line tools (csrc/sk_io.cpp; no GPU involved): result tables formatted
natively with Python's own float formatting, and BLOW5 records decoded straight into int16 rows.

The reference prints one row per read with `"\\t".join("{}".format(v) ...)` (MotifSeq.py:446-449,
segmenter.py:222-227) and reads BLOW5 through pyslow5 one record at a time (segmenter.py:321-396); at a few
microseconds of interpreter time per read neither keeps up with kernels that finish millions of reads per second.
"""
import ctypes as C
import mmap
import os
import struct

import numpy as np

from . import _lib

STR, I32, F64, CONST, I32LIST, STRSPAN = 0, 1, 2, 3, 4, 5


class _Col(C.Structure):
    _fields_ = [("kind", C.c_int32), ("data", C.c_void_p), ("off", C.c_void_p)]


def _addr(buf):
    """Address of a bytes / mmap / numpy buffer (no copy)."""
    if isinstance(buf, np.ndarray):
        return buf.ctypes.data
    if isinstance(buf, bytes):
        return C.cast(C.c_char_p(buf), C.c_void_p).value
    return np.frombuffer(buf, dtype=np.uint8).ctypes.data


def fmt_rows(nrows, cols, skip=None, nthreads=0):
    """The table as bytes: one line per row, columns joined by tabs.  cols: list of
         ("str", blob, off)        strings blob[off[i]:off[i+1]]             (off: int64[nrows + 1])
         ("span", buf, spans)      strings buf[spans[i,0]:spans[i,1]]        (spans: int64[nrows, 2]; buf bytes / mmap)
         ("const", bytes)          the same string in every row
         ("i32", int32[nrows])     ("f64", float64[nrows])  -- floats as Python's repr() writes them
         ("i32list", values, off)  comma-joined int32 values[off[i]:off[i+1]]
    skip: optional uint8[nrows], rows with a non-zero entry are left out."""
    L = _lib.load()
    arr = (_Col * len(cols))()
    keep = []
    for k, c in enumerate(cols):
        kind = c[0]
        if kind == "str":
            off = np.ascontiguousarray(c[2], dtype=np.int64)
            keep += [c[1], off]
            arr[k] = _Col(STR, _addr(c[1]), off.ctypes.data)
        elif kind == "span":
            sp = np.ascontiguousarray(c[2], dtype=np.int64)
            keep += [c[1], sp]
            arr[k] = _Col(STRSPAN, _addr(c[1]), sp.ctypes.data)
        elif kind == "const":
            off = np.array([0, len(c[1])], dtype=np.int64)
            keep += [c[1], off]
            arr[k] = _Col(CONST, _addr(c[1]), off.ctypes.data)
        elif kind == "i32":
            a = np.ascontiguousarray(c[1], dtype=np.int32)
            keep.append(a)
            arr[k] = _Col(I32, a.ctypes.data, None)
        elif kind == "f64":
            a = np.ascontiguousarray(c[1], dtype=np.float64)
            keep.append(a)
            arr[k] = _Col(F64, a.ctypes.data, None)
        elif kind == "i32list":
            a = np.ascontiguousarray(c[1], dtype=np.int32)
            off = np.ascontiguousarray(c[2], dtype=np.int64)
            keep += [a, off]
            arr[k] = _Col(I32LIST, a.ctypes.data, off.ctypes.data)
        else:
            raise ValueError("unknown column kind %r" % (kind,))
    sk = None
    if skip is not None:
        sk = np.ascontiguousarray(skip, dtype=np.uint8)
        keep.append(sk)
    n = C.c_int64(0)
    p = L.sk_fmt_rows(int(nrows), len(cols), C.cast(arr, C.c_void_p), None if sk is None else sk.ctypes.data,
                      int(nthreads), C.byref(n))
    if not p:
        raise MemoryError("sk_fmt_rows failed")
    try:
        return C.string_at(p, n.value)
    finally:
        L.sk_fmt_free(p)


def ndtr(z):
    """scipy.special.ndtr without scipy (csrc/sk_io.cpp restates the Cephes routine; the same doubles, bit for bit):
    an array or a scalar in, the same shape out."""
    a = np.asarray(z, dtype=np.float64)
    flat = np.ascontiguousarray(a).reshape(-1)
    out = np.empty_like(flat)
    _lib.load().sk_ndtr(flat.ctypes.data, out.ctypes.data, flat.size)
    return out.reshape(a.shape) if a.ndim else np.float64(out[0])


def write_stdout(text):
    """Bytes to stdout behind whatever print() already queued (a redirected / captured stdout may be a text-only
    stream without a byte layer)."""
    import sys
    sys.stdout.flush()
    raw = getattr(sys.stdout, "buffer", None)
    if raw is not None:
        raw.write(text)
    else:
        sys.stdout.write(text.decode())


# ------------------------------------------------------------------------------------------------
# BLOW5
# ------------------------------------------------------------------------------------------------
class Blow5Block:
    """A run of decoded records: rows int16 [n, stride] (nsamp[i] samples each), ids (numpy 'S' array), calib [n, 3]
    = digitisation, offset, range; flags[i] & 1: read longer than a row (truncated), & 2: unreadable record."""

    def __init__(self, rows, nsamp, ids, calib, flags):
        self.rows, self.nsamp, self.ids, self.calib, self.flags = rows, nsamp, ids, calib, flags
        self.n = len(nsamp)


def blow5_open(path):
    """(mmap, record compression, offset of the first record)."""
    fh = open(path, "rb")
    mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
    fh.close()
    if mm[:6] != b"BLOW5\x01":
        raise ValueError("not a BLOW5 file: %s" % path)
    major, minor, _patch, comp = struct.unpack_from("<BBBB", mm, 6)
    if comp not in (0, 1):
        raise ValueError("unsupported BLOW5 record compression %d (only none / zlib)" % comp)
    if (major, minor) >= (0, 2) and mm[10] != 0:
        raise ValueError("unsupported BLOW5 signal compression %d" % mm[10])
    (hlen,) = struct.unpack_from("<I", mm, 64)
    return mm, comp, 68 + hlen


def iter_blow5_blocks_i16(path, block_reads=16384, id_width=64, nthreads=0, keep=None):
    """Stream a BLOW5 file as Blow5Block chunks (csrc/sk_io.cpp): the records of a chunk are indexed, then decoded
    on all cores straight from the file mapping into int16 rows whose stride fits the chunk's longest read -- both
    one chunk ahead of the caller, on a background thread.  `keep`: a list that takes over the file mapping and the
    buffers at the end instead of their being released there (a command-line tool about to exit: the unmapping holds
    up a GPU call still in flight).
    The rows are ordinary memory, not page-locked (SK_BLOW5_PIN=1 pins them): the H2D copy of a chunk is hidden
    behind the decoding of the next either way, while pinning three 300 MB buffers cost ~60 ms each when first used
    and ~0.1 s more when the process exits -- 1 M reads x 4 000 samples: 0.72-0.75 s (segmenter.py) / 0.96-1.07 s
    (MotifSeq.py) unpinned against 0.98-1.03 / 1.16-1.19 s pinned, process start to exit, same box."""
    L = _lib.load()
    if _lib.tune("SK_BLOW5_BLOCK"):
        block_reads = max(1, int(_lib.tune("SK_BLOW5_BLOCK")))
    mm, comp, first = blow5_open(path)
    base = _addr(mm)
    flen = len(mm)
    from concurrent.futures import ThreadPoolExecutor
    pools = [{}, {}, {}]                                       # three sets of buffers, reused (fresh pages cost page
                                                               # faults): a GPU call on the previous block, the block
                                                               # the caller holds, the one being decoded

    def buf(pool, name, shape, dtype, pinned=False):
        need = int(np.prod(shape)) * np.dtype(dtype).itemsize
        b = pool.get(name)
        # (page-locked rows on request, once a GPU is bound: the first chunks of a tool that is still starting the
        # HIP runtime are decoded into ordinary memory meanwhile)
        want_pinned = pinned and _lib.tune("SK_BLOW5_PIN", "0") == "1" and _lib.is_ready()
        if b is None or b.nbytes < need or (want_pinned and not pool.get(name + ":pinned")):
            cap = max(need + need // 8, 1)
            b = None
            if want_pinned:
                try:
                    from . import api
                    b = api.pinned_empty((cap,), np.uint8)
                except Exception:                              # noqa: BLE001 -- no pinned memory: ordinary pages
                    b = None
            pool[name + ":pinned"] = want_pinned               # (asked once per pool, whatever came of it)
            if b is None:
                b = np.empty(cap, dtype=np.uint8)
            pool[name] = b
        return b[:need].view(dtype).reshape(shape)

    def decode(pos, pool):
        """Index and decode the chunk that starts at byte `pos`: (block or None, position of the next chunk)."""
        off = buf(pool, "off", (block_reads,), np.int64)
        size = buf(pool, "size", (block_reads,), np.int64)
        nxt = C.c_int64(0)
        n = L.sk_blow5_index_some(base, flen, pos, block_reads, off.ctypes.data, size.ctypes.data, C.byref(nxt))
        if n < 0:
            raise ValueError("truncated BLOW5 file: %s" % path)
        if n == 0:
            return None, nxt.value
        # stored records: the signal is all of the payload but ~60 bytes of fixed fields, the id and aux data, so
        # size / 2 bounds the sample count; zlib: start from the compressed size and grow if a read does not fit
        guess = int(size[:n].max()) // 2 if comp == 0 else int(size[:n].max()) * 2
        stride = max(8, (guess + 7) // 8 * 8)
        idw = id_width
        while True:
            rows = buf(pool, "rows", (n, stride), np.int16, pinned=True)    # (valid until three blocks later)
            nsamp = buf(pool, "nsamp", (n,), np.int32)
            ids = buf(pool, "ids", (n,), "S%d" % idw)
            calib = buf(pool, "calib", (n, 3), np.float64)
            flags = buf(pool, "flags", (n,), np.int32)
            _lib.check(L.sk_blow5_rows_i16(base, flen, off.ctypes.data, size.ctypes.data, n, comp, stride,
                                           rows.ctypes.data, nsamp.ctypes.data, ids.ctypes.data, idw,
                                           calib.ctypes.data, flags.ctypes.data, int(nthreads)))
            if comp == 1 and np.any(flags & 1) and stride < (1 << 24):
                stride = (int(nsamp.max()) + 7) // 8 * 8          # (nsamp holds the true lengths)
                continue
            if np.any(flags & 4) and idw < 65536:                 # a read id longer than the column: decode again, wider
                idw *= 4                                          # (idlen is a uint16: 65 536 always fits)
                continue
            break
        zapper.submit(forget, pos, nxt.value)
        return Blow5Block(rows, nsamp, ids, calib, flags), (nxt.value if n == block_reads else -1)

    # A decoded chunk's pages are dropped from this process's mapping at once (they stay in the page cache), on a
    # thread of their own: left mapped, 8 GB of page-table entries are torn down when the process exits, serially,
    # while whoever started the tool waits (0.15-0.2 s per million 4 000-sample reads).
    page = mmap.PAGESIZE
    try:
        libc = C.CDLL(None, use_errno=True)
        libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        dontneed = mmap.MADV_DONTNEED
    except (OSError, AttributeError):                          # no madvise here: the pages go when the process does
        libc = None

    def forget(a, b):
        a = (a + page - 1) // page * page
        b = b // page * page
        if libc is not None and b > a and _lib.tune("SK_BLOW5_ZAP", "1") != "0":
            libc.madvise(base + a, b - a, dontneed)

    with ThreadPoolExecutor(1) as ex, ThreadPoolExecutor(1) as zapper:
        k = 0
        fut = ex.submit(decode, first, pools[0])
        while fut is not None:
            blk, nxt = fut.result()
            k = (k + 1) % 3
            fut = ex.submit(decode, nxt, pools[k]) if (blk is not None and nxt >= 0) else None
            if blk is not None:
                yield blk
    if keep is not None:
        keep.append((mm, pools))


def iter_npy_blocks_i16(path, block_bytes=128 << 20, nthreads=8, keep=None):
    """Stream a .npy file holding an int16 array [reads, samples] as (first_row, rows) blocks: parallel preads into
    three reused buffers, the next block being read while the caller works on the current one.  (A memory map would
    cost a page fault per 4 KB.)  The buffers are ordinary memory unless SK_I16_PIN=1: page-locked ones make the H2D
    copy faster, but that copy is hidden behind the reading of the next block either way, and pinning 3 x 256 MB
    cost more at first use and at process exit than it saved -- 1 M reads x 4 000 samples, process start to exit,
    same box: 0.57-0.59 s (segmenter.py) / 0.78-0.80 s (MotifSeq.py) against 0.74-0.76 / 0.89-0.94 s pinned."""
    from concurrent.futures import ThreadPoolExecutor
    from . import api
    with open(path, "rb") as fh:
        major, minor = np.lib.format.read_magic(fh)
        shape, fortran, dtype = (np.lib.format.read_array_header_1_0(fh) if major == 1
                                 else np.lib.format.read_array_header_2_0(fh))
        data0 = fh.tell()
    if len(shape) != 2 or dtype != np.int16 or fortran:
        raise ValueError("%s: need a C-ordered 2-D int16 array, got %s %s" % (path, dtype, shape))
    R, M = shape
    if _lib.tune("SK_I16_BLOCK_MB"):
        block_bytes = max(1, int(_lib.tune("SK_I16_BLOCK_MB"))) << 20
    per = max(1, block_bytes // max(1, M * 2))
    fd = os.open(path, os.O_RDONLY)
    def alloc():
        if _lib.tune("SK_I16_PIN", "0") == "1":
            try:
                return api.pinned_empty((min(per, max(R, 1)), M), np.int16)
            except Exception:                                        # noqa: BLE001 -- no device yet / no pinned memory
                pass
        return np.empty((min(per, max(R, 1)), M), dtype=np.int16)
    # three buffers: the caller may still have a GPU call in flight on the previous block while it holds the
    # current one and the next is being read
    nbuf = 3 if R > 2 * per else (2 if R > per else 1)
    bufs = [alloc() for _ in range(nbuf)]
    ex = ThreadPoolExecutor(max(1, nthreads))

    def fill(k, lo):
        n = min(per, R - lo)
        flat = bufs[k][:n].reshape(-1).view(np.uint8)
        nb = flat.size
        step = (nb + nthreads - 1) // nthreads // 4096 * 4096 + 4096

        def part(a):
            b = min(nb, a + step)
            mv = memoryview(flat)[a:b]
            got = 0
            while got < b - a:
                r = os.preadv(fd, [mv[got:]], data0 + lo * M * 2 + a + got)
                if r <= 0:
                    raise IOError("short read from %s" % path)
                got += r
        return [ex.submit(part, a) for a in range(0, nb, step)], n

    try:
        pending = fill(0, 0) if R else None
        lo, k = 0, 0
        while pending is not None:
            futs, n = pending
            for f in futs:
                f.result()
            nxt = lo + n
            pending = fill((k + 1) % nbuf, nxt) if nxt < R else None
            yield lo, bufs[k][:n]
            lo, k = nxt, (k + 1) % nbuf
    finally:
        ex.shutdown(wait=True)
        os.close(fd)
        if keep is not None:                                         # (see iter_blow5_blocks_i16)
            keep.append(bufs)


def write_blow5(path, reads, read_ids=None, compress=False):
    """A minimal BLOW5 0.2.0 file (tools / tests): `reads` = int16 arrays (or a 2-D array), stored or zlib records."""
    import zlib
    hdr = b"#slow5_version\t0.2.0\n#num_read_groups\t1\n@asic_id\t0\n#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\t" \
          b"uint64_t\tint16_t*\n#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\t" \
          b"raw_signal\n"
    with open(path, "wb") as fh:
        head = b"BLOW5\x01" + bytes([0, 2, 0, 1 if compress else 0, 0])
        fh.write(head + b"\0" * (64 - len(head)))
        fh.write(struct.pack("<I", len(hdr)) + hdr)
        if isinstance(reads, np.ndarray) and reads.ndim == 2 and not compress and read_ids is None and len(reads):
            # equal-length stored records of a 2-D array: assembled as one byte matrix, a block of reads at a time
            # (a million struct.pack calls take longer than the tools take to read the file)
            R, M = reads.shape
            idw = len("read%d" % (R - 1))
            fixed = struct.pack("<IddddQ", 0, 8192.0, 10.0, 1400.0, 4000.0, M)
            rec = 2 + idw + len(fixed) + 2 * M
            for lo in range(0, R, 16384):
                n = min(16384, R - lo)
                blk = np.zeros((n, 8 + rec), dtype=np.uint8)
                blk[:, :8] = np.frombuffer(struct.pack("<Q", rec), dtype=np.uint8)
                blk[:, 8:10] = np.frombuffer(struct.pack("<H", idw), dtype=np.uint8)
                ids = np.array(["read%d" % i for i in range(lo, lo + n)], dtype="S%d" % idw)   # (NUL padded: the
                blk[:, 10:10 + idw] = ids.view(np.uint8).reshape(n, idw)                        # readers strip it)
                blk[:, 10 + idw:10 + idw + len(fixed)] = np.frombuffer(fixed, dtype=np.uint8)
                blk[:, 10 + idw + len(fixed):] = np.ascontiguousarray(reads[lo:lo + n], dtype="<i2").view(np.uint8).reshape(n, 2 * M)
                blk.tofile(fh)
            fh.write(b"5WOLB")
            return path
        for i, sig in enumerate(reads):
            sig = np.ascontiguousarray(sig, dtype="<i2")
            rid = (read_ids[i] if read_ids is not None else "read%d" % i).encode()
            rec = struct.pack("<H", len(rid)) + rid + struct.pack("<IddddQ", 0, 8192.0, 10.0, 1400.0, 4000.0, sig.size) \
                + sig.tobytes()
            if compress:
                rec = zlib.compress(rec, 1)
            fh.write(struct.pack("<Q", len(rec)) + rec)
        fh.write(b"5WOLB")
    return path
