"""Input side of the two command-line tools: SquigglePull TSV lines, model files,
and (when h5py is importable) fast5 files.

TSV wire format (SquigglePull.py:243-253): one read per line,
    fast5 <tab> readID [<tab> digitisation <tab> offset <tab> range <tab> sampling_rate] <tab> s0 <tab> s1 ...
The consumers disagree about where the samples start -- segmenter reads from
column 4 (segmenter.py:198-201), MotifSeq from column 8 (MotifSeq.py:270) -- and
this module keeps each tool's convention (the `start_col` argument).
"""
import gzip
import math

import numpy as np


def _default_threads():
    """worker threads of the native tokenizer: 32, never more than the host has (SK_TSV_THREADS under SK_TUNING=1:
    measurement runs -- 16 / 32 / 64 / 128 threads on the 256-thread bench host: 349 k / 392 k / 328 k / 245 k reads/s for
    400 000 integer reads through segmenter.py, 210 k / 225 k / 207 k / 154 k for 200 000 pA reads: the threads are
    created per chunk, and beyond 32 that costs more than the parallelism returns)"""
    import os
    from . import _lib
    want = int(_lib.tune("SK_TSV_THREADS", 32))
    return max(1, min(want, os.cpu_count() or 1))


def open_text(path):
    """Plain or gzip text (the reference's dicSwitch, segmenter.py:300-308; its
    segmenter is broken on .gz under Python 3 -- here .gz simply works)."""
    if path.endswith(".gz"):
        return gzip.open(path, "rt")
    return open(path, "rt")


def parse_segmenter_line(line):
    """segmenter.py:192-201: name = col 0; data = cols 4..; float if col 4 has a '.', else int."""
    cols = line.strip("\n").split("\t")
    name = cols[0]
    if "." in cols[4]:
        sig = np.array([float(v) for v in cols[4:]], dtype=float)
    else:
        sig = np.array([int(v) for v in cols[4:]], dtype=int)
    return name, sig


def parse_motifseq_line(line):
    """MotifSeq.py:265-270: fast5 = col 0, readID = col 1, data = float(cols 8..)."""
    cols = line.strip("\n").split("\t")
    return cols[0], cols[1], np.array([float(v) for v in cols[8:]])


class FloatBlock:
    """One whole-line chunk through the float64 tokenizer (sk_tsv_parse): every line's data tokens in ONE flat float64
    array -- line i's at values[off[i]:off[i+1]] -- its name / read-id columns as byte ranges of `buf` (the chunk is
    buf[base:end]: bytes or a read-only memory map, nothing copied), its flags (SK_TSV_*).  What the float64 batch entry
    points take as it stands."""

    def __init__(self, chunk, n, values, off, flags, name_off, name_len, id_off, id_len, centi=None):
        self.buf, self.base, self.end = chunk
        self.n, self._values, self.off, self.flags = n, values, off, flags
        self.name_off, self.name_len, self.id_off, self.id_len = name_off, name_len, id_off, id_len   # relative to base
        # (round 6) every token of the chunk has at most two decimals -- SquigglePull's np.round(pA, 2) -- and was parsed
        # as an int32 centi-unit (sk_tsv_parse_centi): what the batch entry points take as it stands, half the bytes
        self.centi = centi

    @property
    def values(self):
        """float64 values of every token; for a centi chunk made on demand: c / 100.0 IS float("ddd.dd")"""
        if self._values is None:
            self._values = self.centi / 100.0
        return self._values

    def batch_values(self):
        """what the ragged batch entry points should be handed: the int32 centi-units when the chunk has them"""
        return self.centi if self.centi is not None else self.values

    def clean(self):
        """Every line is a plain decimal (or integer) read with a non-zero value somewhere: the reference's own parse
        would give exactly `values` (flags: not SLOW / SHORT, FIRSTDOT or ALLINT, ANY)."""
        f = self.flags
        return bool(self.n) and not np.any(f & 24) and bool(np.all(f & 5)) and bool(np.all(f & 2))

    def spans(self, which):
        """[n, 2] first / one-past-last byte of the column in `buf` (absolute positions)."""
        o, ln = (self.name_off, self.name_len) if which == "name" else (self.id_off, self.id_len)
        a = self.base + o
        return np.stack([a, a + ln.astype(np.int64)], axis=1)

    def text(self, which, i):
        o, ln = (self.name_off, self.name_len) if which == "name" else (self.id_off, self.id_len)
        a = self.base + int(o[i])
        return bytes(self.buf[a:a + int(ln[i])]).decode()


def parse_block_float(chunk, start_col, nthreads):
    """(FloatBlock or None) of a whole-line chunk (buf, start, end); the tokenizer reads the chunk in place."""
    from . import _lib
    L = _lib.load()
    src, start, end = chunk
    cp, clen = _cptr(src, start), end - start
    n = L.sk_tsv_count_lines(cp, clen)
    if n <= 0:
        return None
    ntok = np.zeros(n, dtype=np.int64)
    _lib.check(L.sk_tsv_count_tokens(cp, clen, start_col, n, _lib.ptr(ntok), nthreads))
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(ntok, out=off[1:])
    name_off = np.zeros(n, dtype=np.int64)
    name_len = np.zeros(n, dtype=np.int32)
    id_off = np.zeros(n, dtype=np.int64)
    id_len = np.zeros(n, dtype=np.int32)
    flags = np.zeros(n, dtype=np.int32)
    if _lib.tune("SK_TSV_NO_CENTI") is None:
        # decimal tokens with at most two decimals (what SquigglePull writes) as int32 centi-units, in one pass; a chunk
        # with any other line goes through the general float64 tokenizer below
        centi = np.empty(max(1, int(off[-1])), dtype=np.int32)
        _lib.check(L.sk_tsv_parse_centi(cp, clen, start_col, n, _lib.ptr(off), _lib.ptr(centi),
                                        _lib.ptr(name_off), _lib.ptr(name_len), _lib.ptr(id_off),
                                        _lib.ptr(id_len), _lib.ptr(flags), nthreads))
        if bool(np.all(flags & 32)):
            flags &= ~np.int32(32)
            return FloatBlock(chunk, n, None, off, flags, name_off, name_len, id_off, id_len, centi=centi)
        del centi
    values = np.empty(max(1, int(off[-1])), dtype=np.float64)
    _lib.check(L.sk_tsv_parse(cp, clen, start_col, n, _lib.ptr(off), _lib.ptr(values),
                              _lib.ptr(name_off), _lib.ptr(name_len), _lib.ptr(id_off),
                              _lib.ptr(id_len), _lib.ptr(flags), nthreads))
    return FloatBlock(chunk, n, values, off, flags, name_off, name_len, id_off, id_len)


def _parse_block_float(chunk, start_col, nthreads):
    """The same chunk line by line: (name, read_id, values, flags, raw)."""
    fb = parse_block_float(chunk, start_col, nthreads)
    if fb is None:
        return
    yield from float_block_lines(fb)


def float_block_lines(fb):
    buf, n, values, off, flags = fb.buf, fb.n, fb.values, fb.off, fb.flags
    pos = fb.base
    for i in range(n):
        nl = buf.find(b"\n", pos, fb.end)
        end = nl if nl >= 0 else fb.end
        fl = int(flags[i])
        raw = bytes(buf[pos:end]) if (fl & 24) or not (fl & 5) else None
        yield (fb.text("name", i), fb.text("id", i), values[off[i]:off[i + 1]], fl, raw)
        pos = end + 1


def iter_tsv_native(path, start_col, chunk_bytes=64 << 20, nthreads=None):
    """Stream a SquigglePull TSV through the native tokenizer (csrc/sk_tsv.cpp).

    Yields (name, read_id, values, flags, raw_line) per line, in file order:
      values  float64 array of the columns from start_col on (integers are exact)
      flags   SK_TSV_* bits (ALLINT=1, ANY=2, FIRSTDOT=4, SLOW=8, SHORT=16)
      raw_line the undecoded line (bytes) -- only needed when flags & SLOW/SHORT tells the
               caller to re-parse it the reference's way.
    No GPU is involved; the library only has to be loadable."""
    import os
    nthreads = nthreads or _default_threads()
    for chunk in _line_blocks(path, chunk_bytes):
        yield from _parse_block_float(chunk, start_col, nthreads)


def _line_blocks(path, chunk_bytes):
    """Whole-line chunks of a text file as (buffer, start, end): for plain files a read-only memory map and byte
    ranges into it (nothing is copied on the Python side), for .gz files decompressed bytes."""
    mm = None
    if not path.endswith(".gz"):
        import mmap
        import os
        import stat
        # a memory map only for regular files: a pipe, /dev/stdin or a process substitution (`-s <(zcat x.gz)`)
        # reports size 0 and cannot be mapped -- those are streamed like the .gz branch below, as the reference's
        # line iteration would read them
        try:
            st = os.stat(path)
            if stat.S_ISREG(st.st_mode):
                if st.st_size == 0:
                    return
                with open(path, "rb") as fh:
                    mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
        except (OSError, ValueError):
            mm = None
    if mm is not None:
        size = len(mm)
        pos = 0
        while pos < size:
            end = min(size, pos + chunk_bytes)
            if end < size:
                cut = mm.rfind(b"\n", pos, end)
                while cut < 0 and end < size:                # one line longer than a chunk: extend
                    end = min(size, end + chunk_bytes)
                    cut = mm.rfind(b"\n", pos, end) if end < size else size - 1
                end = cut + 1
            yield mm, pos, end
            pos = end
        return
    with (gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")) as fh:
        tail = b""
        while True:
            block = fh.read(chunk_bytes)
            if not block and not tail:
                break
            buf = tail + block
            if block:
                cut = buf.rfind(b"\n")
                if cut < 0:
                    tail = buf                      # no complete line yet
                    continue
                tail, buf = buf[cut + 1:], buf[:cut + 1]
            else:
                tail = b""
            if buf:
                yield buf, 0, len(buf)


def _cptr(buf, start):
    """C pointer to buf[start] for bytes or a read-only mmap (no copy)."""
    import ctypes as C
    if isinstance(buf, bytes):
        return C.c_void_p(C.cast(C.c_char_p(buf), C.c_void_p).value + start)
    return C.c_void_p(np.frombuffer(buf, dtype=np.uint8).ctypes.data + start)


class TsvBlock:
    """One chunk of a TSV parsed by sk_tsv_parse_i16: lines whose data tokens are all plain integers that fit
    int16 (flags & 1) sit ready in `rows` (int16 [n, stride], `nsamp` tokens each); the others (flags & 8: some
    other token; & 16: no data column) are handed out as raw bytes by line(i) for the reference's own parse."""

    def __init__(self, chunk, rows, nsamp, flags, name_off, name_len, id_off, id_len, line_off):
        self.buf, self.base, self.end = chunk                     # bytes or mmap; this chunk is buf[base:end]
        self.rows, self.nsamp, self.flags = rows, nsamp, flags
        self._no, self._nl, self._io, self._il, self._lo = name_off, name_len, id_off, id_len, line_off
        self.n = len(flags)

    def name(self, i):
        a = self.base + int(self._no[i])
        return self.buf[a:a + int(self._nl[i])].decode()

    def read_id(self, i):
        a = self.base + int(self._io[i])
        return self.buf[a:a + int(self._il[i])].decode()

    def line(self, i):
        return self.buf[self.base + int(self._lo[i]):self.base + int(self._lo[i + 1])].rstrip(b"\n")

    def mostly_integer(self):
        """False for chunks of decimal (pA) lines: those go through the float64 tokenizer instead."""
        return int(np.count_nonzero(self.flags & 8)) * 20 <= self.n

    def float_lines(self, start_col, nthreads=None):
        import os
        return _parse_block_float((self.buf, self.base, self.end), start_col, nthreads or _default_threads())

    def float_block(self, start_col, nthreads=None):
        """The chunk through the float64 tokenizer as ONE FloatBlock (flat values + offsets), or None."""
        import os
        return parse_block_float((self.buf, self.base, self.end), start_col, nthreads or _default_threads())


def _parse_chunk_i16(chunk, start_col, nthreads):
    """One whole-line chunk through the int16 tokenizer: TsvBlock, or None for an empty chunk."""
    from . import _lib
    L = _lib.load()
    src, start, end = chunk
    cp, clen = _cptr(src, start), end - start
    n = L.sk_tsv_count_lines(cp, clen)
    if n <= 0:
        return None
    ntok = np.zeros(n, dtype=np.int64)
    _lib.check(L.sk_tsv_count_tokens(cp, clen, start_col, n, _lib.ptr(ntok), nthreads))
    stride = max(8, (int(ntok.max()) + 7) // 8 * 8)
    if n * stride * 2 > (3 << 30):               # one enormous line among short ones: not worth a dense block
        stride = max(8, (int(np.percentile(ntok, 99)) + 7) // 8 * 8)   # (longer lines are flagged SLOW)
    rows = np.empty((n, stride), dtype=np.int16)
    nsamp = np.zeros(n, dtype=np.int32)
    flags = np.zeros(n, dtype=np.int32)
    name_off = np.zeros(n, dtype=np.int64)
    name_len = np.zeros(n, dtype=np.int32)
    id_off = np.zeros(n, dtype=np.int64)
    id_len = np.zeros(n, dtype=np.int32)
    line_off = np.zeros(n + 1, dtype=np.int64)
    _lib.check(L.sk_tsv_parse_i16(cp, clen, start_col, n, stride, _lib.ptr(rows), _lib.ptr(nsamp),
                                  _lib.ptr(name_off), _lib.ptr(name_len), _lib.ptr(id_off), _lib.ptr(id_len),
                                  _lib.ptr(flags), _lib.ptr(line_off), nthreads))
    return TsvBlock(chunk, rows, nsamp, flags, name_off, name_len, id_off, id_len, line_off)


def iter_tsv_blocks_i16(path, start_col, chunk_bytes=48 << 20, nthreads=None, prefetch=True):
    """Stream a SquigglePull TSV as TsvBlock chunks (csrc/sk_tsv.cpp: sk_tsv_parse_i16): integer lines land in
    int16 rows without a float64 detour or a Python object per read.  With `prefetch` the next chunk is tokenised on
    a background thread (the tokenizer releases the GIL) while the caller works on the current one."""
    if prefetch:
        yield from _prefetched(iter_tsv_blocks_i16(path, start_col, chunk_bytes, nthreads, prefetch=False))
        return
    import os
    from . import _lib
    L = _lib.load()
    nthreads = nthreads or _default_threads()
    for chunk in _line_blocks(path, chunk_bytes):
        blk = _parse_chunk_i16(chunk, start_col, nthreads)
        if blk is not None:
            yield blk


def _first_token_has_dot(src, start, end, start_col):
    """Does the first line's first data token hold a "." -- the reference's own test for "this is a float read"
    (segmenter.py:198; MotifSeq.py:270 parses everything as float anyway)?"""
    nl = src.find(b"\n", start, end)
    line = bytes(src[start:(nl if nl >= 0 else end)][:4096 + 64 * start_col])
    cols = line.split(b"\t", start_col + 1)
    return len(cols) > start_col and b"." in cols[start_col]


def iter_tsv_blocks(path, start_col, chunk_bytes=48 << 20, nthreads=None, prefetch=True):
    """Stream a SquigglePull TSV chunk by chunk: a chunk whose first line starts its data with a decimal token goes
    straight through the float64 tokenizer (FloatBlock: pA files, SquigglePull's default output), any other one through
    the int16 one (TsvBlock).  Either way the chunk is tokenised in place, one chunk ahead on a background thread."""
    if prefetch:
        yield from _prefetched(iter_tsv_blocks(path, start_col, chunk_bytes, nthreads, prefetch=False))
        return
    import os
    nthreads = nthreads or _default_threads()
    for chunk in _line_blocks(path, chunk_bytes):
        src, start, end = chunk
        if _first_token_has_dot(src, start, end, start_col):
            fb = parse_block_float(chunk, start_col, nthreads)
            if fb is not None:
                yield fb
            continue
        blk = _parse_chunk_i16(chunk, start_col, nthreads)
        if blk is not None:
            yield blk


def _prefetched(gen, depth=2):
    """Items of `gen`, produced up to `depth` ahead on a background thread; an exception of the producer is re-raised
    at the place of the item it replaced."""
    import queue
    import threading
    q = queue.Queue(maxsize=depth)
    done = object()

    def run():
        try:
            for item in gen:
                q.put((item, None))
        except BaseException as e:                      # noqa: BLE001 -- handed to the consumer
            q.put((None, e))
        q.put((done, None))
    threading.Thread(target=run, name="sk-tsv-prefetch", daemon=True).start()
    while True:
        item, err = q.get()
        if err is not None:
            raise err
        if item is done:
            return
        yield item


# ----------------------------------------------------------------------------
# motif models
# ----------------------------------------------------------------------------
def read_scrappie_model(path):
    """scrappie CLI squiggle text ("#name", "pos base current sd dwell" rows):
    each k-mer's current repeated round(dwell) times -- the expansion of
    MotifSeq.read_synth_model (MotifSeq.py:354-379).  Returns (models, order, L)."""
    models, order, lens = {}, [], []
    count, name = 0, None
    with open_text(path) as fh:
        for line in fh:
            line = line.strip("\n")
            if not line:
                continue
            if line[0] == "#":
                if name is not None:
                    lens.append(count)
                count = 0
                name = line[1:]
                models[name] = []
                order.append(name)
            elif line[:3] == "pos":
                continue
            else:
                f = line.split()
                count += 1
                models[name] = models[name] + [float(f[2])] * int(round(float(f[4])))
    if name is not None:
        lens.append(count)
    return models, order, lens


def read_bait_model(path):
    """Custom TSV model: name <tab> kmer_length <tab> (ignored) <tab> v0 <tab> v1 ...
    (layout documented at MotifSeq.py:408-428; the reference forgets to fill
    m_order / L_list there, which makes `-m` print only the header -- here the
    order and lengths are filled so `-m` works)."""
    models, order, lens = {}, [], []
    with open_text(path) as fh:
        for line in fh:
            cols = line.strip("\n").split("\t")
            if len(cols) < 4:
                continue
            models[cols[0]] = np.array([float(v) for v in cols[3:]], dtype=float)
            order.append(cols[0])
            lens.append(int(cols[1]))
    return models, order, lens


def read_model_auto(path):
    """'#'-headed files are scrappie text, anything else the bait TSV."""
    with open_text(path) as fh:
        first = fh.readline()
    if first.startswith("#"):
        return read_scrappie_model(path)
    return read_bait_model(path)


def fasta_to_models(path, scrappie_model):
    """MotifSeq.convert_fasta (MotifSeq.py:382-405): scrappy squiggle per record,
    current repeated round(exp(-log_dwell)) times.  Needs the `scrappy` package."""
    import scrappy            # absent here: the caller reports that and exits
    models, order, lens = {}, [], []
    name = None
    with open(path, "r") as fh:
        for line in fh:
            line = line.strip("\n")
            if not line:
                continue
            if line[0] == ">":
                name = line[1:]
                models[name] = []
                order.append(name)
                continue
            lens.append(len(line))
            squiggle = scrappy.sequence_to_squiggle(line, model=scrappie_model).data(as_numpy=True, sloika=False)
            for row in squiggle:
                models[name] = models[name] + [row[0]] * int(round(math.exp(-row[2])))
    return models, order, lens


# ----------------------------------------------------------------------------
# fast5: h5py when it is importable, else the built-in reader (hdf5min.py)
# ----------------------------------------------------------------------------
def _h5py():
    try:
        import h5py
        return h5py
    except ImportError:
        return None


def have_h5py():
    return _h5py() is not None


def open_fast5(path):
    """An h5py.File-shaped handle (groups by name, keys(), Dataset[()], attrs)."""
    h5py = _h5py()
    if h5py is not None:
        return h5py.File(path, "r")
    from . import hdf5min
    return hdf5min.File(path)


def pA(raw, digitisation, range_, offset):
    """convert_to_pA_numpy + round (segmenter.py:515-517,347-349)."""
    return np.round((np.asarray(raw, dtype=int) + offset) * (range_ / digitisation), 2)


def read_single_fast5(path, raw_signal):
    """segmenter.process_fast5 (segmenter.py:321-356): (signal, read_id).  Exceptions propagate; the *_cli
    wrappers below reproduce the reference's try/except messages."""
    with open_fast5(path) as hdf:
        key = list(hdf["Raw/Reads"].keys())[0]
        read = hdf["Raw/Reads/"][key]
        sig = np.array(read["Signal"][()], dtype=int)
        rid = read.attrs["read_id"]
        rid = rid.decode() if isinstance(rid, bytes) else rid
        if not raw_signal:
            ch = hdf["UniqueGlobalKey/channel_id"].attrs
            sig = pA(sig, ch["digitisation"], float("{0:.2f}".format(ch["range"])), ch["offset"])
    return sig, rid


def segmenter_process_fast5(path, raw_signal, err):
    """segmenter.process_fast5 with its error handling (segmenter.py:326-355): a traceback plus one of two
    messages (no newline, like the reference) on `err`, and an empty signal."""
    import traceback
    try:
        hdf = open_fast5(path)
    except Exception:
        traceback.print_exc(file=err)
        err.write("process_fast5():fast5 file failed to open: {}".format(path))
        return np.array([])
    try:
        with hdf:
            key = list(hdf["Raw/Reads"].keys())[0]
            read = hdf["Raw/Reads/"][key]
            sig = np.array(read["Signal"][()], dtype=int)
            read.attrs["read_id"].decode()
            ch = hdf["UniqueGlobalKey/channel_id"].attrs
            dig, off, rng = ch["digitisation"], ch["offset"], float("{0:.2f}".format(ch["range"]))
            if not raw_signal:
                sig = pA(sig, dig, rng, off)
        return sig
    except Exception:
        traceback.print_exc(file=err)
        err.write("process_fast5():failed to extract events or fastq from: {}".format(path))
        return np.array([])


def motifseq_process_fast5(path, err):
    """MotifSeq.process_fast5 (MotifSeq.py:327-350): (list of ints, read_id as the attribute holds it -- bytes,
    which the reference then formats with "{}" -- ); on failure a traceback, the reference's message and ([], "")."""
    import traceback
    try:
        hdf = open_fast5(path)
    except Exception:
        traceback.print_exc(file=err)
        err.write("process_fast5():fast5 file failed to open: {}\n".format(path))
        return [], ""
    try:
        with hdf:
            key = list(hdf["Raw/Reads"].keys())[0]
            read = hdf["Raw/Reads/"][key]
            squig = [int(v) for v in read["Signal"][()]]
            return squig, read.attrs["read_id"]
    except Exception:
        traceback.print_exc(file=err)
        err.write("process_fast5():failed to extract events or fastq from: {}\n".format(path))
        return [], ""


def read_multi_fast5(path, raw_signal, err=None):
    """segmenter.get_multi_fast5_signal (segmenter.py:358-396): {read_name: signal}; a read that cannot be
    extracted gets a traceback + message on `err` and an empty signal, like the reference."""
    import sys
    import traceback
    err = err or sys.stderr
    out = {}
    with open_fast5(path) as hdf:
        for read in list(hdf.keys()):
            try:
                hdf[read]["Raw"].attrs["read_id"].decode()
                ch = hdf[read]["channel_id"].attrs
                dig, off, rng = ch["digitisation"], ch["offset"], float("{0:.2f}".format(ch["range"]))
                ch["sampling_rate"]
                sig = np.array(hdf[read]["Raw/Signal"][()], dtype=int)
                if not raw_signal:
                    sig = pA(sig, dig, rng, off)
            except Exception:
                traceback.print_exc(file=err)
                err.write("extract_fast5():failed to read readID: {}".format(read))
                sig = np.array([], dtype=int)
            out[read] = sig
    return out
