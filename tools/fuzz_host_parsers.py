#!/usr/bin/env python3
"""Feed the native host parsers -- csrc/sk_tsv.cpp (TSV tokenizer), csrc/sk_io.cpp (BLOW5 decoder, table formatter) --
and the oracle a corpus of malformed input under AddressSanitizer + UndefinedBehaviorSanitizer.

    make -C squigglekit_amd/csrc asan
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/fuzz_host_parsers.py [rounds]

The libraries are the sanitizer builds (squigglekit_amd/libsk_host_asan.so, oracle/libsk_oracle_asan.so), loaded
directly: no GPU, no HIP.  Any sanitizer report aborts the process (non-zero exit); the script itself checks that
what comes back is sane (flags set, counts inside the buffers).  tests/test_sanitizers.py runs this with a small round
count; a longer run: pass a number.

Corpus (formats: SquigglePull.py:243-253 for the TSV; SURVEY section 4.2 for BLOW5):
  TSV    truncated last line, no trailing newline, CRLF, NUL bytes, empty tokens, tokens outside the plain grammar
         (1e400, nan, 0x10, "-", "+."), 400-digit integers, a line of 2^20 columns, start columns past the line's end
  BLOW5  file cut inside a size field / inside a record / behind the last record, idlen past the record's end,
         sample count larger than the payload, a zlib record that inflates to 64 MB, record offsets outside the
         buffer handed to sk_blow5_rows_i16, ids longer than the id column, and seeded byte-level mutations
         (flips, truncations, splices) of the reference's example/slow5/0.blow5 (copy: tests/golden/example_0.blow5)
"""
import ctypes as C
import os
import random
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_vp = C.c_void_p


def load():
    L = C.CDLL(os.path.join(ROOT, "squigglekit_amd", "libsk_host_asan.so"))
    L.sk_tsv_count_lines.restype = C.c_int64
    L.sk_tsv_count_lines.argtypes = [_vp, C.c_size_t]
    L.sk_tsv_count_tokens.argtypes = [_vp, C.c_size_t, C.c_int32, C.c_int64, _vp, C.c_int32]
    L.sk_tsv_parse.argtypes = [_vp, C.c_size_t, C.c_int32, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int32]
    L.sk_tsv_parse_i16.argtypes = [_vp, C.c_size_t, C.c_int32, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, C.c_int32]
    L.sk_blow5_index.restype = C.c_int64
    L.sk_blow5_index.argtypes = [_vp, C.c_int64, C.c_int64, _vp, _vp, C.c_int64]
    L.sk_blow5_index_some.restype = C.c_int64
    L.sk_blow5_index_some.argtypes = [_vp, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, _vp]
    L.sk_blow5_rows_i16.argtypes = [_vp, C.c_int64, _vp, _vp, C.c_int64, C.c_int32, C.c_int64, _vp, _vp, _vp, C.c_int32,
                                    _vp, _vp, C.c_int32]
    L.sk_fmt_rows.restype = _vp
    L.sk_fmt_rows.argtypes = [C.c_int64, C.c_int32, _vp, _vp, C.c_int32, C.POINTER(C.c_int64)]
    L.sk_fmt_free.argtypes = [_vp]
    L.sk_ndtr.argtypes = [_vp, _vp, C.c_int64]
    return L


def p(a):
    return a.ctypes.data_as(_vp)


# ------------------------------------------------------------------------------------------------------- TSV
def tsv_case(L, data, start_col, nthreads):
    """Everything the CLIs do with a chunk, on an exactly-sized heap copy (so that an overrun hits a red zone)."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8).copy() if len(data) else np.zeros(0, dtype=np.uint8)
    n = L.sk_tsv_count_lines(p(buf), buf.size)
    assert 0 <= n <= buf.size + 1, n
    if n == 0:
        return
    ntok = np.zeros(n, dtype=np.int64)
    rc = L.sk_tsv_count_tokens(p(buf), buf.size, start_col, n, p(ntok), nthreads)
    assert rc == 0, rc
    assert ntok.min() >= 0 and ntok.sum() <= buf.size + n
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(ntok, out=off[1:])
    values = np.empty(max(1, int(off[-1])), dtype=np.float64)
    name_off, id_off = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    name_len, id_len = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    flags = np.zeros(n, dtype=np.int32)
    rc = L.sk_tsv_parse(p(buf), buf.size, start_col, n, p(off), p(values), p(name_off), p(name_len), p(id_off),
                        p(id_len), p(flags), nthreads)
    assert rc == 0, rc
    assert np.all(name_off + name_len <= buf.size) and np.all(id_off + id_len <= buf.size)
    # every stride the CLI could pick, including one shorter than the longest line (those lines get SK_TSV_SLOW)
    for stride in sorted({8, max(8, (int(ntok.max()) + 7) // 8 * 8), max(8, int(np.median(ntok)) // 8 * 8)}):
        if n * stride * 2 > (1 << 28):
            continue
        rows = np.empty((n, stride), dtype=np.int16)
        nsamp = np.zeros(n, dtype=np.int32)
        line_off = np.zeros(n + 1, dtype=np.int64)
        rc = L.sk_tsv_parse_i16(p(buf), buf.size, start_col, n, stride, p(rows), p(nsamp), p(name_off), p(name_len),
                                p(id_off), p(id_len), p(flags), p(line_off), nthreads)
        assert rc == 0, rc
        assert nsamp.min() >= 0 and np.all((nsamp <= stride) | (flags & 24 != 0)), (stride, nsamp.max())
        assert np.all(line_off[:n] <= buf.size)


def tsv_corpus(rng):
    good = b"f.fast5\tid1\t" + b"\t".join(str(rng.randrange(300, 700)).encode() for _ in range(50))
    pa = b"f.fast5\tid2\t" + b"\t".join(("%.2f" % rng.uniform(60, 130)).encode() for _ in range(50))
    cases = [b"", b"\n", b"\n\n\n", b"\t", b"\t\t\t\n", good, good + b"\n", good[:37], good + b"\r\n" + pa + b"\r\n",
             good.replace(b"\t", b"\0", 3) + b"\n", b"\0" * 100, b"\0\n\0\n", pa + b"\n" + good,
             b"a\tb\t" + b"\t".join([b"1e400", b"-1e400", b"nan", b"inf", b"0x10", b"-", b"+.", b".", b"1.", b".5",
                                     b"1e", b"1e+", b"--3", b"+-3", b" 12", b"12 ", b"1_000", b"9" * 400,
                                     b"-" + b"9" * 400, b"0." + b"0" * 400 + b"1", b"32767", b"32768", b"-32768",
                                     b"-32769", b"1e-400", b"4.9e-324", b"1.7976931348623157e308", b"2e308"]) + b"\n",
             b"x\ty\t" + b"\t".join([b"7"] * (1 << 20)) + b"\n" + good + b"\n",          # one enormous line
             b"\n".join([good] * 300), b"\t" * 5000, b"1\t2\n" * 2000, b"\r\n" * 50, b"\xff\xfe" * 64 + b"\n"]
    for c in cases:
        yield c
    for _ in range(40):                                     # spliced / truncated / bit-flipped good lines
        b = bytearray((good + b"\n" + pa + b"\n") * rng.randrange(1, 4))
        for _k in range(rng.randrange(1, 12)):
            b[rng.randrange(len(b))] = rng.choice(b"\t\n\r\0 .-+e9xA")
        yield bytes(b[:rng.randrange(1, len(b) + 1)])


# ------------------------------------------------------------------------------------------------------- BLOW5
def blow5_file(records, comp=0, marker=True, version=(0, 2, 0)):
    head = b"#slow5_version\t0.2.0\n#num_read_groups\t1\n#read_id\n"
    out = bytearray(b"BLOW5\x01" + bytes(version) + bytes([comp, 0]))
    out += b"\0" * (64 - len(out))
    out += struct.pack("<I", len(head)) + head
    first = len(out)
    for r in records:
        body = zlib.compress(r) if comp == 1 else r
        out += struct.pack("<Q", len(body)) + body
    if marker:
        out += b"5WOLB"
    return bytes(out), first


def record(read_id, samples, n_claim=None, idlen_claim=None):
    rid = read_id if isinstance(read_id, bytes) else read_id.encode()
    sig = np.asarray(samples, dtype="<i2").tobytes()
    return (struct.pack("<H", len(rid) if idlen_claim is None else idlen_claim) + rid + struct.pack("<I", 0) +
            struct.pack("<dddd", 8192.0, 16.0, 1493.94, 4000.0) +
            struct.pack("<Q", len(samples) if n_claim is None else n_claim) + sig)


def blow5_case(L, data, first, comp, nthreads=4, lie=None):
    buf = np.frombuffer(bytes(data), dtype=np.uint8).copy() if len(data) else np.zeros(0, dtype=np.uint8)
    n = L.sk_blow5_index(p(buf), buf.size, first, None, None, 0)
    cap = 4096
    off, size = np.zeros(cap, dtype=np.int64), np.zeros(cap, dtype=np.int64)
    nxt = C.c_int64(0)
    m = L.sk_blow5_index_some(p(buf), buf.size, first, cap, p(off), p(size), C.byref(nxt))
    if n >= 0 and n <= cap:
        assert m == n, (m, n)
    if m < 0:
        # an invalid file: the decoder must still survive whatever offsets a caller hands it
        m, off[:4], size[:4] = 4, [first, buf.size - 3, -5, 1 << 60], [buf.size, 100, 7, 9]
    if lie is not None:
        off[:m], size[:m] = lie(off[:m].copy(), size[:m].copy(), buf.size)
    m = int(m)
    for stride, idw in ((8, 1), (64, 4), (4096, 64)):
        rows = np.empty((max(1, m), stride), dtype=np.int16)
        nsamp = np.zeros(max(1, m), dtype=np.int32)
        ids = np.zeros(max(1, m) * idw, dtype=np.uint8)
        calib = np.zeros((max(1, m), 3), dtype=np.float64)
        flags = np.zeros(max(1, m), dtype=np.int32)
        rc = L.sk_blow5_rows_i16(p(buf), buf.size, p(off), p(size), m, comp, stride, p(rows), p(nsamp), p(ids), idw,
                                 p(calib), p(flags), nthreads)
        assert rc == 0, rc
        assert np.all(nsamp[:m] >= 0)
        assert np.all((nsamp[:m] <= stride) | (flags[:m] & 1 != 0))
    return n


def blow5_corpus(rng, example):
    sig = [rng.randrange(300, 700) for _ in range(500)]
    recs = [record("read%d" % i, sig[:100 + 37 * i]) for i in range(6)]
    for comp in (0, 1):
        data, first = blow5_file(recs, comp)
        yield "good", data, first, comp, None
        yield "no marker", blow5_file(recs, comp, marker=False)[0], first, comp, None
        for cut in (1, 3, 5, 7, 9, 40, len(data) // 2):
            yield "cut %d" % cut, data[:len(data) - cut], first, comp, None
        yield "junk behind", data + b"junk", first, comp, None
        yield "size field cut", data[:first + 5], first, comp, None
        yield "huge size", data[:first] + struct.pack("<Q", 1 << 62) + data[first + 8:], first, comp, None
        yield "offsets outside", data, first, comp, lambda o, s, n: (o + n, s)
        yield "sizes past the end", data, first, comp, lambda o, s, n: (o, s + n)
        yield "negative", data, first, comp, lambda o, s, n: (-o - 1, -s - 1)
    bad = [record("x", sig[:50], idlen_claim=60000), record("y", sig[:50], n_claim=1 << 40),
           record("z", sig[:50], n_claim=51), record(b"q" * 300, sig[:20]), record("", []), b"", b"\x01",
           record(b"nul\0\0\0", sig[:9])]
    for comp in (0, 1):
        data, first = blow5_file(bad, comp)
        yield "bad records", data, first, comp, None
    bomb = zlib.compress(b"\0" * (64 << 20), 9)
    data, first = blow5_file([], 1)
    data = data[:-5] + struct.pack("<Q", len(bomb)) + bomb + b"5WOLB"
    yield "zlib bomb", data, first, 1, None
    yield "garbage deflate", blow5_file([], 1)[0][:-5] + struct.pack("<Q", 64) + bytes(rng.randrange(256) for _ in range(64)) + b"5WOLB", first, 1, None
    if example is not None:
        (hlen,) = struct.unpack_from("<I", example, 64)
        efirst, ecomp = 68 + hlen, example[9]
        yield "example", example, efirst, ecomp, None
        for k in range(rounds_for_mutations()):
            b = bytearray(example)
            kind = rng.randrange(4)
            if kind == 0:
                for _ in range(rng.randrange(1, 16)):
                    b[rng.randrange(efirst, len(b))] = rng.randrange(256)
            elif kind == 1:
                b = b[:rng.randrange(efirst, len(b))]
            elif kind == 2:
                a = rng.randrange(efirst, len(b))
                b[a:a + rng.randrange(1, 64)] = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 64)))
            else:
                a = rng.randrange(efirst, len(b) - 8)
                b[a:a + 8] = struct.pack("<Q", rng.choice([0, 1, 2, 1 << 31, (1 << 63) - 1, (1 << 64) - 1, len(b)]))
            yield "mutation %d/%d" % (k, kind), bytes(b), efirst, ecomp, None


_ROUNDS = 200


def rounds_for_mutations():
    return _ROUNDS


# ------------------------------------------------------------------------------------------------------- formatter / oracle
class _Col(C.Structure):
    _fields_ = [("kind", C.c_int32), ("data", C.c_void_p), ("off", C.c_void_p)]


def fmt_case(L, rng, nrows):
    """sk_fmt_rows over every column kind (exactly-sized heap copies), the float column against Python's own repr()."""
    f = np.array([rng.choice([0.0, -0.0, 1e-320, 5e-324, 1e22, 1e21, 1e16, 123456789012345680.0, 0.1, 1 / 3, 2.5e-5,
                              1e-4, 9.999e-5, float("nan"), float("inf"), float("-inf"), rng.uniform(-1e6, 1e6),
                              rng.lognormvariate(0, 30)]) for _ in range(nrows)], dtype=np.float64)
    i32 = np.array([rng.choice([0, -1, 2147483647, -2147483648, rng.randrange(-10**6, 10**6)]) for _ in range(nrows)],
                   dtype=np.int32)
    strs = [bytes(rng.randrange(33, 127) for _ in range(rng.randrange(0, 12))) for _ in range(nrows)]
    blob = np.frombuffer(b"".join(strs) or b"x", dtype=np.uint8).copy()
    off = np.concatenate([[0], np.cumsum([len(x) for x in strs])]).astype(np.int64)
    spans = np.stack([off[:-1], off[1:]], axis=1).astype(np.int64).copy()
    lens = [rng.randrange(0, 5) for _ in range(nrows)]
    lvals = np.array([rng.randrange(-5000, 5000) for _ in range(sum(lens))] or [0], dtype=np.int32)
    loff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    const = np.frombuffer(b"model_name", dtype=np.uint8).copy()
    coff = np.array([0, const.size], dtype=np.int64)
    cols = (_Col * 6)(_Col(0, blob.ctypes.data, off.ctypes.data), _Col(1, i32.ctypes.data, None),
                      _Col(2, f.ctypes.data, None), _Col(3, const.ctypes.data, coff.ctypes.data),
                      _Col(4, lvals.ctypes.data, loff.ctypes.data), _Col(5, blob.ctypes.data, spans.ctypes.data))
    skip = np.array([rng.random() < 0.2 for _ in range(nrows)], dtype=np.uint8)
    for sk in (None, skip):
        n = C.c_int64(0)
        ptr_ = L.sk_fmt_rows(nrows, 6, C.cast(cols, C.c_void_p), None if sk is None else sk.ctypes.data,
                             rng.choice((1, 3, 16)), C.byref(n))
        assert ptr_, "sk_fmt_rows failed"
        text = C.string_at(ptr_, n.value).decode("latin-1")
        L.sk_fmt_free(ptr_)
        want = []
        for r in range(nrows):
            if sk is not None and sk[r]:
                continue
            s_ = strs[r].decode("latin-1")
            want.append("\t".join([s_, str(int(i32[r])), repr(float(f[r])), "model_name",
                                   ",".join(str(int(v)) for v in lvals[loff[r]:loff[r + 1]]), s_]))
        assert text == "".join(w + "\n" for w in want), "formatter output differs from Python's"


def fmt_and_ndtr(L, rng):
    for nrows in (0, 1, 7, 300):
        fmt_case(L, rng, nrows)
    z = np.array([0.0, -0.0, 1.0, -1.0, 37.0, -37.0, 1e308, -1e308, np.inf, -np.inf, np.nan, 5e-324] +
                 [rng.uniform(-10, 10) for _ in range(500)])
    out = np.empty_like(z)
    L.sk_ndtr(p(z), p(out), z.size)
    assert np.all((out[np.isfinite(z)] >= 0) & (out[np.isfinite(z)] <= 1))


def oracle_leg(rng):
    sys.path.insert(0, ROOT)
    from oracle import oracle as ora
    ora._SO = os.path.join(ROOT, "oracle", "libsk_oracle_asan.so")      # the wrappers bind the sanitizer build
    ora._lib = None
    done = 0
    for n in (0, 1, 2, 7, 8, 9, 127, 128, 129, 1000, 8192, 8193, 20001):
        x = np.array([rng.gauss(500, 80) for _ in range(n)])
        if n:
            ora.np_sum(x), ora.mean(x), ora.std(x), ora.median(x)
        f = ora.scale_outliers(x, 0, 900)
        if f.size:
            ora.get_segs(f)
            ora.medmad(f)
            ora.zscale(f)
        done += 1
    motif = np.array([rng.gauss(0, 1) for _ in range(37)])
    for n in (1, 2, 36, 37, 38, 500):
        y = np.array([rng.gauss(0, 1) for _ in range(n)])
        ora.dtw_subsequence(motif, y, want_cost=True)
        ora.dtw_subsequence_fwd(motif, y)
        ora.dtw_subsequence_path(motif, y)
        done += 1
    sig = np.array([[rng.randrange(-5, 1200) for _ in range(300)] for _ in range(12)], dtype=np.int16)
    lens = np.array([0, 1, 2, 299, 300] + [rng.randrange(0, 301) for _ in range(7)], dtype=np.int32)
    ora.motifseq_batch_i16(sig, lens, motif)
    ora.motifseq_batch_i16(sig, lens, motif, scale_mode=1)
    ora.segment_batch_i16(sig, lens)
    return done + 3


def main():
    global _ROUNDS
    if len(sys.argv) > 1:
        _ROUNDS = int(sys.argv[1])
    rng = random.Random(20260928)
    L = load()
    ntsv = nb5 = 0
    for data in tsv_corpus(rng):
        for start_col in (0, 2, 4, 8, 1000):
            tsv_case(L, data, start_col, rng.choice((1, 3, 16)))
            ntsv += 1
    try:
        example = open(os.path.join(ROOT, "tests", "golden", "example_0.blow5"), "rb").read()
    except OSError:
        example = None
    for label, data, first, comp, lie in blow5_corpus(rng, example):
        try:
            blow5_case(L, data, first, comp, lie=lie)
        except AssertionError as e:
            print("FAILED on case %r: %r" % (label, e))
            raise
        nb5 += 1
    fmt_and_ndtr(L, rng)
    nor = oracle_leg(rng)
    print("ok: %d TSV cases, %d BLOW5 cases, %d oracle cases, no sanitizer report" % (ntsv, nb5, nor))


if __name__ == "__main__":
    main()
