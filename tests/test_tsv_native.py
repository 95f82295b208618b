"""CPU: the native TSV tokenizer (csrc/sk_tsv.cpp) converts exactly like the reference's
split + int()/float() loops (segmenter.py:192-201, MotifSeq.py:265-270), flags what it must not
touch, and streams files in order across chunk boundaries."""
import gzip
import random

import numpy as np
import pytest


def write(path, lines, gz=False):
    data = "".join(lines)
    if gz:
        with gzip.open(path, "wt") as fh:
            fh.write(data)
    else:
        with open(path, "w") as fh:
            fh.write(data)


def test_values_match_python_float_exactly(tmp_path):
    from squigglekit_amd import tsvio
    rng = random.Random(5)
    toks = []
    for _ in range(20000):
        kind = rng.randrange(6)
        if kind == 0:
            toks.append(str(rng.randrange(-40000, 40000)))
        elif kind == 1:
            toks.append("%.2f" % rng.uniform(0, 200))
        elif kind == 2:
            toks.append(repr(rng.uniform(-1e3, 1e3)))                  # 17 significant digits
        elif kind == 3:
            toks.append("%.6e" % rng.uniform(-1e30, 1e30))
        elif kind == 4:
            toks.append("0.%s%d" % ("0" * rng.randrange(0, 30), rng.randrange(1, 10 ** 9)))
        else:
            toks.append(str(rng.randrange(10 ** 15, 10 ** 19)) + "." + str(rng.randrange(10 ** 6)))
    toks += ["0", "-0", "+7", "1e22", "1e23", "123456789012345678901234567890", "4.9e-324", "1.7976931348623157e308",
             "00012", "5.", ".5", "1E5", "9007199254740993"]
    line = "\t".join(["f.fast5", "rid", "a", "b"] + toks) + "\n"
    p = tmp_path / "x.tsv"
    write(p, [line])
    rows = list(tsvio.iter_tsv_native(str(p), 4))
    assert len(rows) == 1
    name, rid, vals, fl, raw = rows[0]
    assert (name, rid) == ("f.fast5", "rid") and not (fl & 8)
    want = np.array([float(t) for t in toks])
    assert vals.size == want.size
    bad = np.nonzero(vals.view(np.uint64) != want.view(np.uint64))[0]
    assert bad.size == 0, [(toks[i], vals[i], want[i]) for i in bad[:5]]


def test_flags_and_fallback_lines(tmp_path):
    from squigglekit_amd import tsvio
    lines = ["a\tb\tc\td\t1\t2\t3\n",                 # ALLINT | ANY
             "a\tb\tc\td\t1.5\t2\t3\n",               # FIRSTDOT | ANY
             "a\tb\tc\td\t0\t0\t0\n",                 # ALLINT, not ANY
             "a\tb\tc\td\t1\t2.5\t3\n",               # int line with a float token: neither FIRSTDOT nor ALLINT
             "a\tb\tc\td\t1\tnan\t3\n",               # SLOW
             "a\tb\tc\td\t1\t2\t3\r\n",               # SLOW (\r glued to the last token)
             "a\tb\n",                                # SHORT
             "\n",                                    # SHORT
             "a\tb\tc\td\t7"]                         # last line without newline
    p = tmp_path / "f.tsv"
    write(p, lines)
    rows = list(tsvio.iter_tsv_native(str(p), 4))
    fl = [r[3] for r in rows]
    assert len(rows) == 9
    assert fl[0] & 1 and fl[0] & 2 and not fl[0] & 4
    assert fl[1] & 4 and not fl[1] & 1
    assert fl[2] & 1 and not fl[2] & 2
    assert not fl[3] & 5 and rows[3][4] is not None
    assert fl[4] & 8 and fl[5] & 8 and rows[4][4] == lines[4].rstrip("\n").encode()
    assert fl[6] & 16 and fl[7] & 16
    assert rows[8][2].tolist() == [7.0]
    assert rows[0][2].tolist() == [1.0, 2.0, 3.0] and rows[1][2].tolist() == [1.5, 2.0, 3.0]


@pytest.mark.parametrize("gz", [False, True])
def test_streaming_order_across_chunks(tmp_path, gz):
    from squigglekit_amd import tsvio
    rng = np.random.default_rng(1)
    lines, want = [], []
    for i in range(300):
        n = int(rng.integers(0, 400))
        v = rng.integers(0, 1000, n)
        want.append(v)
        lines.append("\t".join(["r%d" % i, "id%d" % i] + ["x"] * 6 + [str(int(t)) for t in v]) + "\n")
    p = tmp_path / ("s.tsv.gz" if gz else "s.tsv")
    write(p, lines, gz)
    got = list(tsvio.iter_tsv_native(str(p), 8, chunk_bytes=5000, nthreads=3))
    assert len(got) == 300
    for i, (name, rid, vals, fl, raw) in enumerate(got):
        assert (name, rid) == ("r%d" % i, "id%d" % i)
        assert np.array_equal(vals, want[i].astype(float))


def test_int16_block_parser_agrees_with_python(tmp_path):
    """sk_tsv_parse_i16: integer lines land in int16 rows exactly as int() would give them; every line it does not
    take (decimals, signs with spaces, underscores, empty tokens, CR, values outside int16, too few columns) is
    flagged for the general path; names / ids / raw lines come back intact."""
    from squigglekit_amd import tsvio
    rng = np.random.default_rng(31)
    lines, kinds = [], []
    for i in range(400):
        n = int(rng.integers(0, 60))
        vals = [str(int(v)) for v in rng.integers(-300, 1300, n)]
        kind = int(rng.integers(0, 12))
        if kind == 0 and n:
            vals[int(rng.integers(n))] = "%.2f" % rng.normal(90, 10)          # a decimal
        elif kind == 1 and n:
            vals[int(rng.integers(n))] = ["5_0", " 7", "", "0x1f", "1e3", "+-3", "nan"][int(rng.integers(7))]
        elif kind == 2 and n:
            vals[int(rng.integers(n))] = str(int(rng.choice([32768, -32769, 99999999999])))
        elif kind == 3:
            vals = ["0"] * n                                                   # all zero
        elif kind == 4 and n:
            vals[0] = "+" + vals[0].lstrip("-")
            vals[-1] = "-0" if n > 1 else vals[-1]
        elif kind == 5 and n:
            vals[-1] = vals[-1] + "\r"
        elif kind == 6:
            vals = vals[:0]                                                    # no data column at all
        elif kind == 7 and n:                                                  # five digits and more: around int16's edge
            for j in rng.integers(0, n, 3):
                vals[int(j)] = str(int(rng.choice([9999, 10000, 32767, 32768, 65535, 99999, 100000, 1234567, 12345678])))
        elif kind == 8 and n:                                                  # leading zeros, widths 1 .. 9
            for j in rng.integers(0, n, 3):
                vals[int(j)] = "0" * int(rng.integers(1, 6)) + str(int(rng.integers(0, 2000)))
        head = ["f%d.fast5" % i, "id-%d" % i, "a", "b"]
        if kind == 6 and rng.random() < 0.5:
            head = head[:int(rng.integers(1, 4))]
        lines.append("\t".join(head + vals))
        kinds.append(kind)
    p = tmp_path / "mix.tsv"
    p.write_text("\n".join(lines) + ("\n" if rng.random() < 0.5 else ""))
    seen = 0
    for blk in tsvio.iter_tsv_blocks_i16(str(p), 4, chunk_bytes=4096, nthreads=3):
        for i in range(blk.n):
            line = lines[seen]
            cols = line.split("\t")
            assert blk.line(i).decode() == line
            assert blk.name(i) == cols[0]
            if len(cols) > 1:
                assert blk.read_id(i) == cols[1]
            data = cols[4:]
            fl = int(blk.flags[i])
            plain = all(t.lstrip("+-").isdigit() and t.isascii() and t.count("+") + t.count("-") <= 1
                        and t[-1].isdigit() and (t[0].isdigit() or t[0] in "+-") for t in data)
            fits = plain and all(-32768 <= int(t) <= 32767 for t in data)
            if len(cols) <= 4:
                assert fl & 16 and not (fl & 1)
            elif fits:
                assert (fl & 25) == 1, (line, fl)
                assert blk.nsamp[i] == len(data)
                assert blk.rows[i, :len(data)].tolist() == [int(t) for t in data]
                assert bool(fl & 2) == any(int(t) != 0 for t in data)
            else:
                assert (fl & 9) == 8, (line, fl)
            seen += 1
    assert seen == len(lines)


def test_non_regular_input_is_streamed_not_dropped(tmp_path):
    """`-s` pointing at a FIFO (what `-s <(zcat x.gz)` or /dev/stdin give): no size, no memory map -- the lines
    must still arrive, in order, through the buffered-read path (round 2 silently yielded nothing)."""
    import os
    import threading
    from squigglekit_amd import tsvio
    lines = ["\t".join(["f%d" % i, "id%d" % i, "a", "b"] + [str((i * 7 + k) % 900) for k in range(50 + i % 13)]) + "\n"
             for i in range(500)]
    fifo = tmp_path / "pipe.tsv"
    os.mkfifo(fifo)

    def feed():
        with open(fifo, "wb") as fh:
            for ln in lines:
                fh.write(ln.encode())
    t = threading.Thread(target=feed)
    t.start()
    seen = 0
    for blk in tsvio.iter_tsv_blocks_i16(str(fifo), 4, chunk_bytes=3000, nthreads=2):
        for i in range(blk.n):
            cols = lines[seen].rstrip("\n").split("\t")
            assert blk.name(i) == cols[0] and (int(blk.flags[i]) & 25) == 1
            assert blk.rows[i, :blk.nsamp[i]].tolist() == [int(v) for v in cols[4:]]
            seen += 1
    t.join()
    assert seen == len(lines)
    # the float tokenizer's entry goes through the same block reader
    t = threading.Thread(target=feed)
    t.start()
    got = list(tsvio.iter_tsv_native(str(fifo), 4, chunk_bytes=3000, nthreads=2))
    t.join()
    assert len(got) == len(lines) and got[-1][0] == "f499"


def test_centi_tokenizer_is_float_exact_and_steps_aside(monkeypatch):
    """sk_tsv_parse_centi (round 6): decimal tokens with at most two decimals -- what SquigglePull writes, np.round(pA, 2)
    (SquigglePull.py:183-189,222) -- as int32 centi-units; c / 100.0 must be float(token) bit for bit
    (segmenter.py:198-199, MotifSeq.py:270), the flags must be the float64 tokenizer's, and a chunk holding anything else
    (a third decimal, an exponent, a negative zero, junk, too many digits) must go through the float64 tokenizer."""
    from squigglekit_amd import tsvio
    rng = random.Random(11)

    def centi_token():
        k = rng.randrange(7)
        v = rng.randrange(0, 200000)
        if k == 0:
            return repr(v / 100.0)                                       # shortest repr: 96.5, 103.25, 88.0
        if k == 1:
            return "%.2f" % (v / 100.0)
        if k == 2:
            return str(v // 100)                                         # an integer token inside a decimal line
        if k == 3:
            return "-%d.%02d" % (rng.randrange(1, 3000), rng.randrange(100))
        if k == 4:
            return "%d.%d000" % (v // 100, rng.randrange(10))            # further decimals, all zero
        if k == 5:
            return rng.choice(["5.", ".5", "+7.25", "0.0", "0", "000.10", "9999999.99", "-.01"])
        return "%d.%d" % (rng.randrange(10 ** 6), rng.randrange(10))

    lines, toks = [], []
    for i in range(300):
        t = [centi_token() for _ in range(rng.randrange(1, 60))]
        if i % 3 == 0:
            t[0] = "%.2f" % rng.uniform(1, 200)                           # FIRSTDOT lines and others
        toks.append(t)
        lines.append("\t".join(["f%d.fast5" % i, "rid%d" % i, "a", "b"] + t) + "\n")
    buf = "".join(lines).encode()
    chunk = (buf, 0, len(buf))
    fb = tsvio.parse_block_float(chunk, 4, 4)
    assert fb.centi is not None and fb.centi.dtype == np.int32 and fb.batch_values() is fb.centi
    want = np.array([float(x) for t in toks for x in t])
    got = fb.values
    assert got.dtype == np.float64 and got.size == want.size
    assert not np.any(got.view(np.uint64) != want.view(np.uint64))
    assert [fb.text("name", i) for i in (0, 299)] == ["f0.fast5", "f299.fast5"] and fb.text("id", 7) == "rid7"
    monkeypatch.setenv("SK_TSV_NO_CENTI", "1")
    ref = tsvio.parse_block_float(chunk, 4, 4)
    monkeypatch.delenv("SK_TSV_NO_CENTI")
    assert ref.centi is None and np.array_equal(ref.flags, fb.flags) and np.array_equal(ref.off, fb.off)
    assert np.array_equal(ref.values.view(np.uint64), got.view(np.uint64))
    assert np.array_equal(ref.name_off, fb.name_off) and np.array_equal(ref.id_len, fb.id_len)
    # one odd token anywhere and the whole chunk is the float64 tokenizer's
    for odd in ("1.234", "1e2", "-0.0", "-0", "nan", "12x", "", "12345678.5", " 3.5", "1_0.5", "3.14159"):
        lines2 = list(lines)
        lines2[150] = "\t".join(["g.fast5", "r", "a", "b", "1.5", odd, "2.25"]) + "\n"
        b2 = "".join(lines2).encode()
        f2 = tsvio.parse_block_float((b2, 0, len(b2)), 4, 3)
        assert f2.centi is None, odd
        monkeypatch.setenv("SK_TSV_NO_CENTI", "1")
        r2 = tsvio.parse_block_float((b2, 0, len(b2)), 4, 3)
        monkeypatch.delenv("SK_TSV_NO_CENTI")
        assert np.array_equal(r2.flags, f2.flags) and np.array_equal(r2.values.view(np.uint64), f2.values.view(np.uint64)), odd
    # lines with too few columns, and a last line without its newline
    b3 = b"a\tb\tc\n" + b"a\tb\tc\td\t1.25\t2.5\n" + b"a\tb\tc\td\t\n" + b"x\ty\tz\tw\t7.75"
    f3 = tsvio.parse_block_float((b3, 0, len(b3)), 4, 2)
    monkeypatch.setenv("SK_TSV_NO_CENTI", "1")
    r3 = tsvio.parse_block_float((b3, 0, len(b3)), 4, 2)
    monkeypatch.delenv("SK_TSV_NO_CENTI")
    assert np.array_equal(r3.flags, f3.flags) and np.array_equal(r3.values.view(np.uint64), f3.values.view(np.uint64))
