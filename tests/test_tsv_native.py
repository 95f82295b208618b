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
