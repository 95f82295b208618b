"""SURVEY section 5's sanitizer leg: the native host parsers (csrc/sk_tsv.cpp, csrc/sk_io.cpp) and the oracle, built with
AddressSanitizer + UndefinedBehaviorSanitizer (`make -C squigglekit_amd/csrc asan`), fed a corpus of malformed TSV and
BLOW5 input plus seeded mutations of the reference's example BLOW5 file (tools/fuzz_host_parsers.py).  No GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    try:
        p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    except OSError:
        return None
    return p if p and os.path.isabs(p) and os.path.exists(p) else None


def test_host_parsers_and_oracle_under_asan_ubsan():
    asan = _libasan()
    if asan is None:
        pytest.skip("no libasan in this image")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "squigglekit_amd", "csrc"), "-s", "asan"])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_host_parsers.py"), "150"], env=env,
                       capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail
    assert r.stdout.startswith("ok:"), tail


def test_blow5_decoder_rejects_offsets_outside_the_buffer():
    """sk_blow5_rows_i16 takes the buffer length: stale or foreign record offsets give flag 2, not a wild read; both
    index calls refuse a file cut inside a size field or with junk behind the last record."""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_host_parsers as fz
    from squigglekit_amd import _lib
    L = _lib.load()
    data, first = fz.blow5_file([fz.record("r%d" % i, list(range(300, 340))) for i in range(3)])
    buf = np.frombuffer(data, dtype=np.uint8).copy()
    off, size = np.zeros(8, dtype=np.int64), np.zeros(8, dtype=np.int64)
    assert L.sk_blow5_index(buf.ctypes.data, buf.size, first, off.ctypes.data, size.ctypes.data, 8) == 3
    for bad in (data[:-6], data[:first + 5], data + b"xx", data[:-5] + b"\x01\x00\x00"):
        b = np.frombuffer(bad, dtype=np.uint8).copy()
        assert L.sk_blow5_index(b.ctypes.data, b.size, first, None, None, 0) < 0
        nxt = C.c_int64()
        assert L.sk_blow5_index_some(b.ctypes.data, b.size, first, 100, off.ctypes.data, size.ctypes.data, C.byref(nxt)) < 0
    assert L.sk_blow5_index(buf.ctypes.data, buf.size, first, off.ctypes.data, size.ctypes.data, 8) == 3
    off[1] = buf.size - 4                      # runs past the end
    off[2] = -7
    rows = np.empty((3, 64), dtype=np.int16)
    nsamp, flags = np.zeros(3, dtype=np.int32), np.zeros(3, dtype=np.int32)
    ids = np.zeros(3, dtype="S8")
    _lib.check(L.sk_blow5_rows_i16(buf.ctypes.data, buf.size, off.ctypes.data, size.ctypes.data, 3, 0, 64,
                                   rows.ctypes.data, nsamp.ctypes.data, ids.ctypes.data, 8, None, flags.ctypes.data, 1))
    assert flags.tolist() == [0, 2, 2] and nsamp.tolist() == [40, 0, 0] and ids[0] == b"r0"
    assert rows[0, :40].tolist() == list(range(300, 340))


def test_blow5_reader_widens_the_id_column(tmp_path):
    """A read id longer than the id column (64) is decoded again with a wider one, not cut (flag 4)."""
    import numpy as np
    from squigglekit_amd import fastio
    sig = np.arange(600, dtype=np.int16).reshape(3, 200) + 400
    ids = ["short", "x" * 200, "y" * 65]
    path = str(tmp_path / "long_ids.blow5")
    fastio.write_blow5(path, sig, read_ids=ids)
    got = []
    for blk in fastio.iter_blow5_blocks_i16(path):
        assert not np.any(blk.flags & 4)
        got += [i.decode() for i in blk.ids]
        assert np.array_equal(np.asarray(blk.rows)[:, :200], sig)
    assert got == ids
