"""The tuning surface: every environment switch the library reads is in one table (csrc/sk_runtime.hip SK_TUNABLES +
_lib.PY_TUNABLES).  Each one is flipped ALONE, to every value the table lists, on a ragged 3 000-read batch through
both tools' batch calls (int16 and float64) and the two dRNA branches -- the records must not change by a byte -- and none of them is read
unless SK_TUNING=1 is set as well."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_getenv_of_the_native_code_goes_through_the_table():
    """Source check (no GPU): the only getenv() calls under csrc/ are the table's own two in sk_runtime.hip, and every
    sk_tune("SK_...") names a row of the table."""
    csrc = os.path.join(ROOT, "squigglekit_amd", "csrc")
    table = open(os.path.join(csrc, "sk_runtime.hip")).read()
    rows = set(re.findall(r'\{"(SK_[A-Z0-9_]+)",', table))
    assert len(rows) >= 20
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hip", ".cpp", ".h")):
            continue
        src = open(os.path.join(csrc, fn)).read()
        src = re.sub(r"//[^\n]*", "", src)
        raw = re.findall(r"\bgetenv\s*\(([^)]*)\)", src)
        if fn == "sk_runtime.hip":
            assert sorted(raw) == ['"SK_TUNING"', "name"], raw
        else:
            assert not raw, (fn, raw)
        for name in re.findall(r'sk_tune\("([A-Z0-9_]+)"\)', src):
            assert name in rows, (fn, name)
    # the Python side: tuning switches go through _lib.tune(), which knows only PY_TUNABLES
    from squigglekit_amd import _lib
    for fn in ("fastio.py",):
        src = open(os.path.join(ROOT, "squigglekit_amd", fn)).read()
        for name in re.findall(r'tune\("([A-Z0-9_]+)"', src):
            assert name in _lib.PY_TUNABLES, (fn, name)
        assert not re.findall(r'environ[^\n]*"SK_(?:BLOW5|I16)', src)


def _batch():
    from squigglekit_amd import synth
    motif = synth.synthetic_motif(200)
    R, M = 3000, 4000
    sig = synth.squiggle_batch(R, M, 31337, motif=motif)
    rng = np.random.default_rng(5)
    lens = rng.integers(1200, M + 1, R).astype(np.int32)
    lens[:8] = [0, 1, 2, 63, 64, 65, 500, M]
    pa = [np.round((sig[r, :lens[r]].astype(np.int64) + 16.0) * (1493.94 / 8192.0), 2) for r in range(0, R, 10)]
    return sig, lens, motif, pa


_DRNA = []


def _records(sig, lens, motif, pa):
    from squigglekit_amd import api, synth
    if not _DRNA:
        _DRNA.extend(synth.drna_reads(12, 5, min_len=5000, max_len=20000))
    drna = [repr(api.drna_segment_reads(_DRNA)).encode(), repr(api.drna_roll_reads(_DRNA)).encode()]
    hits = api.motifseq_batch(sig, lens, motif, scale="medmad")
    hz = api.motifseq_batch(sig[:600], lens[:600], motif, scale="zscale")
    segs, nsegs = api.segment_batch(sig, np.maximum(lens - 1, 0))
    sf = api.segment_reads_f64(pa)
    hf = api.motifseq_reads_f64(pa, motif, scale="medmad")
    return [hits.tobytes(), hz.tobytes(), segs.tobytes(), nsegs.tobytes(), repr(sf).encode(), hf.tobytes()] + drna


@pytest.mark.gpu
def test_every_tuning_switch_alone_keeps_the_records(gpu, monkeypatch):
    tun = gpu.tunables()
    assert len(tun) >= 25
    for name in tun:
        monkeypatch.delenv(name, raising=False)
    data = _batch()
    base = _records(*data)
    flipped = 0
    for name, (vals, _what) in sorted(tun.items()):
        if name in gpu.PY_TUNABLES:
            continue                                       # host readers: test_fastio_switches_keep_the_rows
        for v in vals.split():
            monkeypatch.setenv(name, v)
            got = _records(*data)
            monkeypatch.delenv(name)
            assert got == base, "%s=%s changes the records" % (name, v)
            flipped += 1
    assert flipped >= 30


@pytest.mark.gpu
def test_tuning_switches_are_ignored_without_sk_tuning(gpu, monkeypatch):
    from squigglekit_amd import api
    sig, lens, motif, _ = _batch()
    launches = C.c_int32()
    monkeypatch.setenv("SK_DTW_SCHEME", "full")
    monkeypatch.delenv("SK_TUNING")
    api.motifseq_batch(sig, lens, motif)
    gpu.load().sk_last_dtw_profile(None, C.byref(launches), None, None, None)
    assert launches.value >= 1, "SK_DTW_SCHEME was read although SK_TUNING is not set"
    monkeypatch.setenv("SK_TUNING", "1")
    api.motifseq_batch(sig, lens, motif)
    gpu.load().sk_last_dtw_profile(None, C.byref(launches), None, None, None)
    assert launches.value == 0, "with SK_TUNING=1 the switch must take the exact single pass"


def test_fastio_switches_keep_the_rows(tmp_path, monkeypatch):
    """The host readers' switches (block sizes, pinning, page zapping): same rows, same order (no GPU needed: without a
    bound device the pinned variants fall back to ordinary memory)."""
    from squigglekit_amd import _lib, fastio, synth
    sig = synth.squiggle_batch(700, 1000, 99)
    npy = str(tmp_path / "r.npy")
    np.save(npy, sig)
    b5 = str(tmp_path / "r.blow5")
    fastio.write_blow5(b5, sig)

    def rows():
        a = np.concatenate([np.array(blk[1][:, :1000]) for blk in fastio.iter_npy_blocks_i16(npy)])
        b = np.concatenate([np.array(blk.rows[:, :1000]) for blk in fastio.iter_blow5_blocks_i16(b5)])
        return a.tobytes(), b.tobytes()
    base = rows()
    assert base[0] == sig.tobytes() and base[1] == sig.tobytes()
    for name, (vals, _what) in _lib.PY_TUNABLES.items():
        for v in vals.split():
            monkeypatch.setenv(name, v)
            assert rows() == base, (name, v)
            monkeypatch.delenv(name)
