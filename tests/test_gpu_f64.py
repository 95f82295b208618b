"""GPU parity: the float64 sample path (pA TSVs, segmenter.py:198-199 / MotifSeq.py:270):
radix-select median / MAD, numpy-order mean/std on doubles, same state machine and DTW."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _pa_reads(n, m, seed):
    """pA-like signals: two decimals, ~N(96, 15), a stall plateau, a few spikes."""
    from squigglekit_amd import synth
    raw = synth.squiggle_batch(n, m, seed).astype(np.int64)
    return [np.round((raw[r] + 16.0) * (1493.94 / 8192.0), 2) for r in range(n)]


def test_segmenter_f64_real_read_golden(gpu, example_read):
    from squigglekit_amd import api
    from squigglekit_amd.blow5 import to_pA
    rec = example_read
    pa = to_pA(rec["signal"], rec["digitisation"], rec["offset"], rec["range"])
    want = [g for g in load_golden("segmenter_get_segs.json.gz")["real_read"] if g["kind"] == "pA"][0]
    assert api.segment_reads_f64([pa[:-1]])[0] == want["segs"]


def test_segmenter_f64_vs_oracle(gpu, ora):
    from squigglekit_amd import api
    from squigglekit_amd._lib import SegParams
    rng = np.random.default_rng(1)
    reads = _pa_reads(48, 5000, 2024)
    reads = [r[:int(rng.integers(1, 5001))] for r in reads]
    reads[3] = np.zeros(100)                   # nothing survives
    reads[4] = np.full(300, 95.5)              # std == 0
    reads[5] = reads[5][:1]
    reads.append(np.round(rng.normal(96, 15, 20001), 2))     # > 2 numpy chunks
    for kw in (dict(lim_low=0, lim_hi=900), dict(lim_low=60, lim_hi=130, error=9, corrector=2, window=40)):
        p = SegParams(**kw)
        got = api.segment_reads_f64(reads, p)
        op = ora.SegParams(p.error, p.corrector, p.window, p.seg_dist, p.std_scale, p.stall_len)
        for r, sig in enumerate(reads):
            f = ora.scale_outliers(sig, p.lim_low, p.lim_hi)
            assert got[r] == ora.get_segs(f, op), (kw, r)


@pytest.mark.parametrize("scale", ["medmad", "zscale"])
def test_motifseq_f64_vs_oracle(gpu, ora, example_model, scale):
    from squigglekit_amd import api
    reads = _pa_reads(40, 3000, 7)
    reads[0] = reads[0][:1]
    reads[1] = reads[1][:2]
    reads[2] = reads[2][:777]
    got = api.motifseq_reads_f64(reads, example_model, scale=scale, scale_low=0, scale_hi=1200)
    for r, sig in enumerate(reads):
        f = ora.scale_outliers(sig, 0, 1200)
        y = ora.medmad(f)[0] if scale == "medmad" else ora.zscale(f)[0]
        if not np.all(np.isfinite(y)):
            assert got["flags"][r] & 2
            continue
        d, s, e = ora.dtw_subsequence(example_model, y)
        assert (got["dist"][r], got["start"][r], got["end"][r], got["n"][r]) == (d, s, e, f.size), (scale, r)


def test_f64_and_i16_paths_agree(gpu, example_model):
    """Integer-valued data must give identical results through either kernel family."""
    from squigglekit_amd import api, synth
    sig = synth.squiggle_batch(24, 2500, 99, motif=example_model)
    a = api.motifseq_batch(sig, None, example_model)
    b = api.motifseq_reads_f64([sig[r].astype(float) for r in range(24)], example_model)
    assert np.array_equal(a, b)
    sa = api.segment_reads([sig[r] for r in range(24)])
    sb = api.segment_reads_f64([sig[r].astype(float) for r in range(24)])
    assert sa == sb


def test_f64_and_i16_paths_agree_long_motif(gpu):
    """Same through the chained row chunks of a 1 300-point motif (both sample feeds)."""
    from squigglekit_amd import api, synth
    motif = synth.synthetic_motif(1300, seed=5)
    sig = synth.squiggle_batch(12, 3000, 4321, motif=motif[:300])
    for scale in ("medmad", "zscale"):
        a = api.motifseq_batch(sig, None, motif, scale=scale)
        b = api.motifseq_reads_f64([sig[r].astype(float) for r in range(12)], motif, scale=scale)
        assert np.array_equal(a, b), scale


def test_normalise_f64_matches_reference_pA_row(gpu, ora, example_read):
    """The real read in pA: normalised signal equals the numpy/sklearn result the reference fed to DTW."""
    from squigglekit_amd import api
    from squigglekit_amd.blow5 import to_pA
    rec = example_read
    pa = to_pA(rec["signal"], rec["digitisation"], rec["offset"], rec["range"])
    gold = load_golden("motifseq_cli.json.gz")
    import hashlib
    for run in gold["runs"]:
        if run["tsv"] != "real_pA" or run["flags"][:1] != ["-l"]:
            continue
        y = api.normalise(pa, scale=run["flags"][1])
        g = run["dtw_inputs"][0]
        assert y.size == g["n"]
        assert hashlib.sha256(y.tobytes()).hexdigest() == g["sha256"], run["flags"]


def test_f64_selection_corners(gpu, ora):
    """Median / MAD selection on doubles: heavy duplicates, values that differ only in their lowest
    mantissa bits (every radix digit is needed), wide dynamic range, tiny reads, even and odd sizes."""
    from squigglekit_amd import api
    rng = np.random.default_rng(2024)
    reads = []
    for n in (1, 2, 3, 4, 9, 10, 255, 256, 257, 1025, 3000, 5001):
        reads.append(rng.choice([1.5, 2.5, 3.5, 700.25], n))                        # few distinct values
        reads.append(1.0 + rng.integers(0, 5000, n) * 2.0 ** -50)                   # differ in the last bits
        reads.append(np.exp(rng.normal(3.0, 1.5, n)).clip(1e-3, 1100.0))            # wide range
        reads.append(np.round(rng.normal(90.0, 12.0, n), 1))                        # pA-like, 0.1 steps (ties)
    for i, sig in enumerate(reads):
        f = ora.scale_outliers(sig, 0, 1200)
        for scale in ("medmad", "zscale"):
            want = ora.medmad(f)[0] if scale == "medmad" else ora.zscale(f)[0]
            got = api.normalise(sig, scale=scale)
            assert got.shape == want.shape, (i, scale)
            assert np.array_equal(got, want, equal_nan=True), (i, scale, len(sig))
    # sklearn's "mean not close to zero" re-centring fires on the near-constant reads (the oracle
    # reports it); the library applies the same correction and flags the read
    fired = [bool(ora.zscale(ora.scale_outliers(sig, 0, 1200))[3]) for sig in reads]
    assert any(fired)
    motif = np.array([0.5, -0.25, 1.0, 0.0, -1.5])
    hits = api.motifseq_reads_f64(reads, motif, scale="zscale")
    for i, sig in enumerate(reads):
        y = ora.zscale(ora.scale_outliers(sig, 0, 1200))[0]
        if y.size == 0 or not np.all(np.isfinite(y)):
            continue
        d, s0, e0 = ora.dtw_subsequence(motif, y)
        assert (hits["dist"][i], hits["start"][i], hits["end"][i]) == (d, s0, e0), i
        assert bool(hits["flags"][i] & 4) == fired[i], i
    # the same reads as one ragged batch through the segmenter statistics (median + std)
    segs = api.segment_reads_f64(reads)
    for sig, got in zip(reads, segs):
        f = ora.scale_outliers(sig, 0, 900)
        want = ora.get_segs(f) if f.size else False
        assert got == want


@pytest.mark.parametrize("scale,lanes", [("medmad", None), ("zscale", None), ("medmad", "8"), ("zscale", "64")])
def test_motifseq_f64_batch_through_the_screening_scheme(gpu, ora, example_model, scale, lanes, monkeypatch):
    """Enough float64 (pA) reads for the default scheme -- fixed-point screening, pre-roll, certified window -- on the
    normalise-on-the-fly feed, in the lane layout a batch of this size gets and in the other two (those two also with
    the window passes taking the reads sorted by need, which large chunks get by themselves)."""
    from concurrent.futures import ThreadPoolExecutor
    from squigglekit_amd import api
    if lanes:
        monkeypatch.setenv("SK_DTW_QL", lanes)
        monkeypatch.setenv("SK_DTW_SORT_MIN", "1")                   # ... and the window passes in sorted order
    reads = _pa_reads(288, 2900, 11)
    rng = np.random.default_rng(4)
    for r in range(0, 288, 9):
        reads[r] = reads[r][:int(rng.integers(1200, 2901))]
    got = api.motifseq_reads_f64(reads, example_model, scale=scale, scale_low=0, scale_hi=1200)
    launches = C.c_int32()
    gpu.load().sk_last_dtw_profile(None, C.byref(launches), None, None, None)
    assert launches.value >= 1, "the batch did not take the screening scheme"

    def one(sig):
        f = ora.scale_outliers(sig, 0, 1200)
        y = ora.medmad(f)[0] if scale == "medmad" else ora.zscale(f)[0]
        return (ora.dtw_subsequence(example_model, y) + (f.size,)) if np.all(np.isfinite(y)) else None
    with ThreadPoolExecutor(16) as ex:
        want = list(ex.map(one, reads))
    for r, w in enumerate(want):
        if w is None:
            assert got["flags"][r] & 2
        else:
            assert (got["dist"][r], got["start"][r], got["end"][r], got["n"][r]) == w, (scale, lanes, r)


def _stream_reads(rng):
    """Reads of at most 4 096 samples (the streaming float64 kernel's range) that stress its median selection and its
    certificate: ungridded doubles, heavy ties, an outlier that stretches the histogram range, all-equal and
    two-valued reads, tiny spreads, NaN / inf samples, every length around the 64-sample slots."""
    reads = []
    for n in (1, 2, 3, 63, 64, 65, 127, 128, 129, 1000, 2047, 2048, 2049, 3999, 4000, 4095, 4096):
        reads.append(rng.normal(96.0, 15.0, n))                                       # ungridded
        reads.append(np.round(rng.normal(96.0, 15.0, n), 2))                          # pA grid
        reads.append(rng.choice([80.25, 95.5, 95.51, 120.0], n))                      # four distinct values
        x = np.round(rng.normal(90.0, 6.0, n), 2)
        x[rng.integers(0, n)] = 899.99                                                # one far outlier inside the limits
        reads.append(x)
        reads.append(np.full(n, 77.77))                                               # std == 0
        reads.append(90.0 + rng.integers(0, 3, n) * 2.0 ** -40)                       # tiny spread
        y = rng.normal(96.0, 15.0, n)
        y[rng.integers(0, n, 3)] = [np.nan, np.inf, -np.inf]
        reads.append(y)
        reads.append(np.round(np.abs(rng.standard_cauchy(n)) * 40.0, 3))              # heavy tail, many dropped at 900
    return reads


def test_f64_streaming_segmenter_vs_oracle(gpu, ora, monkeypatch):
    from squigglekit_amd import api
    from squigglekit_amd._lib import SegParams
    reads = _stream_reads(np.random.default_rng(77))
    reads += _pa_reads(64, 4000, 5)
    for kw in (dict(), dict(lim_low=60, lim_hi=130, error=9, corrector=2, window=40), dict(std_scale=0.1, window=20),
               dict(lim_low=-1000, lim_hi=1000, std_scale=-0.5)):
        p = SegParams(**kw)
        op = ora.SegParams(p.error, p.corrector, p.window, p.seg_dist, p.std_scale, p.stall_len)
        want = [ora.get_segs(f, op) if f.size else False
                for f in (ora.scale_outliers(sig, p.lim_low, p.lim_hi) for sig in reads)]
        for delta in (None, "1e13"):                       # as shipped / every read through the numpy-order redo
            if delta:
                monkeypatch.setenv("SK_SEG_DELTA_SCALE", delta)
            got = api.segment_reads_f64(reads, p)
            retried = gpu.load().sk_last_f64_retries()
            monkeypatch.delenv("SK_SEG_DELTA_SCALE", raising=False)
            assert retried >= 0, "the batch did not take the streaming kernel"
            if delta:
                assert retried >= len(reads) // 2
            bad = [r for r in range(len(reads)) if got[r] != want[r]]
            assert not bad, (kw, delta, bad[:8], retried)


def test_f64_streaming_medmad_vs_oracle(gpu, ora, example_model):
    from squigglekit_amd import api
    reads = _stream_reads(np.random.default_rng(78))
    reads += _pa_reads(300, 3000, 6)                       # enough for the screening scheme
    got = api.motifseq_reads_f64(reads, example_model, scale="medmad", scale_low=0, scale_hi=900)
    assert gpu.load().sk_last_f64_retries() >= 0, "the batch did not take the streaming kernel"
    from concurrent.futures import ThreadPoolExecutor

    def one(sig):
        f = ora.scale_outliers(sig, 0, 900)
        if f.size == 0:
            return None
        y = ora.medmad(f)[0]
        return (ora.dtw_subsequence(example_model, y) + (f.size,)) if np.all(np.isfinite(y)) else (f.size,)
    with ThreadPoolExecutor(16) as ex:
        want = list(ex.map(one, reads))
    for r, w in enumerate(want):
        if w is None:
            assert got["n"][r] == 0 and got["flags"][r] & 1, r
        elif len(w) == 1:
            assert got["n"][r] == w[0] and got["flags"][r] & 2, r
        else:
            assert (got["dist"][r], got["start"][r], got["end"][r], got["n"][r]) == w, r


def _long_reads(rng):
    """Reads beyond 4 096 samples (the window-by-window float64 kernel): pA-like, ungridded (hundreds of distinct values
    in the median's bin: resolved in LDS), three distinct values (more than 512 equal-bin members), an outlier-stretched
    range, NaN inside, lengths around the 2 048-sample windows."""
    reads = []
    for n in (4097, 6143, 6144, 6145, 20000, 36977, 70001):
        reads.append(np.round(rng.normal(96.0, 15.0, n), 2))
        reads.append(rng.normal(96.0, 15.0, n))
        reads.append(rng.choice([80.25, 95.5, 95.51], n))
        x = np.round(rng.normal(90.0, 6.0, n), 2)
        x[rng.integers(0, n, 3)] = [899.99, np.nan, 0.011]
        reads.append(x)
    reads.append(90.0 + rng.integers(0, 100000, 150001) * 2.0 ** -30)           # dense: the select recurses on the members' range
    reads.append(np.full(9000, 77.77))
    return reads


def test_f64_long_reads_segmenter_vs_oracle(gpu, ora, monkeypatch):
    from squigglekit_amd import api
    from squigglekit_amd._lib import SegParams
    reads = _long_reads(np.random.default_rng(311))
    for kw in (dict(), dict(lim_low=60, lim_hi=130, window=40), dict(std_scale=0.1, window=20)):
        p = SegParams(**kw)
        op = ora.SegParams(p.error, p.corrector, p.window, p.seg_dist, p.std_scale, p.stall_len)
        want = [ora.get_segs(f, op) if f.size else False
                for f in (ora.scale_outliers(sig, p.lim_low, p.lim_hi) for sig in reads)]
        for delta in (None, "1e13"):
            if delta:
                monkeypatch.setenv("SK_SEG_DELTA_SCALE", delta)
            got = api.segment_reads_f64(reads, p)
            retried = gpu.load().sk_last_f64_retries()
            monkeypatch.delenv("SK_SEG_DELTA_SCALE", raising=False)
            assert retried >= 0, "the batch did not take the streaming kernels"
            if not delta:
                assert retried <= 4, retried                 # (the all-equal read; nothing else should need the redo)
            bad = [r for r in range(len(reads)) if got[r] != want[r]]
            assert not bad, (kw, delta, bad[:8], retried)


def test_f64_long_reads_medmad_vs_oracle(gpu, ora, example_model):
    from concurrent.futures import ThreadPoolExecutor
    from squigglekit_amd import api
    reads = _long_reads(np.random.default_rng(312))[:24]
    got = api.motifseq_reads_f64(reads, example_model, scale="medmad", scale_low=0, scale_hi=900)
    assert 0 <= gpu.load().sk_last_f64_retries() <= 2

    def one(sig):
        f = ora.scale_outliers(sig, 0, 900)
        y = ora.medmad(f)[0]
        return (ora.dtw_subsequence(example_model, y) + (f.size,)) if np.all(np.isfinite(y)) else (f.size,)
    with ThreadPoolExecutor(16) as ex:
        want = list(ex.map(one, reads))
    for r, w in enumerate(want):
        if len(w) == 1:
            assert got["n"][r] == w[0] and got["flags"][r] & 2, r
        else:
            assert (got["dist"][r], got["start"][r], got["end"][r], got["n"][r]) == w, r


def _class_reads(rng, cap):
    """A batch whose longest read is exactly `cap` samples (the batch's longest read picks the kernel: 10 240 / 20 480 /
    41 472 = the workgroup-per-read kernel with 4 / 8 / 12 wavefronts, round 5): the kinds of _long_reads at lengths
    around the wavefronts' 64 x GNJ-sample shares, plus reads far shorter than the cap -- wavefronts with nothing to do."""
    per = cap // {10240: 4, 20480: 8, 41472: 12}[cap]
    reads = []
    for n in (cap, cap - 1, cap - 63, cap - 64, cap - 65, per, per + 1, per - 1, 2 * per + 7, 4097, 64, 3, 1, 0):
        reads.append(np.round(rng.normal(96.0, 15.0, n), 2))
    for n in (cap, cap // 2 + 13, 5000):
        reads.append(rng.normal(96.0, 15.0, n))                                  # ungridded: hundreds of distinct values in a bin
        reads.append(rng.choice([80.25, 95.5, 95.51], n))                        # more than 512 equal-bin members
        x = np.round(rng.normal(90.0, 6.0, n), 2)
        x[rng.integers(0, n, 3)] = [899.99, np.nan, 0.011]                       # an outlier-stretched range, NaN inside
        reads.append(x)
        reads.append(90.0 + rng.integers(0, 100000, n) * 2.0 ** -30)             # dense: the select recurses on the members' range
    reads.append(np.full(cap - 5, 77.77))
    x = np.round(rng.normal(96.0, 15.0, cap), 2)
    x[per - 10:per + 700] = 2000.0                                               # a whole stretch dropped across two wavefronts
    reads.append(x)
    return reads


@pytest.mark.parametrize("cap", [10240, 20480, 41472])
def test_f64_workgroup_kernel_segmenter_vs_oracle(gpu, ora, monkeypatch, cap):
    """k_f64_wg (one look, a workgroup per read) for every wavefront count, against the oracle and against the
    window-by-window kernel it replaces (SK_F64_LONG_LOOKS=1), with and without the forced numpy-order redo."""
    from squigglekit_amd import api
    from squigglekit_amd._lib import SegParams
    reads = _class_reads(np.random.default_rng(cap), cap)
    for kw in (dict(), dict(lim_low=60, lim_hi=130, window=40)):
        p = SegParams(**kw)
        op = ora.SegParams(p.error, p.corrector, p.window, p.seg_dist, p.std_scale, p.stall_len)
        want = [ora.get_segs(f, op) if f.size else False
                for f in (ora.scale_outliers(sig, p.lim_low, p.lim_hi) for sig in reads)]
        for delta in (None, "1e13"):
            if delta:
                monkeypatch.setenv("SK_SEG_DELTA_SCALE", delta)
            got = api.segment_reads_f64(reads, p)
            retried = gpu.load().sk_last_f64_retries()
            monkeypatch.setenv("SK_F64_LONG_LOOKS", "1")
            old = api.segment_reads_f64(reads, p)
            monkeypatch.delenv("SK_F64_LONG_LOOKS")
            monkeypatch.delenv("SK_SEG_DELTA_SCALE", raising=False)
            assert retried >= 0
            if not delta:
                assert retried <= 6, retried             # (the all-equal read, the tiny ones; nothing else should need the redo)
            bad = [r for r in range(len(reads)) if got[r] != want[r]]
            assert not bad, (cap, kw, delta, bad[:8], retried)
            assert got == old


def test_f64_workgroup_kernel_medmad_vs_oracle(gpu, ora, example_model, monkeypatch):
    """medmad takes the workgroup kernel for batches whose longest read has 20 481 .. 41 472 samples."""
    from concurrent.futures import ThreadPoolExecutor
    from squigglekit_amd import api
    reads = _class_reads(np.random.default_rng(77), 41472)
    got = api.motifseq_reads_f64(reads, example_model, scale="medmad", scale_low=0, scale_hi=900)
    assert 0 <= gpu.load().sk_last_f64_retries() <= 4
    monkeypatch.setenv("SK_F64_LONG_LOOKS", "1")
    old = api.motifseq_reads_f64(reads, example_model, scale="medmad", scale_low=0, scale_hi=900)
    monkeypatch.delenv("SK_F64_LONG_LOOKS")
    assert got.tobytes() == old.tobytes()

    def one(sig):
        f = ora.scale_outliers(sig, 0, 900)
        if f.size == 0:
            return None
        y = ora.medmad(f)[0]
        return (ora.dtw_subsequence(example_model, y) + (f.size,)) if np.all(np.isfinite(y)) else (f.size,)
    with ThreadPoolExecutor(16) as ex:
        want = list(ex.map(one, reads))
    for r, w in enumerate(want):
        if w is None:
            assert got["n"][r] == 0 and got["flags"][r] & 1, r
        elif len(w) == 1:
            assert got["n"][r] == w[0] and got["flags"][r] & 2, r
        else:
            assert (got["dist"][r], got["start"][r], got["end"][r], got["n"][r]) == w, r


@pytest.mark.parametrize("scale", ["medmad", "zscale"])
def test_f64_screening_of_near_constant_reads(gpu, ora, example_model, scale):
    """Reads whose spread is 1e-14 of their level, enough of them for the screening scheme: the fixed-point image of a
    sample must come from (x - centre) * (2^22 / scale) -- the constant -centre * 2^22 / scale of the int16 feed's fma
    is only good to 1e-16 of its own size, 1e5 fixed-point units here (found by the round-4 fuzz)."""
    from concurrent.futures import ThreadPoolExecutor
    from squigglekit_amd import api
    rng = np.random.default_rng(2)
    reads = [500.0 + rng.integers(0, 3, int(rng.integers(1500, 2600))) * 2.0 ** -40 for _ in range(150)]
    reads += [90.0 + rng.integers(0, 2000, int(rng.integers(1500, 2600))) * 2.0 ** -44 for _ in range(150)]
    got = api.motifseq_reads_f64(reads, example_model, scale=scale)
    launches = C.c_int32()
    gpu.load().sk_last_dtw_profile(None, C.byref(launches), None, None, None)
    assert launches.value >= 1, "the batch did not take the screening scheme"

    def one(sig):
        f = ora.scale_outliers(sig, 0, 1200)
        y = ora.medmad(f)[0] if scale == "medmad" else ora.zscale(f)[0]
        return (ora.dtw_subsequence(example_model, y) + (f.size,)) if np.all(np.isfinite(y)) else None
    with ThreadPoolExecutor(16) as ex:
        want = list(ex.map(one, reads))
    checked = 0
    for r, w in enumerate(want):
        if w is None:
            assert got["flags"][r] & 2
        else:
            checked += 1
            assert (got["dist"][r], got["start"][r], got["end"][r], got["n"][r]) == w, (scale, r)
    assert checked >= 150


def test_segment_ragged_f64_with_per_read_cuts(gpu, ora):
    """sk_segment_batch_f64_len: read r is the first len[r] samples of its slot (the tools' sig[:Num] cut on a parsed TSV
    chunk, nothing repacked) -- through the 4 096-sample kernel, the window-by-window one and the numpy-order one."""
    from squigglekit_amd import api
    rng = np.random.default_rng(9)
    for maxn in (3000, 9000):
        reads = _pa_reads(40, maxn, 3)
        reads = [r[:int(rng.integers(2, maxn + 1))] for r in reads]
        flat = np.concatenate(reads)
        off = np.concatenate([[0], np.cumsum([r.size for r in reads])]).astype(np.int64)
        for cut in ("all", "minus1", "random"):
            if cut == "all":
                lens = None
            elif cut == "minus1":
                lens = np.array([max(r.size - 1, 0) for r in reads], dtype=np.int32)
            else:
                lens = np.array([int(rng.integers(0, r.size + 1)) for r in reads], dtype=np.int32)
            segs, nsegs = api.segment_ragged_f64(flat, off, lens)
            for r, sig in enumerate(reads):
                x = sig if lens is None else sig[:lens[r]]
                f = ora.scale_outliers(x, 0, 900)
                want = (ora.get_segs(f) if f.size else False) or []
                assert segs[r, :nsegs[r]].tolist() == want, (maxn, cut, r)
    bad = np.array([5, 99999], dtype=np.int32)
    with pytest.raises(Exception):
        api.segment_ragged_f64(np.zeros(10), np.array([0, 4, 10], dtype=np.int64), bad)


def test_multi_motif_ragged_f64_equals_one_call_per_motif(gpu, example_model):
    """sk_motifseq_multi_batch_f64 (the `for name in m_order` loop of MotifSeq.py:436 on pA input: the batch staged and
    filtered once, every motif against it on the device) == one sk_motifseq_batch_f64 call per motif, medmad and zscale,
    with reads the filter drops samples of, an empty read and enough reads for the screening scheme."""
    from squigglekit_amd import api
    rng = np.random.default_rng(5)
    reads = _pa_reads(330, 3000, 21)
    reads[3] = np.zeros(0)
    reads[4][::7] = 1500.0                                   # dropped by the limits
    flat = np.concatenate(reads)
    off = np.concatenate([[0], np.cumsum([r.size for r in reads])]).astype(np.int64)
    motifs = [example_model, example_model[10:90], rng.normal(0.0, 1.0, 33)]
    for scale in ("medmad", "zscale"):
        got = api.motifseq_multi_ragged_f64(flat, off, motifs, scale=scale)
        for k, m in enumerate(motifs):
            want = api.motifseq_reads_f64(reads, m, scale=scale)
            assert got[k].tobytes() == want.tobytes(), (scale, k)
