"""GPU parity: segmenter path (scale_outliers -> median/std thresholds -> get_segs
state machine) vs goldens minted from the reference and vs the oracle.
Bar: bit-exact segment boundaries."""
import types

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["wave", "lane"])
def walk_shape(request, monkeypatch):
    """Every test of this module twice: batches this small take the wavefront-per-read walk (k_seg_walkL, round 6) by
    default; SK_WALK_NOWAVE=1 keeps the lane-per-read walks (k_seg_walk4 / 3 / 2) that large batches of short reads take."""
    if request.param == "lane":
        monkeypatch.setenv("SK_WALK_NOWAVE", "1")
    else:
        monkeypatch.delenv("SK_WALK_NOWAVE", raising=False)
    return request.param


def _params(kw):
    from squigglekit_amd._lib import SegParams
    return SegParams(**kw)


def test_kats_from_reference(gpu):
    """Hand-checkable known-answer tests produced by /root/reference/segmenter.py."""
    from squigglekit_amd import api
    gold = load_golden("segmenter_get_segs.json.gz")
    for k in gold["kats"]:
        res = api.segment_reads([np.array(k["sig"], dtype=np.int16)], _params(k["params"]))[0]
        assert res == k["segs"], (k["name"], res, k["segs"])


def test_synthetic_reads_match_reference(gpu):
    from squigglekit_amd import api, synth
    import hashlib
    gold = load_golden("segmenter_get_segs.json.gz")["synthetic"]
    sig = synth.squiggle_batch(gold["reads"], gold["samples"], gold["seed"])
    assert hashlib.sha256(sig.tobytes()).hexdigest() == gold["sha256"], "synthetic generator drifted"
    lens = np.full(sig.shape[0], sig.shape[1] - 1, dtype=np.int32)      # Num=-1 drops the last sample
    for run in gold["runs"]:
        segs, nsegs = api.segment_batch(sig, lens, _params(run["params"]))
        for r in range(sig.shape[0]):
            got = segs[r, :nsegs[r]].tolist()
            assert got == run["segs"][r], (run["params"], r, got, run["segs"][r])


def test_real_read_golden(gpu, example_read):
    from squigglekit_amd import api
    gold = load_golden("segmenter_get_segs.json.gz")["real_read"]
    raw = example_read["signal"]
    want = [g for g in gold if g["kind"] == "raw"][0]
    res = api.segment_reads([raw[:-1]])[0]
    assert res == want["segs"]


def test_get_segs_mirror(gpu, ora):
    """api.get_segs(sig, args) has the reference's call shape (already-filtered sig)."""
    from squigglekit_amd import api, synth
    sig = synth.squiggle_batch(4, 3000, 99)
    args = types.SimpleNamespace(error=5, corrector=50, window=150, seg_dist=50, std_scale=0.75, stall_len=0.25)
    for r in range(4):
        f = sig[r][(sig[r] > 0) & (sig[r] < 900)]
        assert api.get_segs(f, args) == ora.get_segs(f)


def test_vs_oracle_random_params_and_lengths(gpu, ora):
    """Oracle comparison over ragged lengths and parameter corners (incl. live corrector)."""
    from squigglekit_amd import api, synth
    rng = np.random.default_rng(4242)
    sig = synth.squiggle_batch(96, 6000, 31337)
    lens = rng.integers(1, 6001, size=96).astype(np.int32)
    lens[:6] = [1, 2, 63, 64, 65, 6000]
    sig[7, :] = 0                       # empty after filter
    sig[8, :] = 500                     # std == 0 -> empty band
    variants = [dict(), dict(error=10, corrector=0), dict(error=12, corrector=3, window=30),
                dict(window=10, seg_dist=0, stall_len=0.0), dict(std_scale=3.0),
                dict(error=0), dict(window=1, error=0, seg_dist=1000),
                dict(lim_low=400, lim_hi=600), dict(stall_len=1.5)]
    for kw in variants:
        p = _params(kw)
        segs, nsegs = api.segment_batch(sig, lens, p, max_segs=16)
        okw = {k: v for k, v in kw.items() if k not in ("lim_low", "lim_hi")}
        osegs, onsegs = ora.segment_batch_i16(sig, lens, ora.SegParams(**okw), lo=p.lim_low, hi=p.lim_hi,
                                              max_segs=segs.shape[1])
        assert np.array_equal(nsegs, onsegs), (kw, np.nonzero(nsegs != onsegs)[0][:5])
        for r in range(96):
            assert np.array_equal(segs[r, :nsegs[r]], osegs[r, :nsegs[r]]), (kw, r)


def test_fast_and_general_walk_agree(gpu, ora, monkeypatch):
    """The straight-line walk (error < corrector) against the general one and the oracle, with
    ragged lengths so that the hand-over between the two step forms is exercised inside a wave."""
    from squigglekit_amd import api, synth
    rng = np.random.default_rng(99)
    sig = synth.squiggle_batch(192, 5000, 777)
    lens = rng.integers(1, 5001, size=192).astype(np.int32)
    lens[64:128] = 5000                      # one wave with no ragged tail at all
    lens[130] = 0
    for kw in (dict(), dict(error=49, corrector=50, window=20), dict(error=1, corrector=2, window=5, seg_dist=3),
               dict(error=3, window=2, stall_len=0.5, seg_dist=0), dict(error=-1), dict(std_scale=0.1, window=3)):
        p = _params(kw)
        osegs = onsegs = None
        for general in (False, True):
            if general:
                monkeypatch.setenv("SK_WALK_GENERAL", "1")
            else:
                monkeypatch.delenv("SK_WALK_GENERAL", raising=False)
            segs, nsegs = api.segment_batch(sig, lens, p, max_segs=64)
            if osegs is None:
                osegs, onsegs = ora.segment_batch_i16(sig, lens, ora.SegParams(**kw), lo=p.lim_low, hi=p.lim_hi,
                                                      max_segs=segs.shape[1])
            assert np.array_equal(nsegs, onsegs), (kw, general, np.nonzero(nsegs != onsegs)[0][:5])
            for r in range(192):
                assert np.array_equal(segs[r, :nsegs[r]], osegs[r, :nsegs[r]]), (kw, general, r)
    monkeypatch.delenv("SK_WALK_GENERAL", raising=False)


def test_long_read_chunks_numpy_sum_order(gpu, ora):
    """n > 8192 exercises numpy's chunked pairwise summation inside np.std."""
    from squigglekit_amd import api, synth
    sig = synth.squiggle_batch(6, 40000, 555)
    lens = np.array([40000, 8192, 8193, 16385, 20001, 36977], dtype=np.int32)
    segs, nsegs = api.segment_batch(sig, lens)
    osegs, onsegs = ora.segment_batch_i16(sig, lens, max_segs=segs.shape[1])
    assert np.array_equal(nsegs, onsegs)
    for r in range(6):
        assert np.array_equal(segs[r, :nsegs[r]], osegs[r, :nsegs[r]])


def test_invalid_params_are_loud(gpu):
    from squigglekit_amd import api
    from squigglekit_amd._lib import SquiggleKitError
    with pytest.raises(SquiggleKitError):
        api.segment_batch(np.full((1, 64), 500, dtype=np.int16), None, _params(dict(corrector=-1)))


def _check_vs_oracle(api, ora, sig, lens, kw, label, max_segs=24):
    p = _params(kw)
    segs, nsegs = api.segment_batch(sig, lens, p, max_segs=max_segs)
    okw = {k: v for k, v in kw.items() if k not in ("lim_low", "lim_hi")}
    osegs, onsegs = ora.segment_batch_i16(sig, lens, ora.SegParams(**okw), lo=p.lim_low, hi=p.lim_hi,
                                          max_segs=segs.shape[1])
    assert np.array_equal(nsegs, onsegs), (label, kw, np.nonzero(nsegs != onsegs)[0][:5])
    keep = np.arange(segs.shape[1])[None, :] < nsegs[:, None]
    assert np.array_equal(segs[keep], osegs[keep]), (label, kw)
    assert not segs[~keep].any(), "slots past nsegs must read as zero"
    return nsegs


def _streaming_cases(rng):
    """Reads of up to 4 096 samples (the streaming statistics path, sk_segstat.hip) with everything that path
    treats specially: ragged tails, outliers anywhere (also runs of them, and most of a read), reads that are
    empty after the filter, constant reads, reads shorter than one mask entry."""
    from squigglekit_amd import synth
    sig = synth.squiggle_batch(320, 4096, 424299)
    lens = rng.integers(1, 4097, size=320).astype(np.int32)
    lens[:12] = [1, 2, 7, 8, 9, 63, 64, 65, 511, 512, 513, 4096]
    lens[64:192] = 4096
    sig[12, :] = 0                                   # empty after the filter
    sig[13, :] = 901
    sig[14, :] = 500                                 # std == 0: empty band
    sig[15, :2000] = 500; sig[15, 2000:] = 501       # V > 0, tiny std
    for r in range(16, 48):                          # many outliers: single, runs, and a read that is mostly outliers
        k = int(rng.integers(1, 3000))
        pos = rng.integers(0, 4096, size=k)
        sig[r, pos] = rng.choice(np.array([-5, 0, 900, 950, 1100, 32767, -32768], dtype=np.int16), size=k)
    sig[48, 100:900] = 0
    sig[49, :4000] = 1000                            # only the tail survives
    sig[50, 64:] = 0                                 # only the first entry survives
    sig[51, ::2] = 0                                 # every other sample dropped
    return sig, lens


STREAM_PARAMS = [dict(), dict(error=10, corrector=0), dict(error=12, corrector=3, window=30),
                 dict(window=10, seg_dist=0, stall_len=0.0), dict(std_scale=3.0), dict(std_scale=-0.5),
                 dict(error=0), dict(window=1, error=0, seg_dist=1000), dict(lim_low=400, lim_hi=600),
                 dict(lim_low=-10, lim_hi=1500), dict(lim_low=499, lim_hi=502), dict(lim_low=-32769, lim_hi=-30722),
                 dict(std_scale=1e-9), dict(std_scale=40.0), dict(stall_len=1.5), dict(lim_low=0, lim_hi=2048)]


def test_streaming_statistics_path_vs_oracle(gpu, ora):
    from squigglekit_amd import api
    sig, lens = _streaming_cases(np.random.default_rng(20260928))
    total = 0
    for kw in STREAM_PARAMS:
        total += int(_check_vs_oracle(api, ora, sig, lens, kw, "streaming").sum())
    assert total > 2000                              # the cases do produce segments
    # narrower rows use the 2- and 4-tile instantiations
    for width in (8, 1000, 1024, 1032, 2048, 2056):
        _check_vs_oracle(api, ora, np.ascontiguousarray(sig[:, :width]), np.minimum(lens, width), dict(),
                         "width %d" % width)


@pytest.mark.parametrize("M", [4096, 9000])
def test_jumping_walk_on_pattern_reads(gpu, ora, monkeypatch, M):
    """k_seg_walk4 against the oracle on reads built to defeat its jumps: anchors thousands of samples behind the
    stretch they serve, more stretches than its list holds, isolated quiet entries before the first segment, dropped
    samples inside runs and between anchor and stretch, window at and around the 127 samples the jumps need, error
    0 .. 32; and the same batch through the walk without jumps and the word-synchronous walk."""
    from squigglekit_amd import api, synth
    rng = np.random.default_rng(M)
    R = 256
    sig = synth.pattern_reads(rng, R, M)
    lens = rng.integers(M // 2, M + 1, size=R).astype(np.int32)
    lens[:64] = M
    cases = [dict(), dict(stall_len=1.5), dict(window=127), dict(window=126), dict(window=300, stall_len=0.1),
             dict(error=0), dict(error=31, corrector=40), dict(error=32, corrector=40), dict(error=2, seg_dist=0),
             dict(std_scale=0.3), dict(lim_low=250, lim_hi=760)]
    total = 0
    for kw in cases:
        total += int(_check_vs_oracle(api, ora, sig, lens, kw, "patterns M=%d" % M, max_segs=40).sum())
    assert total > 1000
    for env in ("SK_WALK_NOJUMP", "SK_WALK_SYNC"):
        monkeypatch.setenv(env, "1")
        for kw in cases[:3]:
            _check_vs_oracle(api, ora, sig, lens, kw, "patterns M=%d %s" % (M, env), max_segs=40)
        monkeypatch.delenv(env)


def test_streaming_path_retry_list_and_old_kernels_agree(gpu, ora, monkeypatch):
    """SK_SEG_DELTA_SCALE widens the certification margin until (nearly) every read fails it, so the numpy-order
    redo of listed reads is what produces the masks; SK_SEG_OLD runs the numpy-order kernels for everything;
    SK_WALK_STEP takes the per-sample walk instead of the run-hopping one, SK_WALK_SYNC the run-hopping walk that keeps
    a wavefront's lanes on one word, SK_WALK_NOJUMP the default walk without its jumps; SK_SEG_CHUNKS overlaps walk and
    statistics on two streams.  All must give the oracle's segments."""
    from squigglekit_amd import api
    sig, lens = _streaming_cases(np.random.default_rng(7))
    for env, val in (("SK_SEG_DELTA_SCALE", "1e13"), ("SK_SEG_DELTA_SCALE", "3e10"), ("SK_SEG_OLD", "1"),
                     ("SK_WALK_STEP", "1"), ("SK_WALK_SYNC", "1"), ("SK_WALK_NOJUMP", "1"), ("SK_SEG_CHUNKS", "3")):
        monkeypatch.setenv(env, val)
        for kw in STREAM_PARAMS[:9]:
            _check_vs_oracle(api, ora, sig, lens, kw, "%s=%s" % (env, val))
        monkeypatch.delenv(env)


@pytest.mark.parametrize("M", [4104, 9000, 16384, 16392, 20000, 36984, 49160, 65536, 65544])
def test_streaming_path_long_reads(gpu, ora, monkeypatch, M):
    """Reads longer than one 4 096-sample window: the statistics kernel looks at them twice (window by window);
    with the certification margin blown up, the numpy-order redo runs on them too (LDS-resident copy up to
    ~14 000 samples, a scratch row per workgroup beyond)."""
    from squigglekit_amd import api, synth
    rng = np.random.default_rng(M)
    R = 40
    sig = synth.squiggle_batch(R, M, 1000 + M)
    lens = rng.integers(1, M + 1, size=R).astype(np.int32)
    lens[:8] = np.minimum([M, M - 1, 4096, 4097, 8192, 8193, 1, 4095], M)
    sig[8, :] = 0
    sig[9, :] = 500
    sig[10, 3000:M - 50] = 0                         # a hole of dropped samples across window borders
    k = 5000
    sig[11, rng.integers(0, M, k)] = rng.choice(np.array([-5, 0, 950, 1100], dtype=np.int16), k)
    cases = [dict(), dict(error=12, corrector=3, window=30), dict(std_scale=2.0, window=400), dict(lim_low=300, lim_hi=700)]
    for kw in cases:
        _check_vs_oracle(api, ora, sig, lens, kw, "long M=%d" % M, max_segs=160)
    monkeypatch.setenv("SK_SEG_DELTA_SCALE", "1e13")
    for kw in cases[:2]:
        _check_vs_oracle(api, ora, sig, lens, kw, "long M=%d, redo" % M, max_segs=160)
    monkeypatch.delenv("SK_SEG_DELTA_SCALE")
    # rows of up to 65 536 samples take the workgroup-per-read statistics kernel (one look, round 6): the
    # wavefront-per-read one (two looks) must give the same records
    for env in ("SK_SEG_NO_WG", "SK_SEG_WG_ALL"):
        monkeypatch.setenv(env, "1")
        for kw in cases[:2]:
            _check_vs_oracle(api, ora, sig, lens, kw, "long M=%d, %s" % (M, env), max_segs=160)
        monkeypatch.delenv(env)


@pytest.mark.parametrize("M", [20000, 70000, 140000])
def test_wave_per_read_walk_on_pattern_reads_long_rows_and_overflow(gpu, ora, monkeypatch, walk_shape, M):
    """k_seg_walkL: a wavefront per read, every lane from its piece's first anchor to the next lane's.  Pattern reads
    (anchors thousands of samples apart, near-miss anchors, dropped samples) at lengths whose rows are staged in LDS
    (20 000), are not (70 000, 140 000: beyond 1 024 entries), and -- with a short window -- hold more long runs in one
    piece than a lane's list (the lane-0 sequential fallback); the same records from the lane-per-read walk."""
    from squigglekit_amd import api, synth
    rng = np.random.default_rng(M + 1)
    R = 24
    sig = synth.pattern_reads(rng, R, M)
    lens = rng.integers(M // 2, M + 1, size=R).astype(np.int32)
    lens[:4] = [M, M - 1, M - 63, M - 64]
    # a read of trains: 40 in-band samples, 7 out-of-band ones -- with window 30 a 2 000-sample piece closes 40 long runs
    tr = np.where((np.arange(M) % 47) < 40, 500, np.where(np.arange(M) % 2 == 0, 300, 700)).astype(np.int16)
    sig[5, :] = tr
    sig[6, :] = tr
    sig[6, rng.integers(0, M, 50)] = 0               # ... and dropped samples in it
    total = 0
    for kw in (dict(), dict(window=127), dict(window=30, seg_dist=0), dict(window=30, error=0, seg_dist=5),
               dict(error=20, corrector=30), dict(stall_len=1.5), dict(lim_low=250, lim_hi=760)):
        total += int(_check_vs_oracle(api, ora, sig, lens, kw, "wave walk M=%d %s" % (M, walk_shape), max_segs=4096).sum())
    assert total > 1000
