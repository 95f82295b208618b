"""GPU: the run-time guard of the screening scheme (csrc/sk_sdtwq.hip, sk_sdtw.hip).

The default DTW path is exact because of a certificate whose premise -- every fixed-point screening cost lies within
E = N + n + 2 units of the exact one -- is derived, not observed.  Round 4's fuzz found that premise silently false for
float64 reads after three rounds of green tests.  Since round 5 the premise is observed at run time:
  (a) the window pass tests every result it accepts against the screening values it rests on (premise violations),
  (b) one read in 4 096 is re-run by the exact single pass and compared (audit mismatches),
  (c) pass Q bounds the evaluation error of each read's sample image and keeps reads it cannot bound out of the screening,
  (d) any alarm from (a) or (b) makes the library redo the whole call with the exact pass.
These tests put two known holes back behind SK_DTW_HOLE and check that the guard notices them and that no wrong record
leaves the library; and that on healthy builds the counters stay at zero.  mlpy's one exact pass
(/root/reference/MotifSeq.py:437-439) is what every record must equal."""
import ctypes as C

import numpy as np
import pytest

from conftest import oracle_motifseq_threaded

pytestmark = pytest.mark.gpu


def _screened(gpu):
    launches = C.c_int32()
    gpu.load().sk_last_dtw_profile(None, C.byref(launches), None, None, None)
    return launches.value >= 1


def _same(got, want):
    return ((got["start"] == want["start"]) & (got["end"] == want["end"]) & (got["n"] == want["n"])
            & ((got["dist"] == want["dist"]) | (np.isnan(got["dist"]) & np.isnan(want["dist"]))))


def _near_constant_f64_reads(n_each=150):
    """the reads of the round-4 hole: spread 1e-14 of the level"""
    rng = np.random.default_rng(2)
    reads = [500.0 + rng.integers(0, 3, int(rng.integers(1500, 2600))) * 2.0 ** -40 for _ in range(n_each)]
    reads += [90.0 + rng.integers(0, 2000, int(rng.integers(1500, 2600))) * 2.0 ** -44 for _ in range(n_each)]
    return reads


def _oracle_f64(ora, reads, motif):
    from concurrent.futures import ThreadPoolExecutor

    def one(sig):
        f = ora.scale_outliers(sig, 0, 1200)
        y = ora.medmad(f)[0]
        return (ora.dtw_subsequence(motif, y) + (f.size,)) if np.all(np.isfinite(y)) else None
    with ThreadPoolExecutor(16) as ex:
        return list(ex.map(one, reads))


def test_guard_is_quiet_and_audits_on_a_healthy_build(gpu, ora, monkeypatch):
    """int16 batch through the default path: counters at zero, one audited read per 4 096 (and per SK_DTW_AUDIT_PERIOD
    when set), audited reads really were re-run (period 1: every read), records unchanged by the audit."""
    from squigglekit_amd import api, synth
    motif = synth.synthetic_motif(200)
    R, M = 9000, 4000
    sig = synth.squiggle_batch(R, M, 20260931, motif=motif)
    lens = np.full(R, M, dtype=np.int32)
    lens[::13] = np.random.default_rng(1).integers(1000, M, lens[::13].size)
    monkeypatch.setenv("SK_DTW_AUDIT_PERIOD", "4096")            # (explicit: every call is audited, see the next test)
    base = api.motifseq_batch(sig, lens, motif)
    assert _screened(gpu)
    g = api.last_dtw_guard()
    assert g["audited"] == 3 and g["premise_violations"] == 0 and g["audit_mismatches"] == 0, g
    assert g["alarm"] == 0 and g["exact_fallback"] == 0, g
    for period, expect in (("64", (R + 63) // 64), ("1", R), ("0", 0)):
        monkeypatch.setenv("SK_DTW_AUDIT_PERIOD", period)
        got = api.motifseq_batch(sig, lens, motif)
        g = api.last_dtw_guard()
        assert g["audited"] == expect and g["alarm"] == 0 and g["exact_fallback"] == 0, (period, g)
        assert got.tobytes() == base.tobytes()
    monkeypatch.delenv("SK_DTW_AUDIT_PERIOD")
    monkeypatch.setenv("SK_DTW_NOGUARD", "1")
    assert api.motifseq_batch(sig, lens, motif).tobytes() == base.tobytes()
    assert api.last_dtw_guard()["audited"] == 0
    monkeypatch.delenv("SK_DTW_NOGUARD")
    rows = np.arange(0, R, 9)
    want = oracle_motifseq_threaded(ora, sig[rows], lens[rows], motif)
    assert np.all(_same(base[rows], want))


def test_audit_of_small_calls_is_spread_over_calls(gpu):
    """A call whose window passes are shorter than one exact sweep would WAIT for the audit (9 000 reads: 0.37 ms of
    sweep beside 0.1 ms of windows).  By default such calls are audited one in K, K from the expected exposure (at most
    64), so that the average cost stays near 2 % of the call; large calls (test_gpu_chunks) are audited every time."""
    from squigglekit_amd import api, synth
    motif = synth.synthetic_motif(200)
    R, M = 9000, 4000
    sig = synth.squiggle_batch(R, M, 20260937, motif=motif)
    lens = np.full(R, M, dtype=np.int32)
    base, counts = None, []
    for _ in range(70):
        got = api.motifseq_batch(sig, lens, motif)
        g = api.last_dtw_guard()
        assert g["premise_violations"] == 0 and g["audit_mismatches"] == 0 and g["exact_fallback"] == 0, g
        counts.append(g["audited"])
        base = got if base is None else base
        assert got.tobytes() == base.tobytes()
    assert set(counts) == {0, 3}, counts
    assert 1 <= counts.count(3) <= 35, counts


def test_hole_e_equals_one_is_caught(gpu, ora, monkeypatch):
    """SK_DTW_HOLE=qerr1 runs the scheme with E = 1: "lower bounds" that are none, a candidate set that misses minima.
    Without the guard wrong records leave the library; with it the window pass's premise test fires, the call is
    redone by the exact pass and every record is right."""
    from squigglekit_amd import api, synth
    motif = synth.synthetic_motif(200)
    R, M = 6000, 4000
    sig = synth.squiggle_batch(R, M, 20260932, motif=motif)
    lens = np.full(R, M, dtype=np.int32)
    want = oracle_motifseq_threaded(ora, sig, lens, motif)
    monkeypatch.setenv("SK_DTW_HOLE", "qerr1")
    got = api.motifseq_batch(sig, lens, motif)
    assert _screened(gpu)
    g = api.last_dtw_guard()
    assert g["premise_violations"] > 0 and g["exact_fallback"] == 1, g
    assert np.all(_same(got, want)), "the guard let %d wrong records through" % int((~_same(got, want)).sum())
    # what the hole does when nobody looks (documented, not required: the count may be 0 on some data)
    monkeypatch.setenv("SK_DTW_NOGUARD", "1")
    bare = api.motifseq_batch(sig, lens, motif)
    print("E = 1 without the guard: %d of %d records wrong; with it: %s" % (int((~_same(bare, want)).sum()), R, g))


@pytest.mark.parametrize("hole", ["fma64", "fma64x"])
def test_hole_fma_image_for_float64_reads_is_caught(gpu, ora, example_model, monkeypatch, hole):
    """The round-4 hole put back: float64 reads imaged by fma(x, 2^22 / s, -c 2^22 / s).  fma64: pass Q's image-error
    bound keeps the near-constant reads out of the screening (image_rejects > 0, nothing else fires).  fma64x: that
    bound is switched off too, so the premise test / the audit have to notice -- and the call is redone exactly.
    Either way no wrong record."""
    from squigglekit_amd import api
    reads = _near_constant_f64_reads()
    want = _oracle_f64(ora, reads, example_model)
    monkeypatch.setenv("SK_DTW_HOLE", hole)
    monkeypatch.setenv("SK_DTW_AUDIT_PERIOD", "16")
    got = api.motifseq_reads_f64(reads, example_model, scale="medmad")
    assert _screened(gpu)
    g = api.last_dtw_guard()
    if hole == "fma64":
        assert g["image_rejects"] >= 100 and g["alarm"] == 0 and g["exact_fallback"] == 0, g
    else:
        assert g["image_rejects"] == 0 and g["alarm"] > 0 and g["exact_fallback"] == 1, g
    wrong = 0
    for r, w in enumerate(want):
        if w is None:
            assert got["flags"][r] & 2
        else:
            wrong += (got["dist"][r], got["start"][r], got["end"][r], got["n"][r]) != w
    assert wrong == 0, "%d wrong records with SK_DTW_HOLE=%s (%s)" % (wrong, hole, g)
    if hole == "fma64x":
        monkeypatch.setenv("SK_DTW_NOGUARD", "1")
        bare = api.motifseq_reads_f64(reads, example_model, scale="medmad")
        nwrong = sum(1 for r, w in enumerate(want)
                     if w is not None and (bare["dist"][r], bare["start"][r], bare["end"][r], bare["n"][r]) != w)
        print("fma image without any guard: %d of %d records wrong; with the guard: %s" % (nwrong, len(reads), g))
        assert nwrong > 0, "the hole is not a hole on this data: the test proves nothing"


def test_image_bound_rejects_extreme_int16_reads_only(gpu, ora):
    """int16 reads whose level / MAD ratio times their length exceeds what E's slack covers (level 30 000, MAD 0.5,
    60 000 samples -- out of any real squiggle's range; scale_outliers' default limits would drop every sample) are
    swept exactly; the ordinary reads of the same batch are not touched by the bound."""
    from squigglekit_amd import api, synth
    motif = synth.synthetic_motif(200)
    R, M = 400, 60000
    sig = np.empty((R, M), dtype=np.int16)
    sig[:] = 30000 + np.array([0, 1, 2, 1], dtype=np.int16)[np.arange(M) % 4]          # median 30 001, MAD 0.5
    sig[: R // 2] = synth.squiggle_batch(R // 2, M, 5, motif=motif)
    lens = np.full(R, M, dtype=np.int32)
    got = api.motifseq_batch(sig, lens, motif, scale_low=0, scale_hi=32767)
    g = api.last_dtw_guard()
    assert _screened(gpu)
    assert g["image_rejects"] == R // 2 and g["alarm"] == 0, g
    rows = np.r_[0:6, R // 2:R // 2 + 6]
    want = ora.motifseq_batch_i16(sig[rows], lens[rows], motif, lo=0, hi=32767)
    ok = (got["flags"][rows] & 2) == 0
    assert ok.sum() >= 10 and np.all(_same(got[rows][ok], want[ok]))
