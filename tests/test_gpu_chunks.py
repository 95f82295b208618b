"""GPU: the chunked DTW path and BASELINE.json's full-size MotifSeq configs.

A screening call is split into chunks of (scratch budget / per-read scratch) reads
(`sk_launch_sdtw_screen`), with chunk-relative indexing of the checkpoints, the last-row costs and the
per-read flags.  With the default 12 GB budget only batches beyond ~280 000 reads x 4 000 samples reach a
second chunk, so (1) SK_DTW_SCRATCH_MB forces many small chunks on a batch the oracle covers read for read,
and (2) C4 (1 000 000 x 4 000 x 200-pt) and C5 (100 000 x 20 000 x 500-pt) run at full size from the device
generator, with a strided sample spanning every chunk compared against the oracle."""
import ctypes as C

import numpy as np
import pytest

from conftest import download_rows, oracle_motifseq_threaded, strided_rows

pytestmark = pytest.mark.gpu


def _same(got, want, label):
    bad = np.nonzero((got["start"] != want["start"]) | (got["end"] != want["end"]) | (got["n"] != want["n"])
                     | ~((got["dist"] == want["dist"]) | (np.isnan(got["dist"]) & np.isnan(want["dist"]))))[0]
    assert bad.size == 0, "%s: %d reads differ, first %s: got %s want %s" % (
        label, bad.size, bad[:5], got[bad[:5]], want[bad[:5]])


def _dtw_profile(L):
    da, sb = C.c_float(), C.c_float()
    la, lb, rpl = C.c_int32(), C.c_int32(), C.c_int32()
    assert L.sk_last_dtw_profile(C.byref(da), C.byref(la), C.byref(sb), C.byref(lb), C.byref(rpl)) == 0
    return la.value, rpl.value


@pytest.mark.parametrize("scheme", ["screen", "exact2"])
def test_forced_small_chunks_every_read(gpu, ora, monkeypatch, scheme):
    """~3 000 ragged reads in >= 3 chunks (here 17+): every read equals the oracle, including the reads next to
    chunk boundaries and reads that need the exact retry in chunks >= 1."""
    from squigglekit_amd import api, synth
    L = gpu.load()
    R, M = 3001, 4000
    motif = synth.synthetic_motif(200, seed=3)
    sig = synth.squiggle_batch(R, M, 31337, motif=motif)
    rng = np.random.default_rng(5)
    lens = np.full(R, M, dtype=np.int32)
    lens[rng.choice(R, 600, replace=False)] = rng.integers(1, M + 1, 600)
    lens[[0, 1, 177, 178, 179, R - 1]] = [M, 17, 3999, 1, 2500, 777]
    # a 4x time-stretched copy of the motif: its optimal path is far wider than the look-back window, so the
    # read cannot be certified and takes the exact retry -- placed in many different chunks
    stretched = np.clip(np.rint(np.repeat(motif, 4) * 93.4 + 511.0), 1, 1199).astype(np.int16)
    forced = np.arange(7, R, 97)
    for r in forced:
        lens[r] = M
        off = 100 + (int(r) * 37) % (M - 1000)
        sig[r, off:off + stretched.size] = stretched
    if scheme == "exact2":
        monkeypatch.setenv("SK_DTW_SCHEME", "exact2")
    one = api.motifseq_batch(sig, lens, motif)                       # default budget: one chunk
    launches1, _ = _dtw_profile(L)
    monkeypatch.setenv("SK_DTW_SCRATCH_MB", "8")
    # a look-back far shorter than the motif: most optimal paths cross the restart front, so reads of every
    # chunk go through the exact retry
    monkeypatch.setenv("SK_DTW_SPAN", "40")
    got = api.motifseq_batch(sig, lens, motif)
    launches, per_launch = _dtw_profile(L)
    retries = L.sk_last_dtw_retries()
    assert launches1 == 1 and launches >= 3, (launches1, launches)
    assert per_launch < R // 3
    want = oracle_motifseq_threaded(ora, sig, lens, motif)
    ok = (got["flags"] & 2) == 0                                     # MAD == 0 reads: flagged, not compared
    assert (~ok).sum() <= 8 and np.all(got["n"][~ok] < 50)           # (only the tiny ragged reads)
    _same(got[ok], want[ok], "chunked (%d chunks of <= %d reads)" % (launches, per_launch))
    _same(one[ok], want[ok], "one chunk")
    assert np.array_equal(got["flags"], one["flags"]) and np.array_equal(got["n"], want["n"])
    assert retries >= R // 4, "the short look-back should have sent many reads to the retry (%d)" % retries
    assert (got["end"][forced] - got["start"][forced]).max() > 300


def _full_size(gpu, ora, R, M, N, seed, nsample, min_chunks):
    from squigglekit_amd import synth
    from squigglekit_amd._lib import HIT_DTYPE, check, ptr
    L = gpu.load()
    stride = (M + 7) // 8 * 8
    motif = synth.synthetic_motif(N)
    d_sig = L.sk_dev_alloc(R * stride * 2)
    d_len = L.sk_dev_alloc(R * 4)
    d_out = L.sk_dev_alloc(R * HIT_DTYPE.itemsize)
    assert d_sig and d_len and d_out, L.sk_last_error()
    try:
        lens = np.full(R, M, dtype=np.int32)
        check(L.sk_dev_upload(d_len, ptr(lens), lens.nbytes))
        check(L.sk_synth_squiggles_dev(d_sig, stride, R, M, seed, ptr(motif), N))
        check(L.sk_motifseq_dev_i16(d_sig, stride, d_len, R, ptr(motif), N, 0, 0, 1200, d_out))
        check(L.sk_sync())
        launches, per_launch = _dtw_profile(L)
        assert launches >= min_chunks, "expected the chunked path (%d launches)" % launches
        # the run-time guard of the screening certificate: one read in 4 096 re-run by the exact pass, every accepted
        # window result tested against the screening values it rests on -- nothing to report on a healthy build
        from squigglekit_amd import api
        g = api.last_dtw_guard()
        assert g["audited"] == (R + 4095) // 4096, g
        assert g["premise_violations"] == 0 and g["audit_mismatches"] == 0 and g["exact_fallback"] == 0, g
        hits = np.empty(R, dtype=HIT_DTYPE)
        check(L.sk_dev_download(ptr(hits), d_out, hits.nbytes))
        # size-independent properties over the WHOLE batch
        assert np.all(hits["n"] > 0) and np.all(hits["n"] <= M)
        assert np.all((0 <= hits["start"]) & (hits["start"] <= hits["end"]) & (hits["end"] < hits["n"]))
        assert np.all(np.isfinite(hits["dist"])) and np.all(hits["dist"] >= 0)
        # strided sample over every chunk (runs of 4 reads, so chunk-boundary neighbours are included)
        rows = strided_rows(R, nsample)
        bounds = np.arange(per_launch, R, per_launch)               # first read of chunks 1, 2, ...
        rows = np.unique(np.concatenate([rows, bounds, bounds - 1]))
        assert np.unique(rows // per_launch).size == launches
        sample = download_rows(L, d_sig, stride * 2, rows, np.int16, stride)
        want = oracle_motifseq_threaded(ora, sample, lens[:len(rows)], motif)
        _same(hits[rows], want, "full size %d x %d x %d-pt" % (R, M, N))
    finally:
        L.sk_dev_free(d_sig); L.sk_dev_free(d_len); L.sk_dev_free(d_out)


@pytest.mark.parametrize("budget_mb", [None, 12288])
def test_c4_full_size_1m_reads(gpu, ora, monkeypatch, budget_mb):
    """BASELINE configs[3] on one GPU: 1 000 000 x 4 000 int16 vs a 200-pt motif -- with the default scratch budget
    (64 GB: one chunk) and with 12 GB (4 chunks)."""
    from squigglekit_amd import synth
    if budget_mb:
        monkeypatch.setenv("SK_DTW_SCRATCH_MB", str(budget_mb))
    _full_size(gpu, ora, 1_000_000, 4000, 200, synth.SEED_C4, 12000, 3 if budget_mb else 1)   # (1.2 % of the reads against the oracle)


@pytest.mark.parametrize("budget_mb", [None, 12288])
def test_c5_full_size_100k_long_reads(gpu, ora, monkeypatch, budget_mb):
    """BASELINE configs[4] on one GPU: 100 000 x 20 000 int16 vs a 500-pt motif (L = 64 kernels); one chunk by
    default, 4 with a 12 GB scratch budget."""
    from squigglekit_amd import synth
    if budget_mb:
        monkeypatch.setenv("SK_DTW_SCRATCH_MB", str(budget_mb))
    _full_size(gpu, ora, 100_000, 20000, 500, synth.SEED_C5, 3200, 2 if budget_mb else 1)


def test_early_retry_equals_late_retry(gpu, ora, monkeypatch):
    """Reads pass Q itself cannot screen -- two identical copies of the motif far apart (candidate columns > 512 apart),
    a sample far outside the fixed-point range after normalisation -- are retried on a third stream beside the window
    passes; the result must equal the run with that switched off (SK_DTW_NO_EARLY) and the oracle, also when the
    scratch budget forces several chunks."""
    from squigglekit_amd import api, synth
    L = gpu.load()
    R, M = 1500, 4000
    motif = synth.synthetic_motif(200, seed=21)
    sig = synth.squiggle_batch(R, M, 777001, motif=motif)
    lens = np.full(R, M, dtype=np.int32)
    copy = np.clip(np.rint(motif * 93.4 + 511.0), 1, 1199).astype(np.int16)
    twice = np.arange(3, R, 41)
    for r in twice:                                       # the same samples twice: two exactly equal minima
        sig[r, 300:300 + copy.size] = copy
        sig[r, 2900:2900 + copy.size] = copy
    flat = np.arange(11, R, 97)
    for r in flat:                                        # MAD of a few units and one sample hundreds of MADs away
        sig[r, :] = 500 + (np.arange(M) % 3)
        sig[r, 1234] = 1190
    # (round 5: a read with two clusters of candidate columns gets a second window instead of the exact pass --
    # SK_DTW_NO_SIBLINGS=1 is round 4's behaviour, which this test is about; the default is compared with it below)
    with_siblings = api.motifseq_batch(sig, lens, motif)
    n_sib = api.last_dtw_guard()["second_windows"]
    n_retry_sib = L.sk_last_dtw_retries()
    monkeypatch.setenv("SK_DTW_NO_SIBLINGS", "1")
    got = api.motifseq_batch(sig, lens, motif)
    n_early = L.sk_last_dtw_retries()
    assert with_siblings.tobytes() == got.tobytes()
    assert n_sib >= twice.size // 2 and n_retry_sib < n_early, (n_sib, n_retry_sib, n_early)
    monkeypatch.setenv("SK_DTW_NO_EARLY", "1")
    late = api.motifseq_batch(sig, lens, motif)
    assert got.tobytes() == late.tobytes() and n_early == L.sk_last_dtw_retries()
    monkeypatch.delenv("SK_DTW_NO_EARLY")
    monkeypatch.setenv("SK_DTW_SCRATCH_MB", "8")
    chunked = api.motifseq_batch(sig, lens, motif)
    assert chunked.tobytes() == got.tobytes()
    monkeypatch.delenv("SK_DTW_NO_SIBLINGS")
    chunked_sib = api.motifseq_batch(sig, lens, motif)             # second windows in every chunk
    assert chunked_sib.tobytes() == got.tobytes() and api.last_dtw_guard()["second_windows"] == n_sib
    assert n_early >= twice.size, "the doubled reads should have gone to the exact retry (%d)" % n_early
    want = oracle_motifseq_threaded(ora, sig, lens, motif)
    ok = (got["flags"] & 2) == 0
    _same(got[ok], want[ok], "early retry")
    # (where the doubled copy holds the minimum at all, the FIRST of the two equal columns wins: first argmin)
    assert np.array_equal(got["end"][twice], want["end"][twice]) and np.mean(want["end"][twice] < 600) > 0.8
    assert not np.any((want["end"][twice] > 3000) & (want["end"][twice] < 3200))


@pytest.mark.parametrize("lanes", [None, "8", "64"])
def test_sorted_window_passes_equal_file_order(gpu, ora, monkeypatch, lanes):
    """Large chunks go through the window passes sorted by how many blocks they need (a counting sort on pass Q's
    epilogue records).  Forced onto a small ragged batch (SK_DTW_SORT_MIN=1) -- reads of many lengths, reads that are
    not screened at all, more reads than a multiple of 8, several chunks -- the records must equal the file-order run
    byte for byte, and the oracle."""
    from squigglekit_amd import api, synth
    R, M = 2003, 4000
    motif = synth.synthetic_motif(200, seed=5)
    sig = synth.squiggle_batch(R, M, 99017, motif=motif)
    rng = np.random.default_rng(4)
    lens = rng.integers(900, M + 1, R).astype(np.int32)
    lens[::97] = rng.integers(0, 300, lens[::97].size)              # too short for the screening scheme's window
    sig[5, :] = 500                                                  # MAD = 0
    if lanes:
        monkeypatch.setenv("SK_DTW_QL", lanes)
    monkeypatch.setenv("SK_DTW_NOSORT", "1")
    plain = api.motifseq_batch(sig, lens, motif)
    monkeypatch.delenv("SK_DTW_NOSORT")
    monkeypatch.setenv("SK_DTW_SORT_MIN", "1")
    got = api.motifseq_batch(sig, lens, motif)
    assert got.tobytes() == plain.tobytes()
    monkeypatch.setenv("SK_DTW_SCRATCH_MB", "8")                     # several chunks, each sorted by itself
    assert api.motifseq_batch(sig, lens, motif).tobytes() == plain.tobytes()
    want = oracle_motifseq_threaded(ora, sig, lens, motif)
    ok = (got["flags"] & 2) == 0
    _same(got[ok], want[ok], "sorted window passes")
