"""GPU, BASELINE.json's full single-GPU configs (C2: segmenter 10 000 x 4 000; C3: MotifSeq 10 000 x 4 000
vs the example model): too big for the oracle to cover in seconds, so parity is shown through
size-independent properties plus an oracle comparison on a random subset."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_c3_motifseq_full_size_properties(gpu, ora, example_model):
    from squigglekit_amd import api, synth
    R, M = 10000, 4000
    sig = synth.squiggle_batch(R, M, synth.SEED_C3, motif=example_model)
    lens = np.full(R, M, dtype=np.int32)
    hits = api.motifseq_batch(sig, lens, example_model)                      # two-pass path
    # ranges and ordering
    assert np.all(hits["n"] > 0) and np.all(hits["n"] <= M)
    assert np.all((0 <= hits["start"]) & (hits["start"] <= hits["end"]) & (hits["end"] < hits["n"]))
    assert np.all(np.isfinite(hits["dist"])) and np.all(hits["dist"] >= 0)
    # permutation invariance: a shuffled batch gives the same records, shuffled
    rng = np.random.default_rng(0)
    perm = rng.permutation(R)
    hits_p = api.motifseq_batch(sig[perm], lens, example_model)
    assert np.array_equal(hits_p, hits[perm])
    # batch-size independence: the single-pass kernel (small batch) agrees with the two-pass one
    sub = rng.choice(R, 200, replace=False)
    small = api.motifseq_batch(sig[sub], lens[:200], example_model)
    assert np.array_equal(small, hits[sub])
    # oracle on a tenth of the batch: bit-identical
    from conftest import oracle_motifseq_threaded
    big = rng.choice(R, 1000, replace=False)
    wbig = oracle_motifseq_threaded(ora, sig[big], lens[:1000], example_model)
    for f in ("dist", "start", "end", "n"):
        assert np.array_equal(hits[f][big], wbig[f]), f
    want = ora.motifseq_batch_i16(sig[sub], lens[:200], example_model)
    assert np.array_equal(hits["start"][sub], want["start"]) and np.array_equal(hits["end"][sub], want["end"])
    assert np.array_equal(hits["dist"][sub], want["dist"])
    # anchoring: DTW of the motif against just the matched window reproduces dist, start 0, end = len-1
    ys = []
    for r in sub[:48]:
        y = api.normalise(sig[r])
        ys.append(y[hits["start"][r]:hits["end"][r] + 1])
    w = api.dtw_subsequence_batch(example_model, ys)
    assert np.array_equal(w["dist"], hits["dist"][sub[:48]])
    assert np.all(w["start"] == 0) and np.array_equal(w["end"], np.array([len(y) - 1 for y in ys]))


def test_c2_segmenter_full_size_properties(gpu, ora):
    from squigglekit_amd import api, synth
    R, M = 10000, 4000
    sig = synth.squiggle_batch(R, M, synth.SEED_C2)
    lens = np.full(R, M - 1, dtype=np.int32)                                 # Num = -1
    segs, nsegs = api.segment_batch(sig, lens)
    assert nsegs.min() >= 0 and nsegs.max() <= segs.shape[1]
    frac = [(nsegs == k).mean() for k in (1, 2)]
    assert frac[0] > 0.3 and frac[1] > 0.3                                   # ~50 % / ~50 %
    for r in range(R):                                                       # sorted, disjoint, inside the read
        s = segs[r, :nsegs[r]]
        if len(s):
            assert np.all(s[:, 0] < s[:, 1]) and np.all(s[:, 1] <= M)
            assert np.all(s[1:, 0] - s[:-1, 1] >= 50)                        # merged when closer than seg_dist
    rng = np.random.default_rng(1)
    perm = rng.permutation(R)
    segs_p, nsegs_p = api.segment_batch(sig[perm], lens)
    assert np.array_equal(nsegs_p, nsegs[perm]) and np.array_equal(segs_p, segs[perm])
    sub = rng.choice(R, 2000, replace=False)
    osegs, onsegs = ora.segment_batch_i16(sig[sub], lens[:2000], max_segs=segs.shape[1])
    assert np.array_equal(onsegs, nsegs[sub])
    for k, r in enumerate(sub):
        assert np.array_equal(osegs[k, :onsegs[k]], segs[r, :nsegs[r]])
    # checksum of checksums, stable across runs of the same seed (regression anchor)
    again, nagain = api.segment_batch(sig, lens)
    assert np.array_equal(again, segs) and np.array_equal(nagain, nsegs)


def test_f64_route_full_size_250k_pa_reads(gpu, ora):
    """The float64 route at the size bench.py's other_paths block runs it (250 000 reads x 3 999 samples, 8 GB, device
    resident): (i) on integer-valued doubles it must agree with the int16 kernels for EVERY read -- two kernel
    families, one answer: segments and MotifSeq records byte for byte; (ii) on the pA image of the same reads a sample
    strided over the whole batch against the oracle, and the share of reads the streaming kernel could not certify
    stays tiny."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from conftest import download_rows, strided_rows
    from squigglekit_amd import synth
    from squigglekit_amd._lib import HIT_DTYPE, SegParams, check, ptr
    L = gpu.load()
    R, M, N, MAXS = 250_000, 4000, 200, 16
    Mf = M - 1
    motif = synth.synthetic_motif(N)
    bufs = []

    def alloc(nbytes):
        q = L.sk_dev_alloc(nbytes)
        assert q
        bufs.append(q)
        return q
    try:
        d_sig, d_len = alloc(R * M * 2), alloc(R * 4)
        check(L.sk_synth_squiggles_dev(d_sig, M, R, M, synth.SEED_C2, ptr(motif), N))
        lens = np.full(R, Mf, dtype=np.int32)
        check(L.sk_dev_upload(d_len, ptr(lens), lens.nbytes))
        d_f, d_off = alloc(R * Mf * 8), alloc((R + 1) * 8)
        d_segs, d_ns, d_segs2, d_ns2 = alloc(R * MAXS * 8), alloc(R * 4), alloc(R * MAXS * 8), alloc(R * 4)
        d_h, d_h2 = alloc(R * 24), alloc(R * 24)
        sp = SegParams()

        def fetch(d, shape, dtype):
            a = np.empty(shape, dtype=dtype)
            check(L.sk_dev_download(ptr(a), d, a.nbytes))
            return a
        # (i) integer-valued doubles: offset 0, range / digitisation = 1, two decimals change nothing
        check(L.sk_synth_pa_dev(d_sig, M, R, Mf, 0.0, 1.0, 1.0, d_f, d_off))
        check(L.sk_segment_dev_i16(d_sig, M, d_len, R, C.byref(sp), d_segs, d_ns, MAXS))
        check(L.sk_segment_dev_f64(d_f, d_off, R, R * Mf, Mf, C.byref(sp), d_segs2, d_ns2, MAXS))
        check(L.sk_sync())
        assert L.sk_last_f64_retries() >= 0
        ns, ns2 = fetch(d_ns, R, np.int32), fetch(d_ns2, R, np.int32)
        assert np.array_equal(ns, ns2) and 0.3 < (ns == 1).mean() < 0.7
        assert fetch(d_segs, (R, MAXS, 2), np.int32).tobytes() == fetch(d_segs2, (R, MAXS, 2), np.int32).tobytes()
        check(L.sk_motifseq_dev_i16(d_sig, M, d_len, R, ptr(motif), N, 0, 0, 1200, d_h))
        check(L.sk_motifseq_dev_f64(d_f, d_off, R, R * Mf, Mf, ptr(motif), N, 0, 0, 1200, d_h2))
        check(L.sk_sync())
        assert L.sk_last_f64_retries() >= 0
        h, h2 = fetch(d_h, R, HIT_DTYPE), fetch(d_h2, R, HIT_DTYPE)
        assert h.tobytes() == h2.tobytes()
        assert np.all(h["n"] > 3900) and np.all((0 <= h["start"]) & (h["start"] <= h["end"]) & (h["end"] < h["n"]))
        # (ii) the pA image (SquigglePull's np.round(..., 2)) against the oracle on a strided sample
        check(L.sk_synth_pa_dev(d_sig, M, R, Mf, 16.0, 1493.94, 8192.0, d_f, d_off))
        check(L.sk_segment_dev_f64(d_f, d_off, R, R * Mf, Mf, C.byref(sp), d_segs2, d_ns2, MAXS))
        check(L.sk_sync())
        retried = L.sk_last_f64_retries()
        assert 0 <= retried <= R // 1000, "%d of %d reads could not be certified" % (retried, R)
        check(L.sk_motifseq_dev_f64(d_f, d_off, R, R * Mf, Mf, ptr(motif), N, 0, 0, 1200, d_h2))
        check(L.sk_sync())
        ns2, segs2, h2 = fetch(d_ns2, R, np.int32), fetch(d_segs2, (R, MAXS, 2), np.int32), fetch(d_h2, R, HIT_DTYPE)
        rows = strided_rows(R, 1024)
        pa = download_rows(L, d_f, Mf * 8, rows, np.float64, Mf)
        op = ora.SegParams(sp.error, sp.corrector, sp.window, sp.seg_dist, sp.std_scale, sp.stall_len)

        def one(k):
            f = ora.scale_outliers(pa[k], 0, 900)
            want = ora.get_segs(f, op) or []
            f2 = ora.scale_outliers(pa[k], 0, 1200)
            return want, ora.dtw_subsequence(motif, ora.medmad(f2)[0]) + (f2.size,)
        with ThreadPoolExecutor(32) as ex:
            want = list(ex.map(one, range(len(rows))))
        for k, r in enumerate(rows):
            assert segs2[r, :ns2[r]].tolist() == want[k][0], r
            assert (h2["dist"][r], h2["start"][r], h2["end"][r], h2["n"][r]) == want[k][1], r
    finally:
        for q in bufs:
            L.sk_dev_free(q)


def test_pa_route_long_reads_full_length(gpu, ora):  # noqa: D401
    """fast5 / BLOW5-shaped input at the length of the read the reference ships (example/slow5/0.blow5: 36 978 samples):
    6 000 raw rows + channel constants through the raw-domain pA route (k_seg_stats<.., PA> + k_seg_walkL, round 6) --
    every record equal to the float64 route's (SK_SEG_PA_F64: the float64 image on the device, then the float64 kernels),
    and a strided sample against the oracle on the float64 values numpy makes the reference's way (segmenter.py:345-349)."""
    import os
    from conftest import strided_rows
    from squigglekit_amd import api, synth
    R, M = 6000, 36977
    S = (M + 7) // 8 * 8
    raw = synth.squiggle_batch(R, S, 20260929)
    rng = np.random.default_rng(9)
    lens = rng.integers(M // 2, M + 1, R).astype(np.int32)
    lens[:3] = [M, M - 1, 4097]
    calib = np.empty((R, 3))
    calib[:, 0], calib[:, 1], calib[:, 2] = 8192.0, np.round(rng.uniform(0, 30, R), 0), rng.uniform(1400, 1500, R)
    segs, nsegs = api.segment_batch_pa(raw, lens, calib, max_segs=128)
    assert api.last_pa_retries() == 0
    os.environ["SK_SEG_PA_F64"] = "1"
    try:
        fsegs, fn = api.segment_batch_pa(raw, lens, calib, max_segs=128)
        assert api.last_pa_retries() == -1
    finally:
        del os.environ["SK_SEG_PA_F64"]
    assert np.array_equal(nsegs, fn) and np.array_equal(segs, fsegs) and int(nsegs.sum()) > R
    for r in strided_rows(R, 48):
        unit = float("{0:.2f}".format(calib[r, 2])) / calib[r, 0]
        pa = np.round((raw[r, :lens[r]].astype(np.int64) + calib[r, 1]) * unit, 2)
        assert segs[r, :nsegs[r]].tolist() == (ora.get_segs(ora.scale_outliers(pa, 0, 900)) or []), r
