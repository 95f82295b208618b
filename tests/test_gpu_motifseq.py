"""GPU parity: MotifSeq path (filter -> medmad/zscale -> subsequence DTW) vs the oracle.

Bar: start/end exact, distance bit-identical (the north-star tolerance is 1e-5;
FP64 min-plus is order independent so we assert equality and report the max
deviation if that ever fails)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
DIST_TOL = 1e-5   # BASELINE.json north_star tolerance


def _assert_hits(got, want, label=""):
    bad = np.nonzero((got["start"] != want["start"]) | (got["end"] != want["end"])
                     | (got["n"] != want["n"]))[0]
    assert bad.size == 0, "%s start/end/n mismatch at reads %s: got %s want %s" % (
        label, bad[:5], got[bad[:5]], want[bad[:5]])
    both_nan = np.isnan(got["dist"]) & np.isnan(want["dist"])
    dev = np.where(both_nan, 0.0, np.abs(got["dist"] - want["dist"]))
    assert np.all(dev <= DIST_TOL), "%s max |ddist| = %g" % (label, dev.max())
    assert np.all((got["dist"] == want["dist"]) | both_nan), \
        "%s distances within tol but not bit-identical (max dev %g)" % (label, dev.max())


@pytest.mark.parametrize("nx", [1, 2, 5, 15, 16, 17, 31, 64, 100, 163, 200, 255, 256, 257, 320, 500, 777, 1024,
                                1025, 1100, 2048, 2049, 3333])   # > 1024: row-chunked sweeps
def test_dtw_raw_vs_oracle(gpu, ora, nx):
    """mlpy boundary: dtw_subsequence(x, y) on pre-normalised float64 signals."""
    from squigglekit_amd import api
    rng = np.random.default_rng(1000 + nx)
    x = rng.normal(0, 1, nx)
    ys = []
    for ny in [1, 2, 3, 15, 16, 17, 63, 64, 65, 100, 257, 1000, 2500]:
        ys.append(rng.normal(0, 1, ny))
    # tie-heavy integer-valued cases exercise the back-trace tie order
    xi = rng.integers(-2, 3, nx).astype(float)
    yt = [rng.integers(-2, 3, ny).astype(float) for ny in [5, 40, 333, 1200]]
    for q, sigs in ((x, ys), (xi, yt)):
        got = api.dtw_subsequence_batch(q, sigs)
        for i, y in enumerate(sigs):
            d, s, e = ora.dtw_subsequence(q, y)
            assert (got["start"][i], got["end"][i]) == (s, e), (nx, len(y), got[i], (d, s, e))
            assert got["dist"][i] == d, (nx, len(y), got["dist"][i], d)
            assert got["n"][i] == len(y)


def test_dtw_single_pair_and_last_row(gpu, ora):
    from squigglekit_amd import api
    rng = np.random.default_rng(5)
    x = rng.normal(0, 1, 163)
    y = rng.normal(0, 1, 3000)
    dist, cost, path = api.dtw_subsequence(x, y, last_row=True)
    d, s, e, full = ora.dtw_subsequence(x, y, want_cost=True)
    assert (dist, path[1][0], path[1][-1]) == (d, s, e)
    assert np.array_equal(cost[-1, :], full[-1, :])
    assert np.array_equal(cost[-1, ], full[-1, ])


@pytest.mark.parametrize("scale", ["medmad", "zscale"])
@pytest.mark.parametrize("nmotif", [163, 200])
def test_motifseq_batch_synthetic(gpu, ora, example_model, scale, nmotif):
    """C3-shaped batch (reduced read count so the oracle finishes in seconds)."""
    from squigglekit_amd import api, synth
    motif = example_model if nmotif == 163 else synth.synthetic_motif(nmotif)
    sig = synth.squiggle_batch(384, 4000, synth.SEED_C3, motif=motif)
    lens = np.full(sig.shape[0], sig.shape[1], dtype=np.int32)
    got = api.motifseq_batch(sig, lens, motif, scale=scale)
    want = ora.motifseq_batch_i16(sig, lens, motif, scale_mode=0 if scale == "medmad" else 1)
    _assert_hits(got, want, "synthetic %s N=%d" % (scale, nmotif))


def test_motifseq_ragged_and_edge_reads(gpu, ora, example_model):
    from squigglekit_amd import api, synth
    sig = synth.squiggle_batch(40, 2048, 77, motif=example_model)
    lens = np.array([2048, 1, 2, 7, 8, 9, 63, 64, 65, 100, 163, 164, 500, 1000, 2047, 2048] * 2 + [2048] * 8,
                    dtype=np.int32)
    sig[20, :] = 0            # nothing survives the filter (0 < x strict)
    sig[21, :] = 1500         # nothing survives (x < 1200 strict)
    sig[22, :] = 500          # constant read: MAD == 0 -> degenerate under medmad
    sig[23, :1000] = 1199; sig[23, 1000:] = 1
    for scale in ("medmad", "zscale"):
        got = api.motifseq_batch(sig, lens, example_model, scale=scale)
        want = ora.motifseq_batch_i16(sig, lens, example_model, scale_mode=0 if scale == "medmad" else 1)
        ok = np.ones(len(lens), dtype=bool)
        if scale == "medmad":
            ok &= ~(want["n"] > 0) | np.isfinite(want["dist"])      # MAD==0 rows: reference divides by 0
        assert np.array_equal(got["n"], want["n"])
        _assert_hits(got[ok], want[ok], "ragged " + scale)
        assert got["flags"][20] & 1 and got["flags"][21] & 1
        assert np.isnan(got["dist"][20]) and got["start"][20] == -1 and got["end"][20] == -1
        if scale == "medmad":
            assert got["flags"][22] & 2


def test_motifseq_real_read_golden(gpu, ora, example_read):
    """Rows the reference printed for example/test.fast5 (DTW digits from the oracle stub)."""
    from squigglekit_amd import api
    gold = load_golden("motifseq_cli.json.gz")
    model32 = None
    raw = example_read["signal"]
    for run in gold["runs"]:
        if run["tsv"] != "real_raw" or run["flags"][:1] != ["-l"]:
            continue
        scale = run["flags"][1]
        row = run["stdout"].strip().split("\n")[1].split("\t")
        # the reference built the model through convert_fasta (float32-valued currents)
        if model32 is None:
            vals = np.array(gold["model_expanded"]["values"])
            model32 = vals.astype(np.float32).astype(np.float64)
        got = api.motifseq_batch(raw[None, :], np.array([raw.size], dtype=np.int32), model32, scale=scale)
        assert (int(row[3]), int(row[4])) == (got["start"][0], got["end"][0])
        assert float(row[6]) == got["dist"][0]


def test_normalise_matches_reference_vectors(gpu):
    """Normalised signals vs what the reference's numpy/sklearn code produced."""
    from squigglekit_amd import api, synth
    gold = load_golden("motifseq_norm.json.gz")
    model = np.array(load_golden("motifseq_cli.json.gz")["model_expanded"]["values"])
    sig = synth.squiggle_batch(6, 4000, synth.SEED_C3, motif=model)
    for v in gold["vectors"]:
        y = api.normalise(sig[v["read"]], scale=v["mode"])
        want = np.array(v["y"])
        assert y.shape == want.shape
        assert np.array_equal(y, want), (v["mode"], v["read"], np.abs(y - want).max())


def test_long_reads_and_long_motif(gpu, ora):
    """C5-shaped: 20 000-sample reads vs a 500-point motif (L=64 kernel)."""
    from squigglekit_amd import api, synth
    motif = synth.synthetic_motif(500, seed=11)
    sig = synth.squiggle_batch(12, 20000, synth.SEED_C5, motif=motif)
    lens = np.full(12, 20000, dtype=np.int32)
    got = api.motifseq_batch(sig, lens, motif)
    want = ora.motifseq_batch_i16(sig, lens, motif)
    _assert_hits(got, want, "C5")


def test_invalid_is_loud(gpu):
    from squigglekit_amd import api
    from squigglekit_amd._lib import SquiggleKitError
    sig = np.full((1, 64), 500, dtype=np.int16)
    with pytest.raises(SquiggleKitError):
        api.motifseq_batch(sig, None, np.zeros(0))


def test_two_pass_with_forced_retries(gpu, ora):
    """Batches big enough for the two-pass scheme (distance pass + windowed start pass), with
    reads whose optimal path is far longer than the look-back window (an exact, 4x time-stretched
    copy of the motif: cost 0 over ~800 columns), so the exact single-pass retry is exercised
    too.  Everything must stay bit-exact."""
    from squigglekit_amd import api, synth
    L = gpu.load()
    rng = np.random.default_rng(99)
    motif = synth.synthetic_motif(200, seed=3)
    stretched = np.repeat(motif, 4)
    ys = []
    for r in range(300):
        n = 3000 if r % 11 else 2600
        y = rng.normal(0.0, 1.2, n)
        if r % 3 == 0:
            off = 100 + (r * 37) % (n - 1000)
            y[off:off + stretched.size] = stretched
        ys.append(y)
    got = api.dtw_subsequence_batch(motif, ys)
    retries = L.sk_last_dtw_retries()
    for r, y in enumerate(ys):
        d, s, e = ora.dtw_subsequence(motif, y)
        assert (got["dist"][r], got["start"][r], got["end"][r]) == (d, s, e), r
    assert retries >= 90, "stretched reads should have crossed the restart front (got %d retries)" % retries
    assert (got["end"] - got["start"]).max() > 700


def test_two_pass_l64_long_motif(gpu, ora):
    """Two-pass on the L=64 kernel family (motif > 256 points) with ragged lengths."""
    from squigglekit_amd import api, synth
    motif = synth.synthetic_motif(300, seed=5)
    sig = synth.squiggle_batch(272, 6000, 1212, motif=motif)
    lens = np.full(272, 6000, dtype=np.int32)
    lens[::5] = 4500
    got = api.motifseq_batch(sig, lens, motif)
    want = ora.motifseq_batch_i16(sig, lens, motif)
    _assert_hits(got, want, "two-pass L64")


def test_multi_motif_batch(gpu, ora, example_model):
    """Several motifs of different lengths (both kernel families) against the same reads, with a
    float64 read mixed in: equals the per-motif oracle results, motif-major order."""
    from squigglekit_amd import api, synth
    motifs = [example_model, synth.synthetic_motif(40, seed=1), synth.synthetic_motif(300, seed=2), np.array([0.5])]
    sig = synth.squiggle_batch(24, 3000, 606, motif=example_model)
    reads = [sig[r] for r in range(24)]
    reads[5] = np.round((sig[5].astype(np.float64) + 16.0) * 0.1824, 2)        # a pA read
    outs = api.motifseq_multi(reads, motifs, scale="medmad")
    assert len(outs) == 4
    for k, m in enumerate(motifs):
        for r, x in enumerate(reads):
            f = ora.scale_outliers(np.asarray(x, float), 0, 1200)
            d, s, e = ora.dtw_subsequence(m, ora.medmad(f)[0])
            assert (outs[k]["dist"][r], outs[k]["start"][r], outs[k]["end"][r]) == (d, s, e), (k, r)


def test_screening_scheme_adversarial(gpu, ora):
    """The default DTW scheme screens in 32-bit fixed point and certifies an exact window; these
    inputs aim at its corners: exact and near ties between far-apart columns (two copies of the
    motif, one perturbed by 1e-7 or by nothing), values at the edge of / beyond the fixed-point
    range, constant and tiny reads, costs large enough to saturate.  Everything must equal the
    oracle bit for bit (the scheme falls back to the exact pass whenever it cannot certify)."""
    from squigglekit_amd import api
    L = gpu.load()
    rng = np.random.default_rng(2024)
    motif = np.round(rng.normal(0, 1.0, 120), 3)
    ys = []
    for r in range(288):
        n = 2700 + (r % 5) * 13
        y = rng.normal(0.0, 1.3, n)
        kind = r % 12
        if kind == 0:                                   # two exact copies: exact tie, first column wins
            y[200:320] = motif
            y[1800:1920] = motif
        elif kind == 1:                                 # second copy better by 1e-7
            y[300:420] = motif
            y[300] += 1e-7
            y[2000:2120] = motif
        elif kind == 2:                                 # first copy better by 1e-9 (below one fixed-point unit)
            y[2000:2120] = motif
            y[2000] += 1e-9
            y[400:520] = motif
        elif kind == 3:                                 # samples at the edge of the fixed-point range
            y[::97] = 399.9999
            y[50::101] = -399.9999
        elif kind == 4:                                 # samples beyond it -> exact fallback
            y[1234] = 1.0e6
            y[77] = -401.0
        elif kind == 5:                                 # huge costs everywhere (saturation)
            y += 350.0
        elif kind == 6:                                 # constant read
            y[:] = 0.25
        elif kind == 7:                                 # integer-valued, tie heavy
            y = rng.integers(-2, 3, n).astype(float)
        elif kind == 8:                                 # tiny read inside a big batch
            y = y[:int(rng.integers(1, 40))]
        ys.append(y)
    for q in (motif, np.round(motif)):                  # second query: integers -> ties with kind 7
        got = api.dtw_subsequence_batch(q, ys)
        for r, y in enumerate(ys):
            d, s, e = ora.dtw_subsequence(q, y)
            assert (got["dist"][r], got["start"][r], got["end"][r]) == (d, s, e), (r, r % 12)
    assert L.sk_last_dtw_retries() > 0                  # some of these must have taken the fallback


def test_exact_two_pass_scheme_still_exact(gpu, ora, monkeypatch):
    """Motif values outside the fixed-point range (and SK_DTW_SCHEME=exact2) select the exact FP64
    two-pass scheme (distance pass + checkpointed start pass); it must agree with the oracle too."""
    from squigglekit_amd import api
    rng = np.random.default_rng(77)
    motif = rng.normal(0, 1.0, 150)
    ys = [rng.normal(0.0, 1.2, 2900 + (r % 7)) for r in range(264)]
    for r in range(0, 264, 3):
        ys[r][500:650] = motif
    big = motif * 1000.0                                   # |x| >= 400: screening is not applicable
    got = api.dtw_subsequence_batch(big, [y * 1000.0 for y in ys])
    for r in range(0, 264, 11):
        assert (got["dist"][r], got["start"][r], got["end"][r]) == ora.dtw_subsequence(big, ys[r] * 1000.0)
    monkeypatch.setenv("SK_DTW_SCHEME", "exact2")
    got = api.dtw_subsequence_batch(motif, ys)
    for r in range(0, 264, 7):
        assert (got["dist"][r], got["start"][r], got["end"][r]) == ora.dtw_subsequence(motif, ys[r])


def test_long_motif_int16_batch(gpu, ora):
    """A 1 500-point motif (two chained row chunks) through the whole MotifSeq path: filter,
    medmad / zscale, DTW; ragged lengths, an empty read, reads shorter than the motif."""
    from squigglekit_amd import api, synth
    motif = synth.synthetic_motif(1500, seed=11)
    sig = synth.squiggle_batch(40, 5000, 8080, motif=motif[:400])
    lens = np.full(40, 5000, dtype=np.int32)
    lens[:6] = [0, 1, 700, 1499, 1500, 1501]
    sig[7, :] = 0
    for scale in ("medmad", "zscale"):
        got = api.motifseq_batch(sig, lens, motif, scale=scale)
        want = ora.motifseq_batch_i16(sig, lens, motif, scale_mode={"medmad": 0, "zscale": 1}[scale])
        ok = (got["flags"] & 2) == 0             # MAD == 0 (the one-sample read): flagged, not compared
        assert ok.sum() >= 38 and np.array_equal(got["n"], want["n"])
        _assert_hits(got[ok], want[ok], "long motif %s" % scale)


@pytest.mark.parametrize("nx", [17, 100, 163, 200, 256])
def test_small_batches_on_the_16_lane_layout(gpu, ora, monkeypatch, nx):
    """Few reads normally take the 64-lanes-per-read layout; SK_DTW_NO_SMALL keeps the 16-lane one
    (what large batches use), so that both layouts see the tie-heavy and ragged small cases."""
    from squigglekit_amd import api, synth
    monkeypatch.setenv("SK_DTW_NO_SMALL", "1")
    rng = np.random.default_rng(77 + nx)
    xi = rng.integers(-2, 3, nx).astype(float)
    yt = [rng.integers(-2, 3, ny).astype(float) for ny in [1, 5, 40, 333, 1200, 2500]]
    got = api.dtw_subsequence_batch(xi, yt)
    for i, y in enumerate(yt):
        d, s, e = ora.dtw_subsequence(xi, y)
        assert (got["dist"][i], got["start"][i], got["end"][i]) == (d, s, e), (nx, len(y))
    motif = synth.synthetic_motif(nx, seed=nx)
    sig = synth.squiggle_batch(300, 3000, 9000 + nx, motif=motif)       # >= 256 reads: screening scheme
    lens = rng.integers(1, 3001, 300).astype(np.int32)
    want = ora.motifseq_batch_i16(sig, lens, motif)
    got = api.motifseq_batch(sig, lens, motif)
    ok = (got["flags"] & 2) == 0
    _assert_hits(got[ok], want[ok], "16-lane layout, %d points" % nx)


def test_motifseq_after_stall_extension(gpu, ora, example_model):
    """[extension] search only after the stall the segmenter finds: equals the composition of the two
    reference stages -- get_segs on the filtered read, then MotifSeq on raw[cut:]."""
    from squigglekit_amd import api, synth
    sig = synth.squiggle_batch(48, 4000, 60606, motif=example_model)
    reads = [sig[r] for r in range(48)]
    hits, cuts = api.motifseq_after_stall(reads, example_model)
    ncut = 0
    for r, x in enumerate(reads):
        f = ora.scale_outliers(x.astype(float), 0, 900)
        segs = ora.get_segs(f)
        cut = 0
        if segs:
            kept = np.flatnonzero((x > 0) & (x < 900))
            e = segs[0][1]
            cut = int(kept[e]) if e < kept.size else x.size
            ncut += 1
        assert cuts[r] == cut
        y = ora.medmad(ora.scale_outliers(x[cut:].astype(float), 0, 1200))[0]
        d, s0, e0 = ora.dtw_subsequence(example_model, y)
        assert (hits["dist"][r], hits["start"][r], hits["end"][r]) == (d, s0, e0), r
    assert ncut >= 40                                  # the synthetic reads start with a stall


@pytest.mark.parametrize("scale", ["medmad", "zscale"])
def test_multi_motif_batch_through_the_fused_prologue(gpu, ora, scale):
    """Enough reads for the screening scheme: the first motif's pass Q carries the filter + statistics (medmad, and since
    round 5 zscale) as its prologue, the later motifs find the compacted samples / statistics in place -- records equal
    to one call per motif and to the oracle on a sample."""
    from squigglekit_amd import api, synth
    motifs = [synth.synthetic_motif(200, seed=1), synth.synthetic_motif(163, seed=2), synth.synthetic_motif(64, seed=3)]
    R, M = 900, 4000
    sig = synth.squiggle_batch(R, M, 60708, motif=motifs[0])
    lens = np.full(R, M, dtype=np.int32)
    lens[::11] = np.random.default_rng(3).integers(1500, M, lens[::11].size)
    outs = api.motifseq_multi_batch(sig, lens, motifs, scale=scale)
    mode = {"medmad": 0, "zscale": 1}[scale]
    rows = np.arange(0, R, 7)
    for k, m in enumerate(motifs):
        one = api.motifseq_batch(sig, lens, m, scale=scale)
        assert outs[k].tobytes() == one.tobytes(), (scale, k)
        want = ora.motifseq_batch_i16(sig[rows], lens[rows], m, scale_mode=mode)
        for f in ("dist", "start", "end", "n"):
            assert np.array_equal(outs[k][f][rows], want[f]), (scale, k, f)


def test_two_clusters_of_candidate_columns_take_a_second_window(gpu, ora, monkeypatch):
    """Reads that hold the motif twice, far apart (candidate columns in two clusters more than 512 columns from each
    other): the second cluster is a SIBLING of the read in a second round of the window passes and the combine kernel
    keeps the smaller exact distance, the lower column on a tie (np.argmin's first minimum, MotifSeq.py:437-439).  int16
    rows of 4 000 and 20 000 samples: identical copies (a tie: the first wins), three copies (one cluster too many: exact
    pass), copies 400 columns apart (one window).  float64 reads, where a copy can be worse by less than the screening's
    2 E: copies whose costs differ by a hair either way (the sibling's record wins, or the read's own).  Against the oracle and
    against SK_DTW_NO_SIBLINGS=1."""
    from squigglekit_amd import api, synth
    L = gpu.load()
    motif = synth.synthetic_motif(200, seed=21)
    copy = np.clip(np.rint(motif * 93.4 + 511.0), 1, 1199).astype(np.int16)
    for M in (4000, 20000):
        R = 640
        sig = synth.squiggle_batch(R, M, 880 + M, motif=None)
        lens = np.full(R, M, dtype=np.int32)
        far = M - 1200
        kinds = {}
        for r in range(0, R, 4):
            k = (r // 4) % 3
            kinds[r] = k
            sig[r, 300:500] = copy
            if k == 2:
                sig[r, 700:900] = copy                       # close together: one window
            else:
                sig[r, far:far + 200] = copy                 # k = 0: two clusters, a tie
            if k == 1:
                sig[r, M // 2:M // 2 + 200] = copy           # three clusters
        got = api.motifseq_batch(sig, lens, motif)
        g = api.last_dtw_guard()
        retries = L.sk_last_dtw_retries()
        monkeypatch.setenv("SK_DTW_NO_SIBLINGS", "1")
        plain = api.motifseq_batch(sig, lens, motif)
        retries_plain = L.sk_last_dtw_retries()
        monkeypatch.delenv("SK_DTW_NO_SIBLINGS")
        assert got.tobytes() == plain.tobytes()
        n0 = sum(1 for k in kinds.values() if k == 0)
        # (the copies' surroundings differ, so not every pair of them is within 2 E of each other: those reads have one cluster)
        assert g["second_windows"] >= n0 // 2 and g["alarm"] == 0, (g, n0)
        assert retries <= retries_plain - g["second_windows"] + 4, (retries, retries_plain, g)
        rows = np.array(sorted(kinds))
        want = ora.motifseq_batch_i16(sig[rows], lens[rows], motif)
        for f in ("dist", "start", "end"):
            assert np.array_equal(got[f][rows], want[f]), (M, f)
        k_of = np.array([kinds[r] for r in rows])
        assert np.mean(got["end"][rows][k_of == 0] < 600) > 0.6      # a tie goes to the first copy (some reads match better elsewhere)
    # float64 reads: the second copy better / worse than the first by a hair (0.02 raw units on one sample: 2e-4 signal units)
    R, M = 400, 4000
    base = synth.squiggle_batch(R, M, 991, motif=None).astype(np.float64)
    reads, kind = [], []
    for r in range(R):
        x = base[r].copy()
        k = r % 4
        if k < 2:
            x[300:500] = copy
            x[M - 1200:M - 1000] = copy
            x[(300 if k == 0 else M - 1200) + 77] += 0.02    # k = 0: the first copy is the worse one, k = 1: the second
        kind.append(k)
        reads.append(x)
    got = api.motifseq_reads_f64(reads, motif)
    g = api.last_dtw_guard()
    monkeypatch.setenv("SK_DTW_NO_SIBLINGS", "1")
    plain = api.motifseq_reads_f64(reads, motif)
    monkeypatch.delenv("SK_DTW_NO_SIBLINGS")
    assert got.tobytes() == plain.tobytes() and g["second_windows"] >= 100 and g["alarm"] == 0, g
    kind = np.array(kind)
    two = got["end"][kind < 2]
    # both outcomes occur: the first cluster holds the minimum, the second one does (the sibling's record replaces the read's)
    assert (two < 600).sum() >= 10 and (two > M - 1200).sum() >= 5, (int((two < 600).sum()), int((two > M - 1200).sum()))
    for r in np.flatnonzero(kind < 2):
        f = ora.scale_outliers(reads[r], 0, 1200)
        assert (got["dist"][r], got["start"][r], got["end"][r]) == ora.dtw_subsequence(motif, ora.medmad(f)[0]), r
