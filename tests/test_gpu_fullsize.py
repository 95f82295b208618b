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
    # oracle on the subset: bit-identical
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
    sub = rng.choice(R, 400, replace=False)
    osegs, onsegs = ora.segment_batch_i16(sig[sub], lens[:400], max_segs=segs.shape[1])
    assert np.array_equal(onsegs, nsegs[sub])
    for k, r in enumerate(sub):
        assert np.array_equal(osegs[k, :onsegs[k]], segs[r, :nsegs[r]])
    # checksum of checksums, stable across runs of the same seed (regression anchor)
    again, nagain = api.segment_batch(sig, lens)
    assert np.array_equal(again, segs) and np.array_equal(nagain, nsegs)
