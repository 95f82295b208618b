"""GPU parity for the layout corners of the filter + statistics kernel: row strides that are not
a multiple of 8 samples (scalar loads, 2-byte stores), reads without a single outlier (every
wavefront takes the packed all-survive path, 16-byte stores), and outliers placed on the tile /
wavefront / vector boundaries so that each store alignment (16, 4, 2 bytes) is exercised."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reads(R, M, seed, clean):
    from squigglekit_amd import synth
    sig = synth.squiggle_batch(R, M, seed)
    if clean:
        sig = np.clip(sig, 350, 650).astype(np.int16)         # nothing for either filter to reject
    return np.ascontiguousarray(sig)


def _poke(sig):
    """Outliers at chosen raw positions, a different pattern per read."""
    R, M = sig.shape
    spots = [0, 1, 7, 8, 9, 15, 16, 511, 512, 513, 1023, 1024, 2047, 2048, 2049, M - 2, M - 1]
    for r in range(R):
        chosen = [p for i, p in enumerate(spots) if (r >> (i % 5)) & 1 and 0 <= p < M]
        if r % 7 == 3:
            chosen += list(range(100, 100 + (r % 23)))          # a run of rejects: odd and even shifts
        sig[r, chosen] = -7 if r % 2 else 2000
    return sig


@pytest.mark.parametrize("M", [4000, 4001, 2047, 4093, 520])
@pytest.mark.parametrize("clean", [True, False])
def test_segmenter_layout_corners(gpu, ora, M, clean):
    from squigglekit_amd import api
    sig = _reads(96, M, 4100 + M, clean)
    if not clean:
        sig = _poke(sig)
    lens = np.full(96, M, dtype=np.int32)
    lens[5] = M - 1; lens[6] = M - 7; lens[7] = 9; lens[8] = 8
    segs, nsegs = api.segment_batch(sig, lens)
    osegs, onsegs = ora.segment_batch_i16(sig, lens, max_segs=segs.shape[1])
    assert np.array_equal(nsegs, onsegs), np.nonzero(nsegs != onsegs)[0][:8]
    for r in range(96):
        assert np.array_equal(segs[r, :nsegs[r]], osegs[r, :nsegs[r]]), r


@pytest.mark.parametrize("M", [4000, 4001, 2047, 520])
@pytest.mark.parametrize("clean", [True, False])
@pytest.mark.parametrize("scale", ["medmad", "zscale", "medmad-workgroup"])
def test_motifseq_layout_corners(gpu, ora, example_model, M, clean, scale, monkeypatch):
    from squigglekit_amd import api
    if scale == "medmad-workgroup":         # medmad on the workgroup-per-read kernel instead of the wave-per-read one
        monkeypatch.setenv("SK_PREP_BLOCK", "1")
        scale = "medmad"
    sig = _reads(64, M, 5200 + M, clean)
    if not clean:
        sig = _poke(sig)
    lens = np.full(64, M, dtype=np.int32)
    lens[3] = M - 3; lens[4] = 300; lens[5] = 8
    got = api.motifseq_batch(sig, lens, example_model, scale=scale)
    want = ora.motifseq_batch_i16(sig, lens, example_model, scale_mode={"medmad": 0, "zscale": 1}[scale])
    assert np.array_equal(got["n"], want["n"])
    ok = (got["flags"] & 2) == 0             # MAD == 0: the reference divides by zero (flagged, not compared)
    assert ok.sum() >= 60
    assert np.array_equal(got["start"][ok], want["start"][ok]) and np.array_equal(got["end"][ok], want["end"][ok])
    nan = np.isnan(got["dist"]) & np.isnan(want["dist"])
    assert np.all(((got["dist"] == want["dist"]) | nan)[ok])


def test_normalised_signal_matches_oracle_on_odd_stride(gpu, ora):
    """The compacted + normalised signal itself (what -x prints), odd stride, shifted alignments."""
    from squigglekit_amd import api
    sig = _poke(_reads(16, 1237, 77, clean=False))
    for r in range(16):
        f = ora.scale_outliers(sig[r], 0, 1200)
        for scale in ("medmad", "zscale"):
            got = api.normalise(sig[r], scale=scale)
            want = ora.medmad(f)[0] if scale == "medmad" else ora.zscale(f)[0]
            assert got.shape == want.shape and np.array_equal(got, want), (r, scale)


@pytest.mark.parametrize("lo,hi", [(0, 900), (300, 700), (-100, 1900), (0, 5000), (-2000, 30000), (-32768, 32767), (499, 501)])
def test_motifseq_outlier_limits(gpu, ora, example_model, lo, hi):
    """-scale_low / -scale_hi choose the histogram size, hence the kernel variant (16, 20 or 32 bins per
    lane on the wave-per-read kernel; the workgroup kernel beyond 2 048 values; the float64 kernels when
    the limits span more values than an LDS histogram holds)."""
    from squigglekit_amd import api
    sig = _poke(_reads(48, 3000, 909, clean=False))
    lens = np.full(48, 3000, dtype=np.int32)
    lens[1] = 17
    got = api.motifseq_batch(sig, lens, example_model, scale_low=lo, scale_hi=hi)   # (widest: float64 kernels)
    want = ora.motifseq_batch_i16(sig, lens, example_model, scale_mode=0, lo=lo, hi=hi)
    assert np.array_equal(got["n"], want["n"])
    ok = (got["flags"] & 2) == 0
    assert np.array_equal(got["start"][ok], want["start"][ok]) and np.array_equal(got["end"][ok], want["end"][ok])
    nan = np.isnan(got["dist"]) & np.isnan(want["dist"])
    assert np.all(((got["dist"] == want["dist"]) | nan)[ok])


def test_segmenter_wide_limits_route_to_f64(gpu, ora):
    from squigglekit_amd import api
    from squigglekit_amd._lib import SegParams
    sig = _poke(_reads(24, 2000, 31, clean=False))
    lens = np.full(24, 2000, dtype=np.int32)
    p = SegParams(lim_low=-32768, lim_hi=32767)
    segs, nsegs = api.segment_batch(sig, lens, p)
    osegs, onsegs = ora.segment_batch_i16(sig, lens, lo=p.lim_low, hi=p.lim_hi, max_segs=segs.shape[1])
    assert np.array_equal(nsegs, onsegs)
    for r in range(24):
        assert np.array_equal(segs[r, :nsegs[r]], osegs[r, :nsegs[r]]), r


def test_pipelined_ingest_sub_batches_and_pinned_buffers(gpu, ora, monkeypatch):
    """The host entry points move big batches in sub-batches (copy of one under the kernels of the previous one).
    SK_INGEST_MB=1 forces many sub-batches on a small batch; pageable and pinned (api.pinned_empty) sources must
    give the same records as the single-shot call and the oracle, retry counts summed over sub-batches."""
    from squigglekit_amd import api, synth
    L = gpu.load()
    motif = synth.synthetic_motif(150, seed=21)
    R, M = 9000, 2000
    sig = synth.squiggle_batch(R, M, 777001, motif=motif)
    lens = np.full(R, M, dtype=np.int32)
    lens[::13] = 1234
    one = api.motifseq_batch(sig, lens, motif)
    segs1, nsegs1 = api.segment_batch(sig, lens - 1)
    monkeypatch.setenv("SK_INGEST_MB", "1")                 # 4 096 reads per sub-batch -> 3 sub-batches
    monkeypatch.setenv("SK_DTW_SPAN", "30")                 # many retries, so the summed count is visible
    ref = api.motifseq_batch(sig[:4096], lens[:4096], motif)
    r_first = L.sk_last_dtw_retries()
    got = api.motifseq_batch(sig, lens, motif)
    r_all = L.sk_last_dtw_retries()
    assert got.tobytes() == one.tobytes() and ref.tobytes() == one[:4096].tobytes()
    assert r_first > 100 and r_all > 1.5 * r_first           # summed over the sub-batches of one call
    monkeypatch.delenv("SK_DTW_SPAN")
    pin = api.pinned_empty(sig.shape, np.int16)
    pin[:] = sig
    got2 = api.motifseq_batch(pin, lens, motif)
    segs2, nsegs2 = api.segment_batch(pin, lens - 1)
    assert got2.tobytes() == one.tobytes()
    assert np.array_equal(segs2, segs1) and np.array_equal(nsegs2, nsegs1)
    want = ora.motifseq_batch_i16(sig[8990:], lens[8990:], motif)
    assert np.array_equal(got["dist"][8990:], want["dist"]) and np.array_equal(got["start"][8990:], want["start"])
    del pin


@pytest.mark.parametrize("stride", [4096, 4000, 3001])
def test_fused_zscale_prologue_numpy_order(gpu, ora, monkeypatch, stride):
    """`-l zscale` on reads of up to 4 096 samples: mean / std are computed by the screening pass's own wavefront (round 5;
    csrc/sk_prepw_dev.h zs_read: numpy's pairwise tree walked per lane) instead of by the workgroup kernel.  Lengths
    around every split of np.add.reduce's tree (< 8: serial; <= 128: one leaf; 129 .. 143: the 64 + rest split; powers
    of two; the full row), reads with and without dropped samples, an aligned and two unaligned strides (3001: the
    element-by-element loads) -- records equal to the unfused path byte for byte, and to the oracle."""
    from squigglekit_amd import api, synth
    motif = synth.synthetic_motif(40, seed=8)
    R = 700
    sig = synth.squiggle_batch(R, stride, 60606 + stride, motif=motif)
    rng = np.random.default_rng(stride)
    lens = rng.integers(200, stride + 1, R).astype(np.int32)
    corner = [0, 1, 5, 7, 8, 9, 15, 16, 17, 63, 64, 65, 127, 128, 129, 130, 136, 137, 143, 144, 145, 255, 256, 257, 511, 512,
              1023, 1024, 1025, 2047, 2048, 2049, stride - 1, stride]
    lens[:len(corner)] = [min(c, stride) for c in corner]
    sig[40:80] = np.clip(sig[40:80], 1, 1199)                       # nothing dropped: the all-kept fast path throughout
    sig[80:90, ::3] = 0                                              # a third of the samples dropped
    sig[90] = 500                                                    # std = 0 -> scale 1
    got = api.motifseq_batch(sig, lens, motif, scale="zscale")
    launches = C.c_int32()
    gpu.load().sk_last_dtw_profile(None, C.byref(launches), None, None, None)
    assert launches.value >= 1, "the batch did not take the screening scheme"
    monkeypatch.setenv("SK_DTW_NOFUSE", "1")
    plain = api.motifseq_batch(sig, lens, motif, scale="zscale")
    monkeypatch.delenv("SK_DTW_NOFUSE")
    assert got.tobytes() == plain.tobytes()
    want = ora.motifseq_batch_i16(sig, lens, motif, scale_mode=1)
    ok = got["n"] > 0
    assert np.array_equal(got["n"], want["n"])
    assert np.array_equal(got["dist"][ok], want["dist"][ok]) and np.array_equal(got["start"][ok], want["start"][ok]) \
        and np.array_equal(got["end"][ok], want["end"][ok])
