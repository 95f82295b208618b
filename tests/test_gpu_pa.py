"""GPU: raw int16 reads through the pA route IN THE RAW DOMAIN (round 6; csrc/sk_segstat.hip k_seg_stats<.., PA>).

segmenter.py:345-349 / :366-370 turn every fast5 / slow5 read into np.round((raw + offset) * (range / digitisation), 2)
before scale_outliers (:311-318) and get_segs (:399-470) see it.  That is a monotone map of the int16 sample, so the
library finds the filter's limits, np.median, np.std (exact integer centi-pA sums off the value histogram) and the two
thresholds on the samples themselves and certifies them against numpy's rounding; what it cannot certify -- thresholds
too close to a grid point, a spike above the histogram window, a calibration outside the plain range -- is redone from
the float64 values in numpy's order.  Every record here is compared with the oracle run on the float64 values numpy
makes the reference's way; bit-exact segment boundaries."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MINION = (8192.0, 16.0, 1493.94)
PROMETHION = (2048.0, -237.0, 748.58)


def to_pa(raw, dig, off, rng):
    """segmenter.py:345-349, :385: the range cut to two decimals, np.round(.., 2)"""
    return np.round((raw.astype(np.int64) + off) * (float("{0:.2f}".format(rng)) / dig), 2)


def want_segs(ora, raw, lens, calib, params=None):
    from squigglekit_amd._lib import SegParams
    p = params or SegParams()
    op = ora.SegParams(p.error, p.corrector, p.window, p.seg_dist, p.std_scale, p.stall_len)
    out = []
    for r in range(raw.shape[0]):
        pa = to_pa(raw[r, :lens[r]], *calib[r])
        f = ora.scale_outliers(pa, p.lim_low, p.lim_hi)
        out.append((ora.get_segs(f, op) or []) if f.size else [])
    return out


def check(ora, api, raw, lens, calib, params=None, expect_raw=True):
    calib = np.asarray(calib, dtype=np.float64).reshape(-1, 3)
    if calib.shape[0] == 1:
        calib = np.tile(calib, (raw.shape[0], 1))
    segs, nsegs = api.segment_batch_pa(raw, lens, calib, params)
    retries = api.last_pa_retries()
    assert (retries >= 0) == expect_raw, retries
    want = want_segs(ora, raw, lens, calib, params)
    for r in range(raw.shape[0]):
        assert segs[r, :nsegs[r]].tolist() == want[r], (r, calib[r].tolist(), int(lens[r]))
    return retries, int(nsegs.sum())


def squiggles(R, M, seed, base=0, scale=1.0):
    from squigglekit_amd import synth
    x = synth.squiggle_batch(R, M, seed).astype(np.float64)
    return np.clip(np.rint(x * scale + base), -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("M", [1000, 2048, 4000, 4096, 4100, 16384, 19999, 36977, 50000, 65536, 70000])
@pytest.mark.parametrize("wg", ["default", "SK_SEG_WG_ALL", "SK_SEG_NO_WG"])
def test_raw_domain_pa_matches_float64_oracle(gpu, ora, monkeypatch, M, wg):
    """MinION and PromethION channel constants at every kernel shape (2 / 4 / 8 tiles, reads longer than a window),
    ragged lengths: no read takes the redo, every boundary equals the oracle's on the float64 values."""
    from squigglekit_amd import api
    if wg != "default":
        if M <= 4096:
            pytest.skip("the switch concerns rows beyond 4 096 samples")
        monkeypatch.setenv(wg, "1")                  # statistics kernel: workgroup per read (one look) / wavefront per read
    R = 96 if M > 5000 else 320
    rng = np.random.default_rng(M)
    S = (M + 7) & ~7                                  # (rows as the readers lay them out: a multiple of 8 samples)
    raw = squiggles(R, S, 1000 + M)
    lens = rng.integers(max(1, M // 3), M + 1, R).astype(np.int32)
    lens[:4] = [M, M, 1, 0]
    retries, total = check(ora, api, raw, lens, MINION)
    assert retries <= 1 and total > R // 4           # (the one-sample read: std 0, never certifiable)
    # PromethION: negative offset, a coarser unit -- the raw window starts at 237
    raw2 = squiggles(R, S, 2000 + M, base=237, scale=0.6)
    retries, total = check(ora, api, raw2, lens, PROMETHION)
    assert retries <= 1 and total > R // 4


def test_per_read_calibrations_and_odd_constants(gpu, ora):
    """every read its own constants: ranges whose two-decimal cut matters, float offsets, units from 1/64 to 1.25"""
    from squigglekit_amd import api
    R, M = 400, 4000
    rng = np.random.default_rng(7)
    raw = squiggles(R, M, 77)
    lens = rng.integers(M // 2, M + 1, R).astype(np.int32)
    calib = np.empty((R, 3))
    calib[:, 0] = rng.choice([8192.0, 2048.0, 4096.0], R)
    calib[:, 1] = np.round(rng.uniform(-40, 40, R), 1)
    calib[:, 2] = rng.uniform(600, 1600, R)
    calib[:8] = [[8192.0, 10.0, 1467.6149], [2048.0, -3.5, 748.58496], [8192.0, 0.0, 1200.005], [8192.0, 16.0, 128.0],
                 [100.0, 3.0, 125.0], [8192.0, 0.25, 1493.94], [1.0, 0.0, 0.18], [8192.0, -900.0, 1493.94]]
    retries, total = check(ora, api, raw, lens, calib)
    assert retries < R // 8 and total > 50          # (a few reads may sit above a coarse unit's window: redone)


def test_spikes_constant_reads_and_calibrations_outside_the_plain_range_take_the_redo(gpu, ora):
    from squigglekit_amd import api
    R, M = 64, 4000
    raw = squiggles(R, M, 5)
    lens = np.full(R, M, dtype=np.int32)
    calib = np.tile(np.array(MINION), (R, 1))
    raw[0, 100] = 3000                               # a kept spike above the 2 047-unit window (546 pA < 900)
    raw[1, 100] = 6000                               # a dropped one (1 093 pA): the maximum cannot tell -> redo, still right
    raw[2, :] = 500                                  # constant: std is whatever numpy's rounding makes of it
    raw[3, :] = -20                                  # nothing survives the filter
    raw[4, ::2] = 30000                              # half of the read dropped
    calib[5] = [8192.0, 16.0, 8192.0 * 5]            # unit 5: outside the kernel's range
    calib[6] = [8192.0, 16.0, -1493.94]              # negative unit: decreasing map
    calib[7] = [8192.0, np.nan, 1493.94]             # NaN offset: every value NaN, nothing kept
    calib[8] = [8192.0, 1e9, 1493.94]                # huge offset
    calib[9] = [8192.0, 16.0, 8192.0 * 0.01]         # unit 0.01: a window of 20 pA
    retries, total = check(ora, api, raw, lens, calib)
    assert 6 <= retries <= 12 and total > 10


def test_forced_redo_parameter_corners_and_the_float64_switch(gpu, ora, monkeypatch):
    """every read through the numpy-order redo (SK_SEG_DELTA_SCALE), other limits / scales / windows, and the A/B
    switch that expands to float64: the same records"""
    from squigglekit_amd import api
    from squigglekit_amd._lib import SegParams
    R, M = 200, 6000
    rng = np.random.default_rng(11)
    raw = squiggles(R, M, 314)
    lens = rng.integers(M // 2, M + 1, R).astype(np.int32)
    plain = api.segment_batch_pa(raw, lens, np.tile(np.array(MINION), (R, 1)))
    monkeypatch.setenv("SK_SEG_DELTA_SCALE", "1e13")
    retries, _ = check(ora, api, raw, lens, MINION)
    assert retries >= R - 4                          # (empty reads are not listed)
    monkeypatch.delenv("SK_SEG_DELTA_SCALE")
    monkeypatch.setenv("SK_SEG_PA_F64", "1")
    f64 = api.segment_batch_pa(raw, lens, np.tile(np.array(MINION), (R, 1)))
    assert api.last_pa_retries() == -1
    monkeypatch.delenv("SK_SEG_PA_F64")
    assert np.array_equal(plain[1], f64[1]) and np.array_equal(plain[0], f64[0])
    for kw in (dict(lim_low=60, lim_hi=140), dict(lim_low=-50, lim_hi=2000), dict(std_scale=0.3), dict(std_scale=2.5),
               dict(window=40, error=2), dict(error=40, corrector=10), dict(lim_low=95, lim_hi=97), dict(std_scale=-0.5)):
        p = SegParams()
        for k, v in kw.items():
            setattr(p, k, v)
        check(ora, api, raw[:64], lens[:64], MINION, p)
    # rows the streaming kernel does not take (stride not a multiple of 8): the float64 image, as before round 6
    check(ora, api, np.ascontiguousarray(raw[:32, :5003]), np.minimum(lens[:32], 5003), MINION, expect_raw=False)


def test_example_read_through_the_raw_domain(gpu, ora, example_read):
    """the reference's example read (36 978 samples) with the constants its BLOW5 record carries"""
    from squigglekit_amd import api
    sig = np.asarray(example_read["signal"], dtype=np.int16)
    M = (sig.size + 7) & ~7
    raw = np.zeros((3, M), dtype=np.int16)
    raw[:, :sig.size] = sig
    lens = np.array([sig.size, 20000, 4096], dtype=np.int32)
    calib = [[example_read["digitisation"], example_read["offset"], example_read["range"]]] * 3
    retries, total = check(ora, api, raw, lens, calib)
    assert retries == 0 and total >= 1


def test_centi_unit_batches_equal_the_float64_ones(gpu, ora):
    """pA TSV lines as int32 centi-units (tsvio.FloatBlock.centi, sk_tsv_parse_centi): the device makes c / 100.0 --
    float("ddd.dd") bit for bit (segmenter.py:198-199, MotifSeq.py:270) -- so both tools' ragged batch calls must give
    the records of the float64 batch of the same tokens, and the oracle's."""
    from squigglekit_amd import api, synth
    rng = np.random.default_rng(3)
    motif = synth.synthetic_motif(120)
    raw = synth.squiggle_batch(300, 5000, 808, motif=motif)
    lens = rng.integers(1, 5001, 300)
    lens[:3] = [5000, 1, 4096]
    centi = [np.rint((raw[r, :lens[r]].astype(np.int64) + 16.0) * (1493.94 / 8192.0) * 100).astype(np.int32) for r in range(300)]
    centi[5][:] = 0                                      # nothing survives the filter
    centi[6][::3] = -250                                 # negative values
    centi[7][:] = 12345                                  # constant
    flat = np.concatenate(centi)
    off = np.concatenate([[0], np.cumsum([c.size for c in centi])]).astype(np.int64)
    vals = flat / 100.0
    cut = np.minimum(lens, 3000).astype(np.int32)
    a = api.segment_ragged_f64(flat, off, cut)
    b = api.segment_ragged_f64(vals, off, cut)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[1].sum() > 50
    for r in range(0, 300, 7):
        f = ora.scale_outliers(vals[off[r]:off[r] + cut[r]], 0, 900)
        assert a[0][r, :a[1][r]].tolist() == ((ora.get_segs(f) or []) if f.size else []), r
    for scale in ("medmad", "zscale"):
        ha = api.motifseq_multi_ragged_f64(flat, off, [motif, motif[10:90]], scale=scale)
        hb = api.motifseq_multi_ragged_f64(vals, off, [motif, motif[10:90]], scale=scale)
        assert [h.tobytes() for h in ha] == [h.tobytes() for h in hb]
