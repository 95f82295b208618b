"""CPU: host-side logic that needs no GPU -- packing, format readers, synthetic data."""
import io
import types

import numpy as np

from conftest import load_golden


def test_pack_and_int16_detection():
    from squigglekit_amd import api
    buf, lens = api.pack_i16([np.arange(5), np.arange(17), np.zeros(0, dtype=int)])
    assert buf.shape == (3, 24) and buf.dtype == np.int16 and lens.tolist() == [5, 17, 0]
    assert buf[1, :17].tolist() == list(range(17)) and not buf[0, 5:].any()
    assert api.is_int16_exact(np.array([1.0, 2.0, -3.0]))
    assert not api.is_int16_exact(np.array([1.5]))
    assert not api.is_int16_exact(np.array([40000.0]))
    assert api.is_int16_exact(np.array([], dtype=float))
    for good in ([1.0, 2.0, -3.0], [], [32767, -32768], np.array([5, 6], dtype=np.int64)):
        got = api.as_int16_exact(np.array(good))
        assert got is not None and got.dtype == np.int16 and np.array_equal(got, np.array(good))
    for bad in ([1.5], [40000.0], [np.nan, 1.0], [np.inf], [-32769], [65536 + 7]):
        assert api.as_int16_exact(np.array(bad)) is None


def test_blow5_reader_matches_survey_facts(example_read):
    from squigglekit_amd.blow5 import to_pA
    s = example_read["signal"]
    assert example_read["read_id"] == "db4ae416-40c2-45c2-9cc9-7d49c5711a7c"
    assert (s.size, int(s.min()), int(s.max()), float(np.median(s))) == (36978, -2, 1060, 511.0)
    assert (example_read["digitisation"], example_read["offset"], example_read["sampling_rate"]) == (8192.0, 16.0, 4000.0)
    pa = to_pA(s, example_read["digitisation"], example_read["offset"], example_read["range"])
    assert float(np.median(pa[:-1])) == 96.11


def test_synth_is_deterministic_and_shaped():
    from squigglekit_amd import synth
    a = synth.squiggle_batch(32, 4000, 1)
    b = synth.squiggle_batch(32, 4000, 1)
    assert np.array_equal(a, b) and a.dtype == np.int16 and a.shape == (32, 4000)
    assert not np.array_equal(a, synth.squiggle_batch(32, 4000, 2))
    assert 450 < a.mean() < 560 and 50 < a.std() < 120
    m = synth.synthetic_motif(200)
    assert m.shape == (200,) and np.array_equal(m, synth.synthetic_motif(200))


def test_last_row_cost_view():
    from squigglekit_amd.api import LastRowCost
    c = LastRowCost(np.arange(5.0), 3)
    assert c.shape == (3, 5)
    assert np.array_equal(c[-1, :], np.arange(5.0)) and np.array_equal(c[-1,], np.arange(5.0))
    assert np.array_equal(c[2, 1:3], [1.0, 2.0])
    try:
        c[0, :]
    except IndexError:
        pass
    else:
        raise AssertionError("rows other than the last are not kept")


def test_test_segs_mirror_matches_reference_messages():
    """api.test_segs reproduces segmenter.test_segs (segmenter.py:473-494) incl. its stderr text."""
    from squigglekit_amd import api
    a = types.SimpleNamespace(stall=True, gap=False, stall_start=300, gap_dist=3000)
    err = io.StringIO()
    assert api.test_segs([[400, 500]], a, err) is False and err.getvalue() == "start seg too late!"
    a = types.SimpleNamespace(stall=True, gap=True, stall_start=300, gap_dist=100)
    err = io.StringIO()
    assert api.test_segs([[0, 463], [1494, 1835]], a, err) is False and err.getvalue() == "second seg too far!"
    err = io.StringIO()
    segs = [[0, 463]]
    assert api.test_segs(segs, a, err) is segs                 # IndexError swallowed, read passes
    assert err.getvalue().startswith("something went wrong test_segs()")
    # same verdicts the reference reached on the real read (golden CLI run)
    runs = load_golden("segmenter_cli.json.gz")["runs"]
    r = [x for x in runs if x["tsv"] == "raw_noinfo" and x["flags"] == ["-k", "-g", "-u", "-b", "100"]][0]
    assert "second seg too far!" in r["stderr"] and r["stdout"] == ""
