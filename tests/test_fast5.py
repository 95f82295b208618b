"""fast5 input without h5py (segmenter.py:321-355, MotifSeq.py:327-350).

CPU: the built-in HDF5 reader decodes the reference's example read (tests/golden/example_test.fast5, a data file)
to exactly the samples its BLOW5 copy holds, and reports the attributes the scripts use; the CLIs' -i / -p / -f
branches reproduce what the reference's main() printed (goldens: tools/gen_golden_fast5.py), with the GPU entry
points answered by the oracle.  GPU: the same replay through the HIP path."""
import os
import re
import shutil

import numpy as np
import pytest

from conftest import GOLD, load_golden
from test_cli import oracle_backend, run_cli, scrappy_stub      # noqa: F401  (fixtures)

F5 = os.path.join(GOLD, "example_test.fast5")


def test_hdf5min_decodes_the_example_read(example_read):
    from squigglekit_amd import hdf5min, tsvio
    with hdf5min.File(F5) as f:
        assert sorted(f.keys()) == ["Analyses", "Raw", "UniqueGlobalKey"]
        name = list(f["Raw/Reads"].keys())[0]
        read = f["Raw/Reads"][name]
        sig = read["Signal"][()]
        assert sig.dtype == np.int16 and np.array_equal(sig, example_read["signal"])      # == the BLOW5 payload
        assert read.attrs["read_id"] == example_read["read_id"].encode()
        ch = f["UniqueGlobalKey/channel_id"].attrs
        assert (ch["digitisation"], ch["offset"], ch["sampling_rate"]) == (8192.0, 16.0, 4000.0)
        assert abs(ch["range"] - example_read["range"]) < 1e-9
        assert f["Analyses/Segmentation_000/Summary/segmentation"].attrs["first_sample_template"] == 518
        with pytest.raises(hdf5min.Hdf5Unsupported):
            f["Analyses/Basecall_1D_000/BaseCalled_template/Fastq"][()]                   # a string dataset
        with pytest.raises(KeyError):
            f["Raw/Nope"]
    sig2, rid = tsvio.read_single_fast5(F5, raw_signal=True)
    assert np.array_equal(sig2, example_read["signal"]) and rid == example_read["read_id"]
    pa, _ = tsvio.read_single_fast5(F5, raw_signal=False)
    from squigglekit_amd.blow5 import to_pA
    assert np.array_equal(pa, to_pA(example_read["signal"], 8192.0, 16.0, float("{0:.2f}".format(example_read["range"]))))


def test_hdf5min_rejects_non_hdf5(tmp_path):
    from squigglekit_amd import hdf5min
    p = tmp_path / "x.fast5"
    p.write_bytes(b"nothing to see\n" * 100)
    with pytest.raises(hdf5min.Hdf5Error):
        hdf5min.File(str(p))


def _layout(tmp_path):
    d = tmp_path / "reads" / "sub"
    d.mkdir(parents=True)
    shutil.copyfile(F5, d / "test.fast5")
    (tmp_path / "bad").mkdir()
    (tmp_path / "bad" / "broken.fast5").write_bytes(b"this is not an HDF5 file\n" * 40)
    (tmp_path / "list.txt").write_text("%s\t9.3\n%s\t1.0\n" % (d / "test.fast5", tmp_path / "bad" / "broken.fast5"))


def _no_traceback(s):
    """Tracebacks name the files and lines of whoever raised (the reference's there, ours here): drop them."""
    return re.sub(r"Traceback \(most recent call last\):\n(?:  .*\n)*\S.*\n", "", s)


def _replay(tmp_path):
    from squigglekit_amd.motifseq_cli import main as mot_main
    from squigglekit_amd.segmenter_cli import main as seg_main
    _layout(tmp_path)
    fa = os.path.join(GOLD, "CATCTATCCAGGGTTAAATT.fa")
    gold = load_golden("fast5_cli.json.gz")
    n = 0
    for run in gold["runs"]:
        argv = [a.replace("<TMP>", str(tmp_path)).replace("<FA>", fa) for a in run["argv"]]
        so, se, code = run_cli(seg_main if run["tool"] == "segmenter" else mot_main, argv)
        so, se = so.replace(str(tmp_path), "<TMP>"), se.replace(str(tmp_path), "<TMP>")
        assert so == run["stdout"], (run["argv"], so[-300:], run["stdout"][-300:])
        assert code == run["exit"]
        assert _no_traceback(se) == _no_traceback(run["stderr"]), (run["argv"], se[-400:], run["stderr"][-400:])
        n += 1
    assert n == 8


def test_fast5_cli_branches_cpu(oracle_backend, scrappy_stub, tmp_path):    # noqa: F811
    _replay(tmp_path)


@pytest.mark.gpu
def test_fast5_cli_branches_gpu(gpu, scrappy_stub, tmp_path):               # noqa: F811
    _replay(tmp_path)


def _replay_multi():
    """The multi-read branch (segmenter.py:233-260, 358-396; tsvio.read_multi_fast5 -> segmenter_cli): the reference's
    main() on tests/golden/multi_two_reads.fast5 (two reads with different channel constants, deflate-compressed
    signals; laid out by tools/hdf5_write_min.py, goldens by tools/gen_golden_fast5.py:multi_read)."""
    from squigglekit_amd.segmenter_cli import main as seg_main
    path = os.path.join(GOLD, "multi_two_reads.fast5")
    gold = load_golden("fast5_multi_cli.json.gz")
    for run in gold["runs"]:
        so, se, code = run_cli(seg_main, [a.replace("<F5>", path) for a in run["argv"]])
        assert so.replace(path, "<F5>") == run["stdout"], (run["argv"], so[-300:], run["stdout"][-300:])
        assert code == run["exit"] and _no_traceback(se.replace(path, "<F5>")) == _no_traceback(run["stderr"]), run["argv"]
    assert len(gold["runs"]) == 5 and gold["runs"][0]["stdout"].count("\n") == 2


def test_multi_read_fast5_reader():
    """hdf5min on the multi-read fixture: both groups, their attributes, the signals byte-equal to the stretches of the
    example read they were cut from."""
    from squigglekit_amd import hdf5min, tsvio
    with hdf5min.File(os.path.join(GOLD, "example_test.fast5")) as f:
        name = list(f["Raw/Reads"].keys())[0]
        sig = f["Raw/Reads"][name]["Signal"][()]
    with hdf5min.File(os.path.join(GOLD, "multi_two_reads.fast5")) as f:
        keys = list(f.keys())
        assert keys == ["read_0a1b2c3d-aaaa-4bbb-8ccc-000000000001", "read_0a1b2c3d-aaaa-4bbb-8ccc-000000000002"]
        assert np.array_equal(f[keys[0]]["Raw/Signal"][()], sig[:9000])
        assert np.array_equal(f[keys[1]]["Raw/Signal"][()], sig[14000:26000])
        assert f[keys[1]]["Raw"].attrs["read_id"].decode() == keys[1][5:]
        assert f[keys[1]]["channel_id"].attrs["offset"] == f[keys[0]]["channel_id"].attrs["offset"] + 7.0
    raw = tsvio.read_multi_fast5(os.path.join(GOLD, "multi_two_reads.fast5"), True)
    assert list(raw) == keys and np.array_equal(raw[keys[0]], sig[:9000])


def test_multi_read_fast5_cli_cpu(oracle_backend):                          # noqa: F811
    _replay_multi()


@pytest.mark.gpu
def test_multi_read_fast5_cli_gpu(gpu):
    _replay_multi()
