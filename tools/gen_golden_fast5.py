#!/usr/bin/env python3
"""Goldens for the fast5 branches of the two CLIs (run in the container that has /root/reference).

    python tools/gen_golden_fast5.py

Copies the reference's example read (example/test.fast5, a data file) to tests/golden/example_test.fast5 and runs
the reference's own main() on it -- segmenter.py -i/-p (--single), MotifSeq.py -f/-p -- with `h5py` stood in by
squigglekit_amd.hdf5min (h5py itself is not installable here).  These goldens therefore pin the reference's code
AROUND the HDF5 access (pA conversion and rounding, [:Num], messages, the b'...' read id MotifSeq prints); the
HDF5 decoding itself is pinned separately by byte-equality with the BLOW5 copy of the same read
(tests/test_fast5.py).  mlpy / scrappy are stubbed exactly as in tools/gen_golden.py (DTW digits = the oracle's).
Outputs only -- no reference source is stored."""
import gzip
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden as gg                                   # noqa: E402
from squigglekit_amd import hdf5min                       # noqa: E402

REF, GOLD = gg.REF, gg.GOLD


def main():
    h5 = types.ModuleType("h5py")
    h5.File = lambda path, mode="r": hdf5min.File(path)
    seg, mot, _ = gg.import_reference()
    sys.modules["h5py"] = h5
    seg.h5py = h5
    mot.h5py = h5
    shutil.copyfile(os.path.join(REF, "example", "test.fast5"), os.path.join(GOLD, "example_test.fast5"))
    tmp = tempfile.mkdtemp()
    d = os.path.join(tmp, "reads", "sub")
    os.makedirs(d)
    f5 = os.path.join(d, "test.fast5")
    shutil.copyfile(os.path.join(REF, "example", "test.fast5"), f5)
    os.makedirs(os.path.join(tmp, "bad"))
    bad = os.path.join(tmp, "bad", "broken.fast5")       # (kept out of the -p directory: the reference's -p branch
    with open(bad, "wb") as fh:                           #  crashes on `[].any()` when a file cannot be read)
        fh.write(b"this is not an HDF5 file\n" * 40)
    lst = os.path.join(tmp, "list.txt")
    with open(lst, "w") as fh:
        fh.write(f5 + "\t9.3\n" + bad + "\t1.0\n")
    fa = os.path.join(REF, "example", "CATCTATCCAGGGTTAAATT.fa")
    runs = []

    def rel(text):
        return text.replace(tmp, "<TMP>").replace(fa, "<FA>")

    for argv in (["-i", f5, "--single"], ["-i", f5, "--single", "--raw_signal"], ["-i", f5, "--single", "-n", "6000"],
                 ["-i", f5, "--single", "-ku", "-j", "100"], ["-p", os.path.join(tmp, "reads"), "--single"]):
        # (no unreadable file here: the reference's segmenter goes on to get_segs([]) and dies in sig.min())
        so, se, code = gg.run_main(seg, ["segmenter.py"] + argv)
        runs.append({"tool": "segmenter", "argv": [rel(a) for a in argv], "stdout": rel(so), "stderr": rel(se), "exit": code})
    for argv in (["-f", lst, "-i", fa], ["-f", lst, "-i", fa, "-l", "zscale"], ["-p", os.path.join(tmp, "reads"), "-i", fa]):
        so, se, code = gg.run_main(mot, ["MotifSeq.py"] + argv)
        runs.append({"tool": "motifseq", "argv": [rel(a) for a in argv], "stdout": rel(so), "stderr": rel(se), "exit": code})
    with gzip.open(os.path.join(GOLD, "fast5_cli.json.gz"), "wt") as fh:
        json.dump({"generator": "tools/gen_golden_fast5.py: /root/reference segmenter.py / MotifSeq.py main() on "
                                "example/test.fast5; h5py stood in by squigglekit_amd.hdf5min, mlpy by the oracle",
                   "layout": "<TMP>/reads/sub/test.fast5, <TMP>/bad/broken.fast5, <TMP>/list.txt lists both", "runs": runs}, fh, indent=1)
    shutil.rmtree(tmp)
    for r in runs:
        print(r["tool"], r["argv"], "->", repr(r["stdout"][:90]), "| stderr tail:", repr(r["stderr"][-90:]))
    multi_read(seg)


def multi_read(seg):
    """The multi-read branch (segmenter.py:233-260, 358-396: no --single): the reference ships no multi-read file, so a
    fixture is LAID OUT here (tools/hdf5_write_min.py: classic HDF5, deflate-compressed Signal datasets, the attributes
    the reference reads) from two stretches of the reference's own example read with its channel constants, and the
    reference's main() is run on it."""
    from hdf5_write_min import write_hdf5
    with hdf5min.File(os.path.join(GOLD, "example_test.fast5")) as f:
        name = list(f["Raw/Reads"].keys())[0]
        sig = f["Raw/Reads"][name]["Signal"][()]
        ch = {k: float(v) for k, v in f["UniqueGlobalKey/channel_id"].attrs.items()
              if k in ("digitisation", "offset", "range", "sampling_rate")}
    chan = {"@" + k: v for k, v in ch.items()}
    chan2 = dict(chan, **{"@offset": ch["offset"] + 7.0, "@range": ch["range"] * 1.015})     # (per-read constants differ)
    tree = {"read_0a1b2c3d-aaaa-4bbb-8ccc-000000000001": {"Raw": {"@read_id": b"0a1b2c3d-aaaa-4bbb-8ccc-000000000001",
                                                                  "Signal": np.ascontiguousarray(sig[:9000])},
                                                          "channel_id": chan},
            "read_0a1b2c3d-aaaa-4bbb-8ccc-000000000002": {"Raw": {"@read_id": b"0a1b2c3d-aaaa-4bbb-8ccc-000000000002",
                                                                  "Signal": np.ascontiguousarray(sig[14000:26000])},
                                                          "channel_id": chan2}}
    path = os.path.join(GOLD, "multi_two_reads.fast5")
    write_hdf5(path, tree)
    runs = []
    for argv in (["-i", path], ["-i", path, "--raw_signal"], ["-i", path, "-n", "6000"], ["-i", path, "-ku", "-j", "100"],
                 ["-i", path, "-w", "60", "-e", "9"]):
        so, se, code = gg.run_main(seg, ["segmenter.py"] + argv)
        runs.append({"tool": "segmenter", "argv": [a.replace(path, "<F5>") for a in argv], "stdout": so.replace(path, "<F5>"),
                     "stderr": se.replace(path, "<F5>"), "exit": code})
        print("multi", argv[1:], "->", repr(so[:120]), "| stderr tail:", repr(se[-80:]))
    with gzip.open(os.path.join(GOLD, "fast5_multi_cli.json.gz"), "wt") as fh:
        json.dump({"generator": "tools/gen_golden_fast5.py:multi_read -- /root/reference segmenter.py main() (multi-read branch) on "
                                "tests/golden/multi_two_reads.fast5 (two stretches of example/test.fast5's signal, laid out by "
                                "tools/hdf5_write_min.py); h5py stood in by squigglekit_amd.hdf5min", "runs": runs}, fh, indent=1)


if __name__ == "__main__":
    main()
