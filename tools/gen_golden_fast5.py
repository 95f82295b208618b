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


if __name__ == "__main__":
    main()
