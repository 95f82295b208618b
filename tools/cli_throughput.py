#!/usr/bin/env python3
"""End-to-end time of the drop-in command-line tools on a SquigglePull-style raw TSV (text in, text out, process
start included):   python tools/cli_throughput.py [reads=20000] [samples=4000]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from squigglekit_amd import synth                        # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    d = tempfile.mkdtemp()
    model = os.path.join(ROOT, "tests", "golden", "CATCTATCCAGGGTTAAATT.model")
    sig = synth.squiggle_batch(R, M, 4242)
    for name, ncols in (("seg", 4), ("mot", 8)):
        path = os.path.join(d, name + ".tsv")
        with open(path, "w") as fh:
            for r in range(R):
                fh.write("\t".join(["read%d.fast5" % r, "id%d" % r] + ["x"] * (ncols - 2) + [str(v) for v in sig[r].tolist()]) + "\n")
    size = os.path.getsize(os.path.join(d, "seg.tsv")) / 1e6
    for label, cmd in (("segmenter.py -s", [sys.executable, os.path.join(ROOT, "segmenter.py"), "-s", os.path.join(d, "seg.tsv")]),
                       ("MotifSeq.py -s -m", [sys.executable, os.path.join(ROOT, "MotifSeq.py"), "-s", os.path.join(d, "mot.tsv"), "-m", model])):
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            dt = time.perf_counter() - t0
            assert p.returncode == 0
            best = dt if best is None else min(best, dt)
        rows = p.stdout.count(b"\n")
        print("%-20s %d reads x %d samples (%.0f MB of text): %.2f s -> %.0f reads/s, %.0f MB/s of text; %d output lines"
              % (label, R, M, size, best, R / best, size / best, rows))


if __name__ == "__main__":
    main()
