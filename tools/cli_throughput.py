#!/usr/bin/env python3
"""End-to-end time of the drop-in command-line tools, process start included, text out:
    python tools/cli_throughput.py [tsv_reads=200000] [packed_reads=1000000] [samples=4000]
  * SquigglePull-style raw TSV in (what the reference reads)             segmenter.py -s / MotifSeq.py -s -m
  * BLOW5 in (stored records, native decoder)                            --blow5 (segmenter: with --raw_signal)
  * packed int16 .npy in (memory mapped)                                 --i16
Prints one line per run and, last, one JSON object (bench.py's `cli` block runs the same commands)."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from squigglekit_amd import fastio, synth               # noqa: E402


def run(label, cmd, reads, size_mb, out):
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        dt = time.perf_counter() - t0
        assert p.returncode == 0, (label, p.returncode)
        best = dt if best is None else min(best, dt)
    rows = p.stdout.count(b"\n")
    if os.environ.get("SK_CLI_MARKS"):                   # one more run with the tools' stage timestamps shown
        env = dict(os.environ, SK_T0=repr(time.time()))
        q = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
        marks = [ln for ln in q.stderr.decode(errors="replace").splitlines() if ln.startswith("[t+")]
        nshow = int(os.environ["SK_CLI_MARKS"])
        shown = marks if len(marks) <= nshow + 6 else marks[:nshow] + ["[..."] + marks[-6:]
        print("    " + label + " stages: " + " | ".join(m[1:].replace(" s] ", "s ") for m in shown), flush=True)
    print("%-34s %8d reads (%6.0f MB in): %.2f s -> %9.0f reads/s, %5.0f MB/s in; %d output lines"
          % (label, reads, size_mb, best, reads / best, size_mb / best, rows), flush=True)
    out[label] = {"reads": reads, "seconds": best, "reads_per_s": reads / best, "input_mb": size_mb, "output_lines": rows}


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    RP = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
    d = tempfile.mkdtemp()
    model = os.path.join(ROOT, "tests", "golden", "CATCTATCCAGGGTTAAATT.model")
    py = sys.executable
    seg, mot = os.path.join(ROOT, "segmenter.py"), os.path.join(ROOT, "MotifSeq.py")
    out = {}
    if R > 0:
        sig = synth.squiggle_batch(min(R, 2048), M, 4242)          # 2 048 distinct reads, cycled (the tokenizer
        texts = ["\t".join(str(v) for v in row.tolist()) for row in sig]   # does not care; str() of 800 M values would)
        for name, ncols in (("seg", 4), ("mot", 8)):
            with open(os.path.join(d, name + ".tsv"), "w") as fh:
                for r in range(R):
                    fh.write("\t".join(["read%d.fast5" % r, "id%d" % r] + ["x"] * (ncols - 2)) + "\t"
                             + texts[r % len(texts)] + "\n")
        size = os.path.getsize(os.path.join(d, "seg.tsv")) / 1e6
        run("segmenter.py -s (TSV)", [py, seg, "-s", os.path.join(d, "seg.tsv")], R, size, out)
        run("MotifSeq.py -s -m (TSV)", [py, mot, "-s", os.path.join(d, "mot.tsv"), "-m", model], R, size, out)
        for f in ("seg.tsv", "mot.tsv"):
            os.remove(os.path.join(d, f))
        # the same reads as SquigglePull writes them by default: pA values with two decimals (SquigglePull.py:183-189)
        pa = np.round((sig[:256].astype(np.int64) + 16.0) * (1493.94 / 8192.0), 2)
        texts = ["\t".join(repr(float(v)) for v in row) for row in pa]
        Rp = max(1, R // 2)
        for name, ncols in (("seg", 4), ("mot", 8)):
            with open(os.path.join(d, name + "_pa.tsv"), "w") as fh:
                for r in range(Rp):
                    fh.write("\t".join(["read%d.fast5" % r, "id%d" % r] + ["x"] * (ncols - 2)) + "\t"
                             + texts[r % len(texts)] + "\n")
        size = os.path.getsize(os.path.join(d, "seg_pa.tsv")) / 1e6
        run("segmenter.py -s (pA TSV)", [py, seg, "-s", os.path.join(d, "seg_pa.tsv")], Rp, size, out)
        run("MotifSeq.py -s -m (pA TSV)", [py, mot, "-s", os.path.join(d, "mot_pa.tsv"), "-m", model], Rp, size, out)
        for f in ("seg_pa.tsv", "mot_pa.tsv"):
            os.remove(os.path.join(d, f))
    if RP > 0:
        base = synth.squiggle_batch(min(RP, 65536), M, 4243)
        big = np.lib.format.open_memmap(os.path.join(d, "reads.npy"), mode="w+", dtype=np.int16, shape=(RP, M))
        for lo in range(0, RP, base.shape[0]):
            big[lo:lo + base.shape[0]] = base[:min(base.shape[0], RP - lo)]
        big.flush()
        del big
        size = os.path.getsize(os.path.join(d, "reads.npy")) / 1e6
        run("segmenter.py --i16 (packed)", [py, seg, "--i16", os.path.join(d, "reads.npy")], RP, size, out)
        run("MotifSeq.py --i16 -m (packed)", [py, mot, "--i16", os.path.join(d, "reads.npy"), "-m", model], RP, size, out)
        arr = np.load(os.path.join(d, "reads.npy"), mmap_mode="r")
        fastio.write_blow5(os.path.join(d, "reads.blow5"), arr)
        del arr
        os.remove(os.path.join(d, "reads.npy"))
        size = os.path.getsize(os.path.join(d, "reads.blow5")) / 1e6
        run("segmenter.py --blow5 --raw_signal", [py, seg, "--blow5", os.path.join(d, "reads.blow5"), "--raw_signal"],
            RP, size, out)
        # without --raw_signal the reference converts every slow5 read to pA first (segmenter.py:366-370): since round 6 the
        # rows stay int16 on the device (raw-domain pA route)
        run("segmenter.py --blow5 (pA route)", [py, seg, "--blow5", os.path.join(d, "reads.blow5")], RP, size, out)
        run("MotifSeq.py --blow5 -m", [py, mot, "--blow5", os.path.join(d, "reads.blow5"), "-m", model], RP, size, out)
        os.remove(os.path.join(d, "reads.blow5"))
    os.rmdir(d)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
