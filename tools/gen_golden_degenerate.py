#!/usr/bin/env python3
"""Mint the golden rows for MAD = 0 reads by RUNNING THE REFERENCE here (as tools/gen_golden.py does, same stubs).

    python tools/gen_golden_degenerate.py      # writes tests/golden/motifseq_degenerate.json

medmad divides by scaled_mad = 0 for such reads (MotifSeq.py:196-199): the reference hands inf / nan to
mlpy.dtw_subsequence and prints whatever comes back.  mlpy 3.5.0 is absent, so the stub is the oracle's literal
restatement of cdtw.c (`min3`: a; if b < a; if c < that -- and `fabs`, applied to non-finite values exactly as C does),
np.argmin's first-NaN rule and the back-trace: DTW digits in these rows are the restatement's ("parity unpinned" like
every D1-D3 golden), everything around them -- filter, the medmad loop, scoring with nan, the row's text -- is the
reference's own code.  `MotifSeq.py --strict-compat` must print these rows; the default reports the read on stderr."""
import json
import os
import sys
import tempfile
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg                                   # noqa: E402  (stubs, run_main, REF, GOLD)


def main():
    warnings.simplefilter("ignore")
    _seg, mot, _drna = gg.import_reference()
    fa = os.path.join(gg.REF, "example", "CATCTATCCAGGGTTAAATT.fa")
    rng = np.random.default_rng(20260928)
    normal = [int(v) for v in gg.synth.squiggle_batch(1, 600, 99)[0]]
    reads = {
        "constant": [500] * 300,                                            # every sample the median: 0 / 0 = nan
        "mostly_median": [500] * 200 + [int(v) for v in rng.integers(300, 700, 90)],   # MAD = 0, other samples -> +-inf
        "median_in_the_middle": [int(v) for v in rng.integers(300, 480, 40)] + [505] * 260 +
                                [int(v) for v in rng.integers(530, 700, 45)],
        "normal": normal,
    }
    order = list(reads)

    def mline(name, rid, vals):
        return "\t".join([name, rid] + ["c%d" % i for i in range(6)] + [str(v) for v in vals]) + "\n"
    runs = []
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "deg.tsv")
        with open(path, "w") as fh:
            for k in order:
                fh.write(mline(k + ".fast5", "id_" + k, reads[k]))
        for flags in ([], ["-x"]):
            gg.DTW_CALLS.clear()
            so, se, code = gg.run_main(mot, ["MotifSeq.py", "-s", path, "-i", fa] + flags)
            runs.append({"flags": flags, "stdout": so, "exit": code,
                         "dtw_inputs_nonfinite": [int(np.sum(~np.isfinite(c["y"]))) for c in gg.DTW_CALLS]})
    out = {"generator": "tools/gen_golden_degenerate.py running /root/reference/MotifSeq.py main(); mlpy.dtw_subsequence "
                        "bound to oracle/ (DTW digits = restatement of cdtw.c's arithmetic on inf / nan, parity unpinned)",
           "order": order, "reads": reads, "runs": runs}
    with open(os.path.join(gg.GOLD, "motifseq_degenerate.json"), "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    for r in runs:
        print(r["flags"], r["exit"], r["dtw_inputs_nonfinite"])
        for ln in r["stdout"].splitlines():
            print("   ", ln[:150])


if __name__ == "__main__":
    main()
