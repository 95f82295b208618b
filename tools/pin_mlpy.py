#!/usr/bin/env python3
"""Pin the DTW core to mlpy -- the one-command job for whoever has mlpy 3.5.0 importable.

The reference's DTW is a third-party dependency that is NOT in /root/reference and cannot be installed in the build
container: `from mlpy import dtw_subsequence` (/root/reference/MotifSeq.py:12), called once, at
/root/reference/MotifSeq.py:437-439 (`dist, cost, path = dtw_subsequence(model[name], sig)`; the caller keeps dist,
path[1][0], path[1][-1]); README.md:78,85-96 names the version (mlpy 3.5.0) and where to get it.  oracle/sk_oracle.c restates
mlpy 3.5.0's cdtw.c (`subsequence`, `subsequence_path`) and every DTW parity test of this repository compares the HIP
kernels with THAT restatement -- so until this script has run somewhere, DTW parity is "unpinned" (DESIGN.md section 5,
tests/test_oracle_golden.py::test_dtw_pin_against_mlpy reports it by that name).

    pip3 install numpy cython && pip3 install mlpy-3.5.0.tar.gz      # README.md:85-96
    python3 tools/pin_mlpy.py                                        # writes tests/golden/dtw_mlpy.json, diffs vs oracle/
    python3 -m pytest tests/test_oracle_golden.py -k mlpy            # the oracle against the file, bit for bit

Inputs are committed or regenerated from fixed seeds (pin_cases): the normalised signals the reference itself handed to
dtw_subsequence on the example read and on synthetic reads (tests/golden/motifseq_norm.json.gz) against the example model,
the tie-heavy integer cases of tests/test_gpu_motifseq.py::test_dtw_raw_vs_oracle (back-trace tie order D3, first-minimum
argmin D2), random float cases over the kernel shapes, and signals holding inf / nan (what medmad makes of a MAD == 0 read,
tests/golden/motifseq_degenerate.json).  For each case mlpy's dist, path[1][0], path[1][-1] and the last cost row's
sha-256 are recorded -- dist as a hex float, so the comparison is bit for bit.

Exit status: 0 mlpy agreed with the oracle on every case; 1 a case differs (printed); 3 mlpy is not importable
(nothing written)."""
import gzip
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
OUT = os.path.join(GOLD, "dtw_mlpy.json")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pin_cases():
    """[(name, x, y)]: x = the motif (first argument of dtw_subsequence), y = the signal."""
    cases = []
    with gzip.open(os.path.join(GOLD, "motifseq_cli.json.gz"), "rt") as fh:
        model = np.array(json.load(fh)["model_expanded"]["values"], dtype=np.float64)
    with gzip.open(os.path.join(GOLD, "motifseq_norm.json.gz"), "rt") as fh:
        for k, v in enumerate(json.load(fh)["vectors"]):
            cases.append(("reference_normalised_%02d_%s_read%d" % (k, v["mode"], v["read"]), model,
                          np.array(v["y"], dtype=np.float64)))
    for nx in (1, 2, 5, 16, 17, 64, 163, 200, 257, 500):
        rng = np.random.default_rng(1000 + nx)                   # (the seeds of test_dtw_raw_vs_oracle)
        x = rng.normal(0, 1, nx)
        for ny in (1, 2, 3, 17, 64, 257, 1000):
            cases.append(("float_nx%d_ny%d" % (nx, ny), x, rng.normal(0, 1, ny)))
        xi = rng.integers(-2, 3, nx).astype(float)
        for ny in (5, 40, 333, 1200):
            cases.append(("ties_nx%d_ny%d" % (nx, ny), xi, rng.integers(-2, 3, ny).astype(float)))
    rng = np.random.default_rng(7)
    base = rng.normal(0, 1, 300)
    for name, poke in (("inf_inside", {50: np.inf}), ("minus_inf_inside", {120: -np.inf}), ("nan_inside", {7: np.nan}),
                       ("nan_first", {0: np.nan}), ("nan_last", {299: np.nan}), ("inf_and_nan", {3: np.inf, 200: np.nan}),
                       ("all_nan", {i: np.nan for i in range(300)}), ("all_inf", {i: np.inf for i in range(300)})):
        y = base.copy()
        for i, v in poke.items():
            y[i] = v
        cases.append(("degenerate_" + name, model[:40], y))
    return cases


def record(dist, path_y, last_row):
    return {"dist_hex": float(dist).hex(), "start": int(path_y[0]), "end": int(path_y[-1]), "path_len": int(len(path_y)),
            "last_row_sha256": hashlib.sha256(np.ascontiguousarray(last_row, dtype=np.float64).tobytes()).hexdigest()}


def oracle_record(ora, x, y):
    """the same record from oracle/ (finite signals: the restatement proper; inf / nan: its cdtw.c-literal variant)"""
    d, s, e, cost = ora.dtw_subsequence(x, y, want_cost=True)
    px, py = ora.dtw_subsequence_path(x, y)
    return record(d, py, cost[-1])


def main():
    try:
        import mlpy
    except Exception as e:                                       # noqa: BLE001
        sys.stderr.write("pin_mlpy: mlpy is not importable here (%r): DTW parity stays UNPINNED; nothing written.\n"
                         "          install mlpy 3.5.0 (/root/reference/README.md:78,85-96) and run this again.\n" % (e,))
        return 3
    from oracle import oracle as ora
    ora.build()
    out = {"generator": "tools/pin_mlpy.py", "mlpy_version": getattr(mlpy, "__version__", "unknown"),
           "numpy_version": np.__version__, "cases": {}}
    bad = 0
    for name, x, y in pin_cases():
        dist, cost, path = mlpy.dtw_subsequence(x, y)            # /root/reference/MotifSeq.py:437
        rec = record(dist, path[1], cost[-1])
        out["cases"][name] = rec
        mine = oracle_record(ora, x, y)
        same = all(rec[k] == mine[k] or (k == "dist_hex" and rec[k] == "nan" == mine[k]) for k in rec)
        if not same:
            bad += 1
            print("DIFFERS %-40s mlpy %s\n%48s oracle %s" % (name, rec, "", mine))
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("pin_mlpy: %d cases written to %s; %d differ from oracle/" % (len(out["cases"]), OUT, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
