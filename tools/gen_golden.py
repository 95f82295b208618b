#!/usr/bin/env python3
"""Mint golden vectors by RUNNING THE REFERENCE in this container.

    python tools/gen_golden.py            # writes tests/golden/*.json(.gz)

The reference (/root/reference, Python) is imported with empty stub modules for
the packages this image lacks (h5py, scrappy, mlpy); nothing of it is copied --
only inputs and the outputs it printed/returned are stored.  The GPU box never
sees /root/reference, so every test there runs from these committed fixtures.

What the fixtures pin
  * segmenter.py: scale_outliers + get_segs (KAT table, synthetic reads, the real
    example read) and main() stdout/stderr on TSVs built the way
    SquigglePull.print_data builds them           -> pins oracle S1-S4, harness S0
  * MotifSeq.py: everything AROUND the DTW -- filter, medmad/zscale (real numpy /
    sklearn), scoring, row formatting.  The `mlpy.dtw_subsequence` stub is bound
    to oracle/ (mlpy 3.5.0 is third party and absent): DTW digits in these rows
    are the restatement's, labelled "parity unpinned".
"""
import contextlib
import gzip
import hashlib
import io
import json
import os
import shutil
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import oracle as ora                      # noqa: E402
from squigglekit_amd import synth                      # noqa: E402
from squigglekit_amd.blow5 import read_blow5, to_pA    # noqa: E402


# --------------------------------------------------------------------------
# import the reference with stubs
# --------------------------------------------------------------------------
DTW_CALLS = []


def _dtw_stub(x, y):
    """Stands where mlpy.dtw_subsequence would be; records what it was given."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    dist, s, e = ora.dtw_subsequence(x, y)
    px, py = ora.dtw_subsequence_path(x, y)
    DTW_CALLS.append({"y": y.copy(), "x": x.copy()})
    return dist, None, (px, py)


class _FakeSquiggle:
    def __init__(self, rows):
        self._rows = rows

    def data(self, as_numpy=True, sloika=False):
        return self._rows


def _scrappy_stub_factory(model_path):
    """scrappy is absent; serve the pre-computed example .model (scrappie CLI
    output for the same sequence) in the (current, sd, -log dwell) layout that
    MotifSeq.convert_fasta consumes (MotifSeq.py:401-404)."""
    rows = []
    with open(model_path) as fh:
        for line in fh:
            if line[0] == "#" or line.startswith("pos"):
                continue
            f = line.split()
            rows.append((np.float32(f[2]), np.float32(f[3]), -np.log(float(f[4]))))

    def sequence_to_squiggle(seq, model=None):
        assert len(seq) == len(rows)
        return _FakeSquiggle(rows)
    return sequence_to_squiggle


def import_reference():
    sys.modules["h5py"] = types.ModuleType("h5py")
    mlpy = types.ModuleType("mlpy")
    mlpy.dtw_subsequence = _dtw_stub
    sys.modules["mlpy"] = mlpy
    scrappy = types.ModuleType("scrappy")
    scrappy.sequence_to_squiggle = _scrappy_stub_factory(
        os.path.join(REF, "example", "CATCTATCCAGGGTTAAATT.model"))
    sys.modules["scrappy"] = scrappy
    import matplotlib
    matplotlib.use("Agg")
    _use = matplotlib.use
    matplotlib.use = lambda *a, **k: None            # segmenter.py:3 asks for TkAgg
    sys.path.insert(0, REF)
    import segmenter
    import MotifSeq
    pys = types.ModuleType("pyslow5")
    pys.Open = lambda path, mode: types.SimpleNamespace(seq_reads=lambda: iter(SLOW5_READS))
    sys.modules["pyslow5"] = pys
    import dRNA_segmenter
    matplotlib.use = _use
    return segmenter, MotifSeq, dRNA_segmenter


SLOW5_READS = []          # what the pyslow5 stand-in serves: dicts {read_id, signal}


def run_main(mod, argv):
    """Run mod.main() with argv; capture (stdout, stderr, exit code)."""
    out, err = io.StringIO(), io.StringIO()
    old = sys.argv
    sys.argv = argv
    code = 0
    try:
        with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
            try:
                mod.main()
            except SystemExit as e:
                code = e.code if isinstance(e.code, int) else 1
    finally:
        sys.argv = old
    return out.getvalue(), err.getvalue(), code


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
def tsv_line(name, read_id, values, extra=None):
    """SquigglePull.print_data layout (SquigglePull.py:243-253)."""
    cols = [name, read_id]
    if extra is not None:
        cols += [str(v) for v in extra]
    return "\t".join(cols + [str(v) for v in values]) + "\n"


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def alt(n):
    return [300, 700] * (n // 2)


def ns(**kw):
    d = dict(error=5, corrector=50, window=150, seg_dist=50, std_scale=0.75,
             stall_len=0.25, lim_hi=900, lim_low=0)
    d.update(kw)
    return types.SimpleNamespace(**d)


def dump(name, obj):
    path = os.path.join(GOLD, name)
    data = json.dumps(obj, indent=None, separators=(",", ":")).encode()
    if name.endswith(".gz"):
        with gzip.GzipFile(path, "wb", mtime=0) as fh:
            fh.write(data)
    else:
        with open(path, "wb") as fh:
            fh.write(data)
    print("wrote", path, len(data), "bytes")


# --------------------------------------------------------------------------
def main():
    os.makedirs(GOLD, exist_ok=True)
    seg, mot, drna = import_reference()
    tmp = os.path.join(ROOT, ".golden_tmp")
    os.makedirs(tmp, exist_ok=True)

    # the only real read in the reference: example/slow5/0.blow5 (a data file)
    shutil.copyfile(os.path.join(REF, "example", "slow5", "0.blow5"),
                    os.path.join(GOLD, "example_0.blow5"))
    shutil.copyfile(os.path.join(REF, "example", "CATCTATCCAGGGTTAAATT.model"),
                    os.path.join(GOLD, "CATCTATCCAGGGTTAAATT.model"))
    shutil.copyfile(os.path.join(REF, "example", "CATCTATCCAGGGTTAAATT.fa"),
                    os.path.join(GOLD, "CATCTATCCAGGGTTAAATT.fa"))
    rec = next(read_blow5(os.path.join(GOLD, "example_0.blow5")))
    raw = rec["signal"]
    pa = to_pA(raw, rec["digitisation"], rec["offset"], rec["range"])
    extra = [rec["digitisation"], rec["offset"],
             float("{0:.2f}".format(rec["range"])), rec["sampling_rate"]]

    # ---------------- segmenter: KATs -------------------------
    kats = [
        ("const", [500] * 1000, {}),
        ("basic", [500] * 200 + alt(800), {}),
        ("middle", alt(400) + [500] * 200 + alt(400), {}),
        ("open_at_eof", alt(800) + [500] * 200, {}),
        ("first_stall_rule", [500] * 40 + alt(600) + [500] * 40 + alt(600), {}),
        ("cumulative_err", alt(300) + [500] * 100 + [700] * 3 + [500] * 100 + alt(300), {}),
        ("merge", alt(200) + [500] * 160 + alt(30) + [500] * 160 + alt(200), {}),
        ("no_merge", alt(200) + [500] * 160 + alt(60) + [500] * 160 + alt(200), {}),
        ("outlier_shift", [0, 1000] * 10 + alt(200) + [500] * 200 + alt(200), {}),
        ("corrector_live", alt(200) + [500] * 200 + [700] * 8 + [500] * 100 + alt(200),
         {"corrector": 0, "error": 10}),
        ("corrector_live2", alt(100) + ([500] * 60 + [700] * 4) * 12 + alt(100),
         {"corrector": 2, "error": 6, "window": 40}),
        ("small_window", alt(50) + ([500] * 30 + alt(20)) * 20, {"window": 20, "seg_dist": 5}),
        ("wide_band", alt(100) + [480, 520] * 300 + alt(100), {"std_scale": 2.0}),
    ]
    kat_out = []
    for name, sig, kw in kats:
        args = ns(**kw)
        a = np.array(sig, dtype=int)
        f = seg.scale_outliers(a, args)
        res = seg.get_segs(f, args)
        kat_out.append({"name": name, "sig": sig, "params": kw,
                        "segs": res if res else False})

    # ---------------- segmenter: synthetic reads through the reference -----
    syn = synth.squiggle_batch(256, 4000, synth.SEED_C2)
    variants = [{}, {"error": 10, "corrector": 0}, {"window": 80, "seg_dist": 20, "std_scale": 0.5},
                {"error": 8, "corrector": 3, "window": 60}, {"lim_hi": 700, "lim_low": 300}]
    syn_out = []
    for kw in variants:
        args = ns(**kw)
        rows = []
        for r in range(syn.shape[0]):
            s = syn[r].astype(int)[:-1]               # Num=-1 (segmenter.py:104,207)
            f = seg.scale_outliers(s, args)
            res = seg.get_segs(f, args)
            rows.append(res if res else [])
        syn_out.append({"params": kw, "segs": rows})
    # the real read, raw and pA, direct get_segs
    real_out = []
    for kind, arr in (("raw", raw.astype(int)), ("pA", pa)):
        args = ns()
        f = seg.scale_outliers(arr[:-1], args)
        res = seg.get_segs(f, args)
        real_out.append({"kind": kind, "n_after_filter": int(f.size),
                         "median": float(np.median(f)), "std": float(np.std(f)),
                         "segs": res})
    dump("segmenter_get_segs.json.gz", {
        "generator": "tools/gen_golden.py importing /root/reference/segmenter.py",
        "kats": kat_out,
        "synthetic": {"seed": synth.SEED_C2, "reads": 256, "samples": 4000,
                      "sha256": digest(syn), "drop_last": True, "runs": syn_out},
        "real_read": real_out})

    # ---------------- segmenter: main() on TSVs ------------------------------
    cli = []
    tsvs = {
        "pA_noinfo": tsv_line("test.fast5", rec["read_id"], pa),
        "raw_noinfo": tsv_line("test.fast5", rec["read_id"], raw),
        "pA_info": tsv_line("test.fast5", rec["read_id"], pa, extra),
        "raw_info": tsv_line("test.fast5", rec["read_id"], raw, extra),
    }
    # a multi-read synthetic TSV (8 reads, raw ints, 4 dummy leading columns so
    # that l[4:] is exactly the signal) incl. one all-zero read and one flat read
    lines = []
    for r in range(8):
        vals = syn[r]
        if r == 3:
            vals = np.zeros(50, dtype=np.int16)
        if r == 5:
            vals = np.full(800, 500, dtype=np.int16)
        lines.append("\t".join(["read%d.fast5" % r, "id%d" % r, "x", "y"] + [str(int(v)) for v in vals]) + "\n")
    tsvs["synthetic8"] = "".join(lines)
    for key, text in tsvs.items():
        path = os.path.join(tmp, key + ".tsv")
        with open(path, "w") as fh:
            fh.write(text)
        for flags in ([], ["-ku", "-j", "100"], ["-k", "-g", "-u", "-b", "100"],
                      ["-n", "20000"], ["-e", "10", "-c", "0", "-w", "100"]):
            so, se, code = run_main(seg, ["segmenter.py", "-s", path] + flags)
            cli.append({"tsv": key, "flags": flags, "stdout": so, "stderr": se, "exit": code})
    so, se, code = run_main(seg, ["segmenter.py"])
    cli.append({"tsv": None, "flags": [], "stdout": so, "stderr_head": se[:60], "exit": code})
    dump("segmenter_cli.json.gz", {
        "generator": "tools/gen_golden.py running /root/reference/segmenter.py main()",
        "synthetic8_sha256": digest(syn[:8]), "runs": cli})

    # ---------------- MotifSeq: main() with DTW stub -------------------------
    fa = os.path.join(REF, "example", "CATCTATCCAGGGTTAAATT.fa")
    model, m_order, L = mot.read_synth_model(os.path.join(REF, "example", "CATCTATCCAGGGTTAAATT.model"))
    mrows = []
    # MotifSeq wants data from column 8 (MotifSeq.py:270)
    def mline(name, rid, vals):
        return "\t".join([name, rid] + ["c%d" % i for i in range(6)] + [str(v) for v in vals]) + "\n"
    mot_syn = synth.squiggle_batch(6, 4000, synth.SEED_C3, motif=np.array(model[m_order[0]]))
    mt = {
        "real_raw": mline("test.fast5", rec["read_id"], raw),
        "real_pA": mline("test.fast5", rec["read_id"], pa),
        "synthetic6": "".join(mline("r%d.fast5" % r, "id%d" % r, [int(v) for v in mot_syn[r]]) for r in range(6)),
    }
    norm_vectors = []
    for key, text in mt.items():
        path = os.path.join(tmp, "m_" + key + ".tsv")
        with open(path, "w") as fh:
            fh.write(text)
        for flags in (["-l", "medmad"], ["-l", "zscale"], ["-x"],
                      ["--slope", "3.1", "--intercept", "-8", "--std_const", "0.1", "-scale_hi", "800", "-scale_low", "70"]):
            if "-x" in flags and key != "synthetic6":
                continue
            DTW_CALLS.clear()
            so, se, code = run_main(mot, ["MotifSeq.py", "-s", path, "-i", fa] + flags)
            calls = [{"n": int(c["y"].size), "sha256": digest(c["y"]),
                      "head": [float(v) for v in c["y"][:6]],
                      "tail": [float(v) for v in c["y"][-3:]]} for c in DTW_CALLS]
            mrows.append({"tsv": key, "flags": flags, "stdout": so, "stderr": se,
                          "exit": code, "dtw_inputs": calls})
            if key == "synthetic6" and flags[:2] in (["-l", "medmad"], ["-l", "zscale"]):
                for r, c in enumerate(DTW_CALLS):
                    norm_vectors.append({"mode": flags[1], "read": r,
                                         "y": [float(v) for v in c["y"]]})
    so, se, code = run_main(mot, ["MotifSeq.py", "-V"])
    mrows.append({"tsv": None, "flags": ["-V"], "stdout": so, "stderr": se, "exit": code})
    dump("motifseq_cli.json.gz", {
        "generator": "tools/gen_golden.py running /root/reference/MotifSeq.py main(); "
                     "mlpy.dtw_subsequence bound to oracle/ (DTW digits = restatement, parity unpinned)",
        "model_expanded": {"name": m_order[0], "L": L[0], "values": [float(v) for v in model[m_order[0]]]},
        "synthetic6_sha256": digest(mot_syn), "runs": mrows})
    dump("motifseq_norm.json.gz", {
        "generator": "normalised signals the reference handed to dtw_subsequence "
                     "(numpy medmad loop MotifSeq.py:192-200 / sklearn.scale :186-191)",
        "seed": synth.SEED_C3, "vectors": norm_vectors})

    # ---------------- dRNA_segmenter.py main() on its slow5 branch ----------------
    dreads = synth.drna_reads(24, 777)
    SLOW5_READS.clear()
    SLOW5_READS.append({"read_id": rec["read_id"], "signal": raw.copy()})
    for i, r in enumerate(dreads):
        SLOW5_READS.append({"read_id": "drna%02d" % i, "signal": r.copy()})
    so, se, code = run_main(drna, ["dRNA_segmenter.py", "-f", "served-by-stub.blow5"])
    dump("drna_cli.json", {
        "generator": "tools/gen_golden.py running /root/reference/dRNA_segmenter.py main() (-f branch; pyslow5 "
                     "stand-in serving the example read + synth.drna_reads(24, 777))",
        "seed": 777, "n": 24, "sha256": digest(np.concatenate(dreads)),
        "stdout": so, "stderr": se, "exit": code})

    # ---------------- dRNA_segmenter.py main() on its --signal branch (rolling mean) ----------------
    # The branch reads the local `w` before it is assigned (dRNA_segmenter.py:282, UnboundLocalError); the
    # script's own commented-out default is `# w = 2000` (:81).  To let the reference's code run (pandas
    # rolling / mean / std, the scan, the print), that one line is un-commented IN MEMORY: the module source
    # is read, patched and exec'd here; nothing of it is written anywhere.
    import tempfile
    import types
    import pandas as pd
    src = open(os.path.join(REF, "dRNA_segmenter.py")).read()
    assert src.count("    # w = 2000\n") == 1

    def patched_drna(w):
        mod = types.ModuleType("dRNA_segmenter_w%d" % w)
        exec(compile(src.replace("    # w = 2000\n", "    w = %d\n" % w), "dRNA_segmenter.py", "exec"), mod.__dict__)
        return mod
    rreads = synth.drna_reads(20, 4242, min_len=9000, max_len=30000)
    rreads.append(np.full(9000, 500, dtype=np.int16))                       # constant: std 0, nothing below
    rreads.append(np.concatenate([np.full(4000, 300), np.full(9000, 600)]).astype(np.int16))
    roll_runs = []
    for w in (2000, 1000):
        with tempfile.NamedTemporaryFile("w", suffix=".tsv", delete=False) as fh:
            for i, r in enumerate(rreads):
                fh.write(tsv_line("roll%02d.fast5" % i, "rid%02d" % i, r.tolist(), extra=[0, 0]))
            path = fh.name
        so, se, code = run_main(patched_drna(w), ["dRNA_segmenter.py", "-s", path])
        os.unlink(path)
        stats = []
        for r in rreads:                                                    # the dependency's own arithmetic
            f = r.astype(np.int64)
            f = f[(f > 0) & (f < 1200)]
            t = pd.Series(f).rolling(window=w).mean()
            stats.append({"n": int(f.size), "mn": float(t.mean()), "std": float(t.std()),
                          "t_sha256": digest(t.values)})
        roll_runs.append({"w": w, "stdout": so, "stderr": se, "exit": code, "stats": stats})
    dump("drna_roll.json", {
        "generator": "tools/gen_golden.py running /root/reference/dRNA_segmenter.py main() (-s branch) with its "
                     "commented-out `# w = 2000` (line 81) enabled in memory (w = 2000 / 1000); pandas %s"
                     % pd.__version__,
        "reads": "synth.drna_reads(20, 4242, min_len=9000, max_len=30000) + constant 500 x 9000 + step 300 x 4000 | 600 x 9000",
        "sha256": digest(np.concatenate(rreads)), "runs": roll_runs})

    # ---------------- numpy reductions the oracle must match bit-for-bit -----
    rng = np.random.default_rng(123)
    red = []
    for n in [1, 2, 7, 8, 9, 63, 127, 128, 129, 255, 1000, 3999, 4000, 8191, 8192, 8193, 20001, 36977]:
        xi = rng.integers(1, 900, size=n).astype(np.int64)
        xf = np.round(rng.normal(96.0, 15.0, size=n), 2)
        red.append({"n": n, "seed_note": "default_rng(123) sequential draws",
                    "int": {"sha256": digest(xi), "mean": float(np.mean(xi)), "std": float(np.std(xi)),
                            "median": float(np.median(xi))},
                    "flt": {"sha256": digest(xf), "mean": float(np.mean(xf)), "std": float(np.std(xf)),
                            "median": float(np.median(xf))}})
    dump("numpy_reductions.json", {"generator": "numpy %s" % np.__version__, "cases": red})

    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
