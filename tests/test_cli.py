"""The two command-line harness rows: stdout, stderr and exit code of
the drop-in tools vs what the reference printed (goldens from tools/gen_golden.py).

CPU variant: the GPU entry points of squigglekit_amd.api are replaced by oracle-backed
stand-ins so that the HARNESS (parsing, batching, ordering, formatting, messages) is pinned
without a GPU.  GPU variant (marked gpu): the same comparisons through the real HIP path."""
import contextlib
import hashlib
import io
import os
import sys
import types

import numpy as np
import pytest

from conftest import GOLD, load_golden


# ------------------------------------------------------------------ inputs, rebuilt like gen_golden
def tsv_line(name, read_id, values, extra=None):
    cols = [name, read_id] + ([str(v) for v in extra] if extra is not None else [])
    return "\t".join(cols + [str(v) for v in values]) + "\n"


@pytest.fixture(scope="module")
def tsv_files(tmp_path_factory, example_read):
    from squigglekit_amd import synth
    from squigglekit_amd.blow5 import to_pA
    d = tmp_path_factory.mktemp("tsv")
    rec = example_read
    raw = rec["signal"]
    pa = to_pA(raw, rec["digitisation"], rec["offset"], rec["range"])
    extra = [rec["digitisation"], rec["offset"], float("{0:.2f}".format(rec["range"])), rec["sampling_rate"]]
    texts = {"pA_noinfo": tsv_line("test.fast5", rec["read_id"], pa),
             "raw_noinfo": tsv_line("test.fast5", rec["read_id"], raw),
             "pA_info": tsv_line("test.fast5", rec["read_id"], pa, extra),
             "raw_info": tsv_line("test.fast5", rec["read_id"], raw, extra)}
    syn = synth.squiggle_batch(256, 4000, synth.SEED_C2)
    assert hashlib.sha256(syn[:8].tobytes()).hexdigest() == load_golden("segmenter_cli.json.gz")["synthetic8_sha256"]
    lines = []
    for r in range(8):
        vals = syn[r]
        if r == 3:
            vals = np.zeros(50, dtype=np.int16)
        if r == 5:
            vals = np.full(800, 500, dtype=np.int16)
        lines.append("\t".join(["read%d.fast5" % r, "id%d" % r, "x", "y"] + [str(int(v)) for v in vals]) + "\n")
    texts["synthetic8"] = "".join(lines)

    def mline(name, rid, vals):
        return "\t".join([name, rid] + ["c%d" % i for i in range(6)] + [str(v) for v in vals]) + "\n"
    model = np.array(load_golden("motifseq_cli.json.gz")["model_expanded"]["values"])
    msyn = synth.squiggle_batch(6, 4000, synth.SEED_C3, motif=model)
    texts["m_real_raw"] = mline("test.fast5", rec["read_id"], raw)
    texts["m_real_pA"] = mline("test.fast5", rec["read_id"], pa)
    texts["m_synthetic6"] = "".join(mline("r%d.fast5" % r, "id%d" % r, [int(v) for v in msyn[r]]) for r in range(6))
    paths = {}
    for k, t in texts.items():
        p = d / (k + ".tsv")
        p.write_text(t)
        paths[k] = str(p)
    return paths


@pytest.fixture
def scrappy_stub(monkeypatch):
    """Same stand-in tools/gen_golden.py gave the reference: float32 currents from the example model."""
    rows = []
    with open(os.path.join(GOLD, "CATCTATCCAGGGTTAAATT.model")) as fh:
        for line in fh:
            if line[0] == "#" or line.startswith("pos"):
                continue
            f = line.split()
            rows.append((np.float32(f[2]), np.float32(f[3]), -np.log(float(f[4]))))
    mod = types.ModuleType("scrappy")
    mod.sequence_to_squiggle = lambda seq, model=None: types.SimpleNamespace(
        data=lambda as_numpy=True, sloika=False: rows)
    monkeypatch.setitem(sys.modules, "scrappy", mod)


def run_cli(main, argv):
    out, err = io.StringIO(), io.StringIO()
    code = 0
    with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
        try:
            main(argv)
        except SystemExit as e:
            code = e.code if isinstance(e.code, int) else 1
    return out.getvalue(), err.getvalue(), code


@pytest.fixture
def oracle_backend(monkeypatch, ora):
    """CPU only: api's GPU calls answered by the oracle (tests may use the oracle as a checker/fake)."""
    from squigglekit_amd import _lib, api

    def seg_any(reads, params=None):
        params = params or _lib.SegParams()
        op = ora.SegParams(params.error, params.corrector, params.window, params.seg_dist,
                           params.std_scale, params.stall_len)
        return [ora.get_segs(ora.scale_outliers(np.asarray(r, float), params.lim_low, params.lim_hi), op)
                for r in reads]

    def norm(sig, scale="medmad", lo=0, hi=1200):
        f = ora.scale_outliers(np.asarray(sig, float), lo, hi)
        return ora.medmad(f)[0] if scale == "medmad" else ora.zscale(f)[0]

    def mot_any(reads, motif, scale="medmad", lo=0, hi=1200):
        out = np.zeros(len(reads), dtype=_lib.HIT_DTYPE)
        for i, r in enumerate(reads):
            y = norm(r, scale, lo, hi)
            out[i]["n"] = y.size
            if y.size and not np.all(np.isfinite(y)):
                out[i]["flags"] = 2                                     # MAD = 0: flagged, as the library does
                out[i]["dist"], out[i]["start"], out[i]["end"] = np.nan, -1, -1
            elif y.size:
                out[i]["dist"], out[i]["start"], out[i]["end"] = ora.dtw_subsequence(motif, y)
            else:
                out[i]["flags"] = 1
        return out

    def seg_batch(sig, lens=None, params=None, max_segs=64, devices=None):
        params = params or _lib.SegParams()
        op = ora.SegParams(params.error, params.corrector, params.window, params.seg_dist,
                           params.std_scale, params.stall_len)
        sig = np.ascontiguousarray(sig, dtype=np.int16)
        lens = np.full(sig.shape[0], sig.shape[1], dtype=np.int32) if lens is None else np.asarray(lens, dtype=np.int32)
        while True:
            segs, nsegs = ora.segment_batch_i16(sig, lens, op, lo=params.lim_low, hi=params.lim_hi, max_segs=max_segs)
            if nsegs.size == 0 or nsegs.max() <= max_segs:
                return segs, nsegs
            max_segs = int(nsegs.max()) + 8

    def mot_batch(sig, lens, motifs, scale="medmad", lo=0, hi=1200):
        reads = [np.asarray(sig[r, :lens[r]]) for r in range(sig.shape[0])]
        return [mot_any(reads, m, scale, lo, hi) for m in motifs]

    def as_f64(values):
        """int32 = centi-units (tsvio.FloatBlock.centi, round 6): the device makes c / 100.0 = float("ddd.dd")"""
        values = np.asarray(values)
        return values / 100.0 if values.dtype == np.int32 else values

    def seg_ragged(values, off, lens=None, params=None, max_segs=64):
        values = as_f64(values)
        R = len(off) - 1
        reads = [np.asarray(values[off[r]:off[r] + (int(lens[r]) if lens is not None else off[r + 1] - off[r])]) for r in range(R)]
        res = seg_any(reads, params)
        k = max([len(x) for x in res if x] + [1])
        segs = np.zeros((R, k, 2), dtype=np.int32)
        nsegs = np.zeros(R, dtype=np.int32)
        for r, x in enumerate(res):
            if x:
                nsegs[r] = len(x)
                segs[r, :len(x)] = x
        return segs, nsegs

    def mot_ragged(values, off, motifs, scale="medmad", lo=0, hi=1200):
        values = as_f64(values)
        reads = [np.asarray(values[off[r]:off[r + 1]]) for r in range(len(off) - 1)]
        return [mot_any(reads, m, scale, lo, hi) for m in motifs]

    def seg_pa(sig, lens, calib, params=None, max_segs=64):
        from squigglekit_amd.blow5 import to_pA
        reads = [to_pA(np.asarray(sig[r, :lens[r]]), calib[r][0], calib[r][1], calib[r][2]) for r in range(len(lens))]
        off = np.concatenate([[0], np.cumsum([x.size for x in reads])]).astype(np.int64)
        return seg_ragged(np.concatenate(reads) if reads else np.zeros(0), off, None, params, max_segs)

    monkeypatch.setattr(api, "segment_batch_pa", seg_pa)
    monkeypatch.setattr(api, "segment_ragged_f64", seg_ragged)
    monkeypatch.setattr(api, "motifseq_multi_ragged_f64", mot_ragged)
    monkeypatch.setattr(api, "segment_batch", seg_batch)
    monkeypatch.setattr(api, "motifseq_multi_batch", mot_batch, raising=False)
    monkeypatch.setattr(api, "segment_any", seg_any)
    monkeypatch.setattr(api, "motifseq_any", mot_any)
    monkeypatch.setattr(api, "motifseq_multi",
                        lambda reads, motifs, scale="medmad", lo=0, hi=1200: [mot_any(reads, m, scale, lo, hi)
                                                                               for m in motifs])
    monkeypatch.setattr(api, "normalise", norm)
    monkeypatch.setattr(api, "dtw_subsequence_cref", lambda x, y: ora.dtw_subsequence(x, y))
    monkeypatch.setattr(_lib, "init", lambda device=None: 0)


# ------------------------------------------------------------------ the comparisons
def check_segmenter(tsv_files):
    from squigglekit_amd.segmenter_cli import main
    gold = load_golden("segmenter_cli.json.gz")
    n = 0
    for run in gold["runs"]:
        if run["tsv"] is None:
            so, se, code = run_cli(main, [])
            assert (so, code) == (run["stdout"], run["exit"]) and se.startswith("usage:")
            continue
        so, se, code = run_cli(main, ["-s", tsv_files[run["tsv"]]] + run["flags"])
        assert so == run["stdout"], (run["tsv"], run["flags"])
        assert code == run["exit"]
        assert _strip_path(se) == _strip_path(run["stderr"]), (run["tsv"], run["flags"], se, run["stderr"])
        n += 1
    assert n >= 20


def _strip_path(s):
    """The reference's 'No signal found in file: <path> <name>' embeds the temp path it was given,
    and tracebacks name the file/line of whoever raised: both are normalised before comparing."""
    import re
    s = re.sub(r"No signal found in file: \S+ ", "No signal found in file: <tsv> ", s)
    return re.sub(r'  File "[^"]+", line \d+, in ', "  File <f>, in ", s)


def check_motifseq(tsv_files):
    from squigglekit_amd.motifseq_cli import main
    gold = load_golden("motifseq_cli.json.gz")
    fa = os.path.join(GOLD, "CATCTATCCAGGGTTAAATT.fa")
    n = 0
    for run in gold["runs"]:
        if run["tsv"] is None:
            so, se, code = run_cli(main, run["flags"])
            assert (so, se, code) == (run["stdout"], run["stderr"], run["exit"])
            continue
        so, se, code = run_cli(main, ["-s", tsv_files["m_" + run["tsv"]], "-i", fa] + run["flags"])
        assert so == run["stdout"], (run["tsv"], run["flags"], so[-300:], run["stdout"][-300:])
        assert (se, code) == (run["stderr"], run["exit"])
        n += 1
    assert n >= 9


def test_segmenter_cli_harness_cpu(oracle_backend, tsv_files):
    check_segmenter(tsv_files)


def test_motifseq_cli_harness_cpu(oracle_backend, scrappy_stub, tsv_files):
    check_motifseq(tsv_files)


def test_stats_json_side_file_leaves_the_output_alone(oracle_backend, scrappy_stub, tsv_files, tmp_path):
    """[extension] --stats-json PATH / --stats (SURVEY section 5 "metrics"): reads, reads per second, input GB per second
    and GPU calls of the run go to a side file / one stderr line; stdout is the reference's, byte for byte."""
    import json as _json
    from squigglekit_amd.segmenter_cli import main as seg_main
    from squigglekit_amd.motifseq_cli import main as mot_main
    gold = load_golden("segmenter_cli.json.gz")
    run = [r for r in gold["runs"] if r["tsv"] == "pA_noinfo" and r["flags"] == []][0]
    js = str(tmp_path / "seg.json")
    so, se, code = run_cli(seg_main, ["-s", tsv_files["pA_noinfo"], "--stats-json", js, "--stats"])
    assert so == run["stdout"] and code == run["exit"]
    assert se.startswith(run["stderr"][:-4] if run["stderr"].endswith("Done") else run["stderr"][:10]) and "[stats] segmenter:" in se
    rec = _json.load(open(js))
    assert rec["tool"] == "segmenter" and rec["reads"] >= 1 and rec["gpu_calls"] >= 1 and rec["reads_per_s"] > 0
    assert rec["input_bytes"] == os.path.getsize(tsv_files["pA_noinfo"])
    gm = load_golden("motifseq_cli.json.gz")
    mrun = [r for r in gm["runs"] if r["tsv"] == "real_raw" and r["flags"] == ["-l", "medmad"]][0]
    jm = str(tmp_path / "mot.json")
    model = os.path.join(GOLD, "CATCTATCCAGGGTTAAATT.model")
    so, se, code = run_cli(mot_main, ["-s", tsv_files["m_real_raw"], "-m", model, "-l", "medmad", "--stats-json", jm])
    rec = _json.load(open(jm))
    assert rec["tool"] == "MotifSeq" and rec["reads"] == 1 and rec["gpu_calls"] == 1 and "[stats]" not in se
    assert so.strip().split("\n")[1].split("\t")[3:5] == mrun["stdout"].strip().split("\n")[1].split("\t")[3:5]


@pytest.mark.gpu
def test_segmenter_cli_gpu(gpu, tsv_files):
    check_segmenter(tsv_files)


@pytest.mark.gpu
def test_motifseq_cli_gpu(gpu, scrappy_stub, tsv_files):
    check_motifseq(tsv_files)


def test_motifseq_model_file_variants(oracle_backend, tsv_files, tmp_path):
    """-m with the shipped scrappie text works (the reference crashes there); a bait TSV works;
    --strict-compat reproduces the reference's header-only behaviour."""
    from squigglekit_amd.motifseq_cli import main, HEADER
    model = os.path.join(GOLD, "CATCTATCCAGGGTTAAATT.model")
    so, se, code = run_cli(main, ["-s", tsv_files["m_synthetic6"], "-m", model])
    rows = so.strip().split("\n")
    assert rows[0].split("\t") == HEADER and len(rows) == 7 and code == 0
    assert rows[1].split("\t")[2] == "3_prime_end" and rows[1].split("\t")[7] == "48.4"
    vals = load_golden("motifseq_cli.json.gz")["model_expanded"]["values"]
    bait = tmp_path / "bait.tsv"
    bait.write_text("3_prime_end\t20\t.\t" + "\t".join(repr(v) for v in vals) + "\n")
    so2, _, _ = run_cli(main, ["-s", tsv_files["m_synthetic6"], "-m", str(bait)])
    assert so2 == so
    so3, _, _ = run_cli(main, ["-s", tsv_files["m_synthetic6"], "-m", model, "--strict-compat"])
    assert so3.strip().split("\n") == ["\t".join(HEADER)]


def test_gz_and_error_paths(oracle_backend, tsv_files, tmp_path):
    import gzip
    from squigglekit_amd.segmenter_cli import main
    gz = tmp_path / "x.tsv.gz"
    with gzip.open(gz, "wt") as fh:
        fh.write(open(tsv_files["synthetic8"]).read())
    so, se, code = run_cli(main, ["-s", str(gz)])
    so0, _, _ = run_cli(main, ["-s", tsv_files["synthetic8"]])
    assert so == so0 and so.count("\n") >= 5
    so, se, code = run_cli(main, ["--bogus"])
    assert code == 2 and se.startswith("error: ")


@pytest.mark.gpu
def test_motifseq_cli_constant_read_is_reported_not_printed(gpu, tmp_path):
    """A read whose MAD is 0 (medmad divides by it, MotifSeq.py:196-199): the library flags it
    (SK_FLAG_DEGENERATE) and the CLI says so on stderr instead of printing a row of inf / nan."""
    from squigglekit_amd.motifseq_cli import main
    p = tmp_path / "const.tsv"
    good = (np.arange(3000) % 97 + 400).tolist()
    p.write_text("\t".join(["a.fast5", "flat"] + ["c%d" % i for i in range(6)] + ["500"] * 2000) + "\n"
                 + "\t".join(["b.fast5", "fine"] + ["c%d" % i for i in range(6)] + [str(v) for v in good]) + "\n")
    so, se, code = run_cli(main, ["-s", str(p), "-m", os.path.join(GOLD, "CATCTATCCAGGGTTAAATT.model")])
    rows = so.strip().split("\n")
    assert code == 0 and len(rows) == 2 and rows[1].startswith("b.fast5\tfine\t")
    assert "the MAD of flat is 0" in se and "flat" not in so
    assert "note: -m searches for the model's motif(s)" in se


@pytest.mark.gpu
def test_packed_and_blow5_inputs_equal_the_tsv_route(gpu, tmp_path):
    """[extensions] --i16 / --blow5: the same reads through the packed-npy, the BLOW5 (stored and zlib) and the TSV
    inputs give the same numbers -- every column but the name columns -- for both tools; a second model file with two
    motifs checks the read-major row order of the native table."""
    from squigglekit_amd import fastio, synth
    from squigglekit_amd.motifseq_cli import main as mmain
    from squigglekit_amd.segmenter_cli import main as smain
    R, M = 300, 3000
    motif = synth.synthetic_motif(163, seed=11)
    sig = synth.squiggle_batch(R, M, 777, motif=motif)
    sig[7, :] = 500                                           # MAD = 0: flagged -> the per-read route inside a block
    np.save(tmp_path / "r.npy", sig)
    ids = ["read-%04d" % i for i in range(R)]
    fastio.write_blow5(str(tmp_path / "r.blow5"), sig, ids)
    fastio.write_blow5(str(tmp_path / "rz.blow5"), sig, ids, compress=True)
    with open(tmp_path / "m.tsv", "w") as fm, open(tmp_path / "s.tsv", "w") as fs:
        for i in range(R):
            vals = "\t".join(str(int(v)) for v in sig[i])
            fm.write("\t".join(["f.fast5", ids[i]] + ["x"] * 6) + "\t" + vals + "\n")
            fs.write("\t".join([ids[i], "a", "b", "c"]) + "\t" + vals + "\n")
    model = os.path.join(GOLD, "CATCTATCCAGGGTTAAATT.model")
    two = tmp_path / "two.model"
    vals = load_golden("motifseq_cli.json.gz")["model_expanded"]["values"]
    two.write_text("mA\t20\t.\t" + "\t".join(repr(v) for v in vals) + "\n" +
                   "mB\t12\t.\t" + "\t".join(repr(v) for v in vals[20:120]) + "\n")

    def cols(text, first):
        return [ln.split("\t")[first:] for ln in text.strip().split("\n")]

    for mfile, K in ((model, 1), (str(two), 2)):
        ref, err, code = run_cli(mmain, ["-s", str(tmp_path / "m.tsv"), "-m", mfile])
        assert code == 0 and len(ref.strip().split("\n")) == 1 + K * (R - 1), err[-500:]
        for argv in (["--i16", str(tmp_path / "r.npy")], ["--blow5", str(tmp_path / "r.blow5")],
                     ["--blow5", str(tmp_path / "rz.blow5")]):
            got, err2, code = run_cli(mmain, argv + ["-m", mfile])
            assert code == 0 and "MAD" in err2
            assert cols(got, 2) == cols(ref, 2), argv                # model, start .. hit_Probability
            if argv[0] == "--blow5":
                assert [r[1] for r in cols(got, 0)[1:]] == [r[1] for r in cols(ref, 0)[1:]]     # readID column
    ref, _, code = run_cli(smain, ["-s", str(tmp_path / "s.tsv")])
    assert code == 0 and ref.count("\n") > R // 2
    for argv in (["--i16", str(tmp_path / "r.npy")], ["--blow5", str(tmp_path / "r.blow5"), "--raw_signal"],
                 ["--blow5", str(tmp_path / "rz.blow5"), "--raw_signal"]):
        got, _, code = run_cli(smain, argv)
        assert code == 0 and cols(got, 1) == cols(ref, 1), argv
        if argv[0] == "--blow5":
            assert got == ref
    got, _, _ = run_cli(smain, ["--i16", str(tmp_path / "r.npy"), "-u", "-k"])          # -u: the per-read checks
    want, _, _ = run_cli(smain, ["-s", str(tmp_path / "s.tsv"), "-u", "-k"])
    assert cols(got, 1) == cols(want, 1)


@pytest.mark.gpu
def test_strict_compat_rows_while_the_next_block_is_on_the_gpu(gpu, scrappy_stub, tmp_path, monkeypatch):
    """--strict-compat prints a MAD = 0 read's nan row by calling the GPU from the MAIN thread (normalise + the literal
    mlpy kernel) -- while the worker thread may already run the next block on the same context (one stream, shared
    scratch).  Many small blocks, a constant read in every one: the table must equal the one-block run, where nothing
    overlaps (round-4 advisor finding: the worker has to be finished first)."""
    from squigglekit_amd import synth
    from squigglekit_amd.motifseq_cli import main as mmain
    R, M = 3000, 2000
    motif = synth.synthetic_motif(163, seed=11)
    sig = synth.squiggle_batch(R, M, 4242, motif=motif)
    sig[5::37, :] = 500                                       # MAD = 0 reads in every block
    np.save(tmp_path / "r.npy", sig)
    fa = os.path.join(GOLD, "CATCTATCCAGGGTTAAATT.fa")        # (-m under --strict-compat is the reference's header-only defect)
    monkeypatch.setenv("SK_I16_BLOCK_MB", "1024")
    one, err1, code = run_cli(mmain, ["--i16", str(tmp_path / "r.npy"), "-i", fa, "--strict-compat"])
    nan_rows = one.count("\tnan\t")                            # (one per constant read and motif of the fasta)
    assert code == 0 and nan_rows and nan_rows % len(range(5, R, 37)) == 0, err1[-300:]
    monkeypatch.setenv("SK_I16_BLOCK_MB", "1")                # 262 reads per block: a dozen blocks in flight one after the other
    for _ in range(3):
        many, err2, code = run_cli(mmain, ["--i16", str(tmp_path / "r.npy"), "-i", fa, "--strict-compat"])
        assert code == 0 and many == one


@pytest.mark.gpu
def test_launchers_as_processes_flush_everything_before_the_fast_exit(gpu, tmp_path):
    """The root launchers leave through os._exit once main() has returned (no interpreter / HIP teardown): everything
    printed must still arrive -- through a pipe, the case in which Python block-buffers stdout -- with exit code 0,
    and equal what main() prints in-process; a usage error keeps its ordinary exit code."""
    import subprocess
    from squigglekit_amd import synth
    from squigglekit_amd.motifseq_cli import main as mmain
    from squigglekit_amd.segmenter_cli import main as smain
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    R, M = 500, 2000
    sig = synth.squiggle_batch(R, M, 4711, motif=synth.synthetic_motif(60, seed=3))
    np.save(tmp_path / "r.npy", sig)
    with open(tmp_path / "s.tsv", "w") as fs:
        for i in range(40):
            fs.write("\t".join(["id%d" % i, "a", "b", "c"]) + "\t" + "\t".join(str(int(v)) for v in sig[i]) + "\n")
    model = os.path.join(GOLD, "CATCTATCCAGGGTTAAATT.model")
    for tool, main, argv in (("MotifSeq.py", mmain, ["--i16", str(tmp_path / "r.npy"), "-m", model]),
                             ("segmenter.py", smain, ["--i16", str(tmp_path / "r.npy")]),
                             ("segmenter.py", smain, ["-s", str(tmp_path / "s.tsv")])):
        want, _, code = run_cli(main, argv)
        assert code == 0
        p = subprocess.run([sys.executable, os.path.join(root, tool)] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=300)
        assert p.returncode == 0, p.stderr.decode()[-500:]
        assert p.stdout.decode() == want, tool
    p = subprocess.run([sys.executable, os.path.join(root, "MotifSeq.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=300)
    assert p.returncode == 1 and b"usage" in p.stderr.lower()


@pytest.mark.gpu
def test_after_stall_with_packed_and_blow5_inputs(gpu, tmp_path):
    """[extensions] --after_stall must mean the same thing whatever the input route: get_segs, then the search behind
    the stall, a search_from column in every row (the packed routes used to search the whole read and print 12 columns
    under a 13-column header)."""
    from squigglekit_amd import fastio, synth
    from squigglekit_amd.motifseq_cli import main as mmain
    R, M = 120, 3000
    motif = synth.synthetic_motif(163, seed=11)
    sig = synth.squiggle_batch(R, M, 4242, motif=motif)
    np.save(tmp_path / "r.npy", sig)
    ids = ["read-%04d" % i for i in range(R)]
    fastio.write_blow5(str(tmp_path / "r.blow5"), sig, ids)
    with open(tmp_path / "m.tsv", "w") as fm:
        for i in range(R):
            fm.write("\t".join(["f.fast5", ids[i]] + ["x"] * 6) + "\t" + "\t".join(str(int(v)) for v in sig[i]) + "\n")
    model = os.path.join(GOLD, "CATCTATCCAGGGTTAAATT.model")
    ref, err, code = run_cli(mmain, ["-s", str(tmp_path / "m.tsv"), "-m", model, "--after_stall"])
    assert code == 0, err[-400:]
    rows = [ln.split("\t") for ln in ref.strip().split("\n")]
    assert rows[0][-1] == "search_from" and all(len(r) == len(rows[0]) for r in rows)
    assert any(int(r[-1]) > 0 for r in rows[1:]), "no read had a stall to search behind"
    plain, _, _ = run_cli(mmain, ["-s", str(tmp_path / "m.tsv"), "-m", model])
    assert [r[3:12] for r in rows[1:]] != [ln.split("\t")[3:12] for ln in plain.strip().split("\n")[1:]]
    for argv in (["--i16", str(tmp_path / "r.npy")], ["--blow5", str(tmp_path / "r.blow5")]):
        got, err2, code = run_cli(mmain, argv + ["-m", model, "--after_stall"])
        assert code == 0, err2[-400:]
        grows = [ln.split("\t") for ln in got.strip().split("\n")]
        assert [r[2:] for r in grows] == [r[2:] for r in rows], argv          # every column but the two name columns


def test_blow5_errors_are_messages_not_tracebacks(oracle_backend, tmp_path):
    """A truncated BLOW5 file, one cut inside a size field, one with zstd records: one line on stderr, exit 1, the rows
    decoded before the damage still printed -- for every --blow5 branch of both tools."""
    from squigglekit_amd import fastio, synth
    from squigglekit_amd.motifseq_cli import main as mmain
    from squigglekit_amd.segmenter_cli import main as smain
    sig = synth.squiggle_batch(40, 1500, 3)
    good = tmp_path / "g.blow5"
    fastio.write_blow5(str(good), sig, ["r%d" % i for i in range(40)])
    data = good.read_bytes()
    cut = tmp_path / "cut.blow5"
    cut.write_bytes(data[:len(data) - 1000])
    cut2 = tmp_path / "cut2.blow5"
    cut2.write_bytes(data[:len(data) - 5 - 3000 - 3])             # inside a record
    zstd = tmp_path / "zstd.blow5"
    zstd.write_bytes(data[:9] + b"\x02" + data[10:])
    model = os.path.join(GOLD, "CATCTATCCAGGGTTAAATT.model")
    for path in (cut, cut2, zstd):
        for main, argv, tool in ((smain, ["--blow5", str(path), "--raw_signal"], "segmenter"),
                                 (smain, ["--blow5", str(path)], "segmenter"),
                                 (mmain, ["--blow5", str(path), "-m", model], "MotifSeq")):
            out, err, code = run_cli(main, argv)
            assert code == 1, (path.name, argv, err[-300:])
            assert "Traceback" not in err and "%s: --blow5:" % tool in err, (path.name, argv, err[-300:])
    out, err, code = run_cli(smain, ["--blow5", str(good), "--raw_signal"])
    assert code == 0 and "--blow5:" not in err


def _degenerate_case(tmp_path):
    gold = load_golden("motifseq_degenerate.json")
    path = tmp_path / "deg.tsv"
    with open(path, "w") as fh:
        for k in gold["order"]:
            fh.write("\t".join([k + ".fast5", "id_" + k] + ["c%d" % i for i in range(6)] +
                               [str(v) for v in gold["reads"][k]]) + "\n")
    return gold, str(path), os.path.join(GOLD, "CATCTATCCAGGGTTAAATT.fa")


def _check_degenerate(tmp_path):
    """MAD = 0 reads (MotifSeq.py:196-199 divides by zero): --strict-compat prints the rows the reference prints (minted
    by tools/gen_golden_degenerate.py running the reference: nan distance at the first sample that equals the median);
    the default reports those reads on stderr and prints the others unchanged."""
    from squigglekit_amd.motifseq_cli import main as mmain
    gold, tsv, fa = _degenerate_case(tmp_path)
    for run in gold["runs"]:
        out, err, code = run_cli(mmain, ["-s", tsv, "-i", fa, "--strict-compat"] + run["flags"])
        assert code == 0 and out == run["stdout"], (run["flags"], out[-400:], err[-300:])
        assert "nan" in out
    out, err, code = run_cli(mmain, ["-s", tsv, "-i", fa])
    want = [ln for ln in gold["runs"][0]["stdout"].splitlines() if "\tnan\t" not in ln]
    assert code == 0 and out.splitlines() == want and err.count("MAD of") == 3


def test_motifseq_strict_compat_degenerate_rows_cpu(oracle_backend, scrappy_stub, tmp_path):
    _check_degenerate(tmp_path)


@pytest.mark.gpu
def test_motifseq_strict_compat_degenerate_rows_gpu(gpu, scrappy_stub, tmp_path):
    """... and through the real backend: the division on the GPU (sk_normalise_*), mlpy's C arithmetic evaluated
    literally by k_dtw_cref (sk_dtw_subsequence_cref)."""
    _check_degenerate(tmp_path)
    # the literal kernel agrees with the systolic ones wherever both apply (finite input)
    from squigglekit_amd import api, synth
    x = synth.synthetic_motif(37, seed=3)
    y = api.normalise(synth.squiggle_batch(1, 700, 5)[0])
    d, cost, path = api.dtw_subsequence(x, y)
    assert api.dtw_subsequence_cref(x, y) == (d, int(path[1][0]), int(path[1][-1]))


@pytest.mark.gpu
def test_pa_tsv_block_route(gpu, ora, tmp_path):
    """pA TSVs (SquigglePull's default output: decimals) go to the GPU a whole chunk at a time, straight from the float64
    tokenizer (tsvio.FloatBlock, sk_segment_batch_f64_len / sk_motifseq_batch_f64): the table must be what the per-line
    route prints -- checked against the oracle read by read, with and without the -n cut, and with a chunk that holds an
    odd token (falls back to the per-line route) or an integer line among the decimal ones."""
    from squigglekit_amd import synth
    from squigglekit_amd.motifseq_cli import main as mmain
    from squigglekit_amd.segmenter_cli import main as smain
    R, M = 200, 3000
    sig = synth.squiggle_batch(R, M, 606)
    pa = np.round((sig.astype(np.int64) + 16.0) * (1493.94 / 8192.0), 2)
    pa[5] = np.rint(pa[5])                                       # a line of integer-valued tokens ("95.0" -> still decimal)
    lines_s = ["\t".join(["r%d.fast5" % r, "a", "b", "c"] + [repr(float(v)) for v in pa[r]]) for r in range(R)]
    lines_m = ["\t".join(["f.fast5", "id%d" % r] + ["x"] * 6 + [repr(float(v)) for v in pa[r]]) for r in range(R)]
    (tmp_path / "s.tsv").write_text("\n".join(lines_s) + "\n")
    (tmp_path / "m.tsv").write_text("\n".join(lines_m) + "\n")
    odd = list(lines_s)
    odd[17] = odd[17].replace("\t", "\t ", 5)                    # tokens with a leading blank: SK_TSV_SLOW -> per-line route
    (tmp_path / "odd.tsv").write_text("\n".join(odd) + "\n")

    def want_seg(num):
        out = []
        for r in range(R):
            x = pa[r][:num] if num else pa[r][:-1]               # segmenter.py:104-105,207
            segs = ora.get_segs(ora.scale_outliers(x, 0, 900))
            if segs:
                out.append("r%d.fast5\t%s" % (r, ",".join(str(v) for p in segs for v in p)))
        return out
    for argv, num in ((["-s", str(tmp_path / "s.tsv")], 0), (["-s", str(tmp_path / "s.tsv"), "-n", "2500"], 2500),
                      (["-s", str(tmp_path / "odd.tsv")], 0)):
        got, err, code = run_cli(smain, argv)
        assert code == 0 and got.strip().split("\n") == want_seg(num), (argv, err[-300:])
    model = os.path.join(GOLD, "CATCTATCCAGGGTTAAATT.model")
    motif = np.array(load_golden("motifseq_cli.json.gz")["model_expanded"]["values"])
    got, err, code = run_cli(mmain, ["-s", str(tmp_path / "m.tsv"), "-m", model])
    rows = [ln.split("\t") for ln in got.strip().split("\n")[1:]]
    assert code == 0 and len(rows) == R, err[-300:]
    for r in (0, 5, 17, 99, R - 1):
        d, s0, e0 = ora.dtw_subsequence(motif, ora.medmad(ora.scale_outliers(pa[r], 0, 1200))[0])
        assert rows[r][1] == "id%d" % r and (int(rows[r][3]), int(rows[r][4]), float(rows[r][6])) == (s0, e0, d), r


@pytest.mark.gpu
def test_blow5_default_pa_route_on_gpu(gpu, ora, tmp_path):
    """`segmenter.py --blow5 x` without --raw_signal works in pA like the reference does for fast5 / slow5 input
    (segmenter.py:345-349): records decoded natively, np.round((raw + offset) * (float("%.2f" % range) / digitisation), 2)
    made on the GPU (sk_segment_batch_i16_pa), float64 segmenter -- the table must be what the record-by-record Python
    route (kept for -u) prints, and the reference arithmetic restated by blow5.to_pA + the oracle."""
    from squigglekit_amd import blow5, fastio, synth
    from squigglekit_amd.segmenter_cli import main as smain
    R, M = 300, 3000
    sig = synth.squiggle_batch(R, M, 4711)
    ids = ["read-%04d" % i for i in range(R)]
    path = str(tmp_path / "r.blow5")
    fastio.write_blow5(path, sig, ids)
    got, err, code = run_cli(smain, ["--blow5", path])
    assert code == 0, err[-300:]
    want = []
    for rec in blow5.read_blow5(path):
        pa = blow5.to_pA(rec["signal"].astype(int), rec["digitisation"], rec["offset"], rec["range"])[:-1]
        segs = ora.get_segs(ora.scale_outliers(pa, 0, 900))
        if segs:
            want.append("%s\t%s" % (rec["read_id"], ",".join(str(v) for p in segs for v in p)))
    assert got.strip().split("\n") == want and len(want) > R // 2
    slow, _, code = run_cli(smain, ["--blow5", path, "-u"])              # per-read checks: the record-by-record route
    assert code == 0 and slow == got
    # odd channel constants: a range whose two-decimal cut matters, a float offset
    from squigglekit_amd import api
    calib = np.array([[8192.0, 10.0, 1467.6149], [2048.0, -3.5, 748.58496], [8192.0, 0.0, 1200.005]])
    lens = np.array([M, M - 7, 1], dtype=np.int32)
    segs, nsegs = api.segment_batch_pa(sig[:3], lens, calib)
    for r in range(3):
        pa = blow5.to_pA(sig[r, :lens[r]].astype(int), *[calib[r][k] for k in (0, 1, 2)])
        w = ora.get_segs(ora.scale_outliers(pa, 0, 900)) or []
        assert segs[r, :nsegs[r]].tolist() == w, r
