"""bench_extras.py -- what `python bench.py` measures beyond the driver's contract, at N = 1, after the timed region: the
drop-in command-line tools end to end (`cli`), the headline on less friendly data (`sensitivity`), the reads-per-call
sweep behind `predicted_strong_scaling` (`sweep`), and the paths the headline does not take -- float64 pA reads at 4 000,
20 000 and 37 000 samples, int16 zscale, four motifs, both dRNA branches (`other_paths`).  Each block is device resident,
kernels only unless it says otherwise, with a roofline and an oracle sample; each goes out as a JSON line of its own in
front of the headline (bench.py main).  Split out of bench.py in round 5 (it had grown to six programmes in one file)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

from bench_common import (HBM_PEAK_GBS, HIT_BYTES, MAX_SEGS, ROOT, WAVE_ISSUE_SLOTS, download_rows, strided_rows,
                          workload_name,
                          counters_from_profiles, roof_with_counters)


def cli_block(a, L, main):
    """N = 1 extras: the drop-in command-line tools end to end, process start and text output included, on the
    batch's own reads: all of them (up to 1 M) as a packed int16 .npy (--i16) and as a BLOW5 file (--blow5), 200 000
    lines (3.2 GB) of the SquigglePull-style TSV the reference reads (-s; 256 distinct reads cycled: the tokenizer does
    not care).
    tools/cli_throughput.py is the same thing stand-alone (profiles/r03_cli_throughput.txt)."""
    import shutil
    import subprocess
    import tempfile
    from squigglekit_amd import fastio
    from squigglekit_amd._lib import check, ptr
    d = tempfile.mkdtemp()
    out = {"note": "wall clock of the whole process (interpreter start, HIP start-up, ingest, kernels, text out), best "
                   "of 2; ~0.1 s of interpreter start and ~0.1 s of process exit are in every figure, the HIP start-up "
                   "(~0.25 s) runs beside the first chunks"}
    try:
        Rp = min(main.R, 1_000_000)
        host = np.empty((Rp, main.stride), dtype=np.int16)
        check(L.sk_dev_download(ptr(host), main.d_sig, host.nbytes))
        reads = host[:, :main.M]
        model = os.path.join(ROOT, "tests", "golden", "CATCTATCCAGGGTTAAATT.model")
        seg, mot = os.path.join(ROOT, "segmenter.py"), os.path.join(ROOT, "MotifSeq.py")

        def timed_runs(runs):
            for label, n, cmd in runs:
                best, lines = None, 0
                for _ in range(2):
                    t0 = time.perf_counter()
                    p = subprocess.run([sys.executable] + cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300)
                    dt = time.perf_counter() - t0
                    if p.returncode != 0:
                        best = None
                        break
                    best = dt if best is None else min(best, dt)
                    lines = p.stdout.count(b"\n")
                    del p
                out[label] = {"reads": n, "seconds": best, "reads_per_s": n / best if best else None, "output_lines": lines}

        # one input file at a time on the scratch disk (8 GB each at C4)
        f = os.path.join(d, "r.npy")
        np.save(f, reads)
        timed_runs((("segmenter_i16", Rp, [seg, "--i16", f]), ("motifseq_i16", Rp, [mot, "--i16", f, "-m", model])))
        os.remove(f)
        f = os.path.join(d, "r.blow5")
        fastio.write_blow5(f, reads)
        timed_runs((("segmenter_blow5", Rp, [seg, "--blow5", f, "--raw_signal"]),
                    ("motifseq_blow5", Rp, [mot, "--blow5", f, "-m", model])))
        os.remove(f)
        Rt = min(Rp, 200_000)
        texts = ["\t".join(str(v) for v in reads[r].tolist()) for r in range(min(Rt, 256))]
        reads256 = np.array(reads[:min(Rt, 256)])
        del host, reads
        for label, ncols, cmd in (("segmenter_tsv", 4, [seg, "-s"]), ("motifseq_tsv", 8, [mot, "-m", model, "-s"])):
            f = os.path.join(d, label)
            with open(f, "w") as fh:
                for r in range(Rt):
                    fh.write("\t".join(["read%d.fast5" % r, "id%d" % r] + ["x"] * (ncols - 2)) + "\t"
                             + texts[r % len(texts)] + "\n")
            timed_runs(((label, Rt, cmd + [f]),))
            os.remove(f)
        # ... and as SquigglePull writes them by default: pA values with two decimals (float64 route, 100 000 lines = 2.5 GB)
        Rq = min(Rt, 100_000)
        pa = np.round((reads256.astype(np.int64) + PA_OFFSET) * (PA_RANGE / PA_DIGITISATION), 2)
        texts = ["\t".join(repr(float(v)) for v in pa[r]) for r in range(pa.shape[0])]
        for label, ncols, cmd in (("segmenter_tsv_pA", 4, [seg, "-s"]), ("motifseq_tsv_pA", 8, [mot, "-m", model, "-s"])):
            f = os.path.join(d, label)
            with open(f, "w") as fh:
                for r in range(Rq):
                    fh.write("\t".join(["read%d.fast5" % r, "id%d" % r] + ["x"] * (ncols - 2)) + "\t"
                             + texts[r % len(texts)] + "\n")
            timed_runs(((label, Rq, cmd + [f]),))
            os.remove(f)
    except Exception as e:                                            # noqa: BLE001 -- report, keep the line
        out["error"] = repr(e)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


def sensitivity_block(a, L, main):
    """N = 1, after everything else: what the headline is worth on less friendly data.  (i) reads that are windows of
    the one measured squiggle the reference ships (example/slow5/0.blow5, 36 978 samples; copy under tests/golden)
    plus N(0, 3) noise, against the example model (163 points) and the synthetic 200-point motif; (ii) the C4 batch
    with a given share of the reads forced through the exact retry (what a retry rate of x % costs); (iii) the C4
    batch with half of the reads also carrying the motif stretched 2 / 3 / 4 times in time (wide optimal paths: second
    tier of the window pass, then the retry).  Three steps each, best taken."""
    from squigglekit_amd import blow5
    from squigglekit_amd._lib import check, ptr
    out = {"reads": main.R, "note": "HBM-resident, kernels only (as the headline); ms = best of 3 steps"}

    def run(motif):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            check(L.sk_motifseq_dev_i16(main.d_sig, main.stride, main.d_len, main.R, ptr(motif), motif.size,
                                        main.mode, 0, 1200, main.d_out))
            check(L.sk_sync())
            ts.append(time.perf_counter() - t0)
        return {"reads_per_s": main.R / min(ts), "ms": min(ts) * 1e3, "retried_reads": int(L.sk_last_dtw_retries()),
                "second_tier_reads": int(L.sk_last_dtw_tier2())}

    try:
        read = next(blow5.read_blow5(os.path.join(ROOT, "tests", "golden", "example_0.blow5")))
        raw = np.asarray(read["signal"], dtype=np.int16)
        import gzip
        with gzip.open(os.path.join(ROOT, "tests", "golden", "motifseq_cli.json.gz"), "rt") as fh:
            model163 = np.array(json.load(fh)["model_expanded"]["values"], dtype=np.float64)
        main.regenerate(tmpl=raw, tmpl_noise=3.0)
        out["real_signal_windows"] = {
            "source": "tests/golden/example_0.blow5 (%d samples): random %d-sample windows + N(0,3) noise" % (raw.size, main.M),
            "vs_example_model_163pt": run(model163),
            "vs_synthetic_%dpt_motif" % main.N: run(main.motif)}
    except Exception as e:                                            # noqa: BLE001 -- report, keep the line
        out["real_signal_windows"] = {"error": repr(e)}
    # (ii) what a retry costs: a given share of the reads is sent to the exact single pass whatever the window pass
    # found (SK_DTW_FORCE_RETRY_PM, a switch of the window kernel for exactly this measurement) -- on the C4 batch
    main.regenerate()
    sweep = {}
    for pm in (0, 10, 100, 500):
        os.environ["SK_DTW_FORCE_RETRY_PM"] = str(pm)
        try:
            sweep["%g%%" % (pm / 10.0)] = run(main.motif)
        finally:
            del os.environ["SK_DTW_FORCE_RETRY_PM"]
    base = sweep["0%"]["reads_per_s"]
    for v in sweep.values():
        v["vs_no_retries"] = v["reads_per_s"] / base
    out["forced_retry_sweep"] = {"what": "C4 batch; this share of the reads (by hash of the read index) takes the exact "
                                         "single-pass retry regardless of what the window pass certified",
                                 "by_share": sweep}
    # (iii) data that produces wide paths by itself: half of the reads also carry the motif stretched k times in
    # time (k x N samples); a match wider than the first look-back goes to the second tier, wider than that to the retry
    wide = {}
    for k in (2, 3, 4):
        main.regenerate(stretch_permille=500, stretch=k)
        wide["x%d" % k] = run(main.motif)
    out["stretched_motif_in_half_of_the_reads"] = wide
    main.regenerate()                                                 # the default batch again
    return out


def sweep_block(a, L, main):
    """N = 1: the headline's kernels on 1 M / 500 k / 250 k / 125 k / 62.5 k / 31 250 of the resident reads per call --
    what each GPU sees when C4 (1 M reads IN TOTAL) is block-sharded over 1 / 2 / 4 / 8 / 16 / 32 GPUs -- for MotifSeq and
    for the segmenter.  No multi-GPU node was available to any round so far: this is the one-GPU prediction of the
    strong-scaling curve (the per-rank work is exactly a call of that size; what it leaves out is the 24 B/read
    all-gather, 3 MB per GPU at N = 8, and PCIe ingest, which bench.py's end_to_end block times).  Fit: ms = fixed + per_read * R
    over the sizes; predicted efficiency at N GPUs = rate(R / N) / rate(R)."""
    from squigglekit_amd._lib import SegParams, check, ptr
    sizes = [main.R // d for d in (1, 2, 4, 8, 16, 32) if main.R // d >= 4096]

    def fit(rows):
        x = np.array([r["reads_per_call"] for r in rows], dtype=np.float64)
        y = np.array([r["ms"] for r in rows], dtype=np.float64)
        b, c = np.polyfit(x, y, 1)
        return {"fixed_ms_per_call": float(c), "us_per_1000_reads": float(b * 1e6),
                "note": "least-squares line through (reads per call, ms)"}

    def run(call, cells_per_read=None, bytes_per_read=None):
        rows = []
        clk = None
        for R in sizes:
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                call(R)
                check(L.sk_sync())
                ts.append(time.perf_counter() - t0)
            t = min(ts[1:])
            row = {"reads_per_call": R, "ms": t * 1e3, "reads_per_s": R / t}
            if cells_per_read:
                ghz = C.c_double(0.0)
                L.sk_last_dtw_clock(C.byref(ghz))
                clk = ghz.value if 0.5 < ghz.value < 3.0 else clk
                roof = WAVE_ISSUE_SLOTS / 8.0 * ((clk or 2.4) / 2.4)
                row["issue_roof_frac"] = R * cells_per_read / t / roof
            if bytes_per_read:
                row["hbm_frac"] = R * bytes_per_read / t / 1e9 / HBM_PEAK_GBS
            rows.append(row)
        base = rows[0]["reads_per_s"]
        for r in rows:
            r["vs_full_batch_rate"] = r["reads_per_s"] / base
        return rows

    out = {"note": sweep_block.__doc__.split("\n\n")[0].replace("\n    ", " ")}
    if main.kind == "motifseq":
        rows = run(lambda R: check(L.sk_motifseq_dev_i16(main.d_sig, main.stride, main.d_len, R, ptr(main.motif), main.N,
                                                          main.mode, 0, 1200, main.d_out)),
                   cells_per_read=float(main.N) * main.M)
        out["motifseq"] = {"by_reads_per_call": rows, "fit": fit(rows)}
        out["predicted_strong_scaling"] = {
            "what": "C4 (%d reads in total) over N GPUs from one GPU's rate at %d / N reads per call; gather and ingest not included" % (main.R, main.R),
            "efficiency": {str(main.R // r["reads_per_call"]): r["vs_full_batch_rate"] for r in rows},
            "reads_per_s": {str(main.R // r["reads_per_call"]): r["reads_per_s"] * (main.R // r["reads_per_call"]) for r in rows}}
        # ... with the one exchange put back in as a MODEL (it has never run on more than one rank): an all-gather of
        # 24 B per read in total, every rank receiving (N - 1) / N of it, priced at one xGMI link's ~48 GB/s per direction
        # (a third of the 153 GB/s the guide quotes, ring-style, no overlap with the kernels) plus 30 us of launch /
        # synchronisation latency per step.  The measured rate of this is what SCALE_rNN.json is for.
        link_gbs, lat_ms = 48.0, 0.030
        eff = {}
        for r in rows:
            n = main.R // r["reads_per_call"]
            gather_ms = 0.0 if n == 1 else lat_ms + (main.R * 24.0 * (n - 1) / n) / (link_gbs * 1e9) * 1e3
            eff[str(n)] = (rows[0]["ms"] / n) / (r["ms"] + gather_ms)
        out["predicted_strong_scaling"]["efficiency_with_gather_model"] = eff
        out["predicted_strong_scaling"]["gather_model"] = (
            "all-gather of 24 B/read priced at %.0f GB/s per rank + %.0f us per step, not overlapped; ingest (PCIe) is per GPU "
            "and does not enter a device-resident step -- bench.py --gpus N times it on every rank at once (end_to_end)"
            % (link_gbs, lat_ms * 1e3))
    return out


def sweep_segmenter(L, w):
    """the segmenter leg of sweep_block, on a resident segmenter workload"""
    from squigglekit_amd._lib import check
    sizes = [w.R // d for d in (1, 2, 4, 8, 16, 32) if w.R // d >= 4096]
    rows = []
    for R in sizes:
        ts = []
        for _ in range(6):
            t0 = time.perf_counter()
            check(L.sk_segment_dev_i16(w.d_sig, w.stride, w.d_len, R, C.byref(w.sp), w.d_segs, w.d_out, MAX_SEGS))
            check(L.sk_sync())
            ts.append(time.perf_counter() - t0)
        t = min(ts[2:])
        rows.append({"reads_per_call": R, "ms": t * 1e3, "reads_per_s": R / t,
                     "hbm_frac": R * (2 * w.M + 4 + 16) / t / 1e9 / HBM_PEAK_GBS})
    for r in rows:
        r["vs_full_batch_rate"] = r["reads_per_s"] / rows[0]["reads_per_s"]
    x = np.array([r["reads_per_call"] for r in rows], dtype=np.float64)
    y = np.array([r["ms"] for r in rows], dtype=np.float64)
    b, c = np.polyfit(x, y, 1)
    return {"by_reads_per_call": rows, "fit": {"fixed_ms_per_call": float(c), "us_per_1000_reads": float(b * 1e6)}}


PA_OFFSET, PA_RANGE, PA_DIGITISATION = 16.0, 1493.94, 8192.0      # channel constants of the pA image (as tests/test_gpu_f64.py)


def other_paths_block(a, L, main):
    """N = 1 extras: the paths the headline does not take, each device resident, kernels only, with a roofline and a
    parity sample against the oracle.
      segmenter_f64_pA   -- the segmenter on float64 pA reads (segmenter.py:198-201; the values SquigglePull.py:183-189
                            writes for the batch's reads: np.round((raw + offset) * range / digitisation, 2))
      motifseq_f64_medmad-- MotifSeq on the same float64 reads (MotifSeq.py:270 parses every sample as float)
      motifseq_i16_zscale-- MotifSeq -l zscale on the C4 batch (MotifSeq.py:186-191,275-280)
      motifseq_multi_k4  -- four motifs against the C4 batch (the `for name in m_order` loop, MotifSeq.py:436)"""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as ora
    from squigglekit_amd import synth
    from squigglekit_amd._lib import HIT_DTYPE, SegParams, check, ptr
    out = {"note": "HBM resident, kernels only (as the headline): wall clock of the best of 3 steps after a warm-up, "
                   "HIP-event kernel times beside it"}
    M, N = main.M, main.N
    T = max(1, min(32, os.cpu_count() or 1))
    nominal = WAVE_ISSUE_SLOTS / 8.0

    def best_of(fn, n=3):
        fn()
        check(L.sk_sync())
        ts, ev = [], None
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            check(L.sk_sync())
            ts.append(time.perf_counter() - t0)
            if ts[-1] == min(ts):
                ev = main.kernel_ms()
        return min(ts), ev

    def dtw_view(R, cells_per_read, secs):
        ach = R * cells_per_read / secs
        return {"bound": "valu_issue", "achieved": ach / 1e12, "unit": "T cell-updates/s",
                "peak_at_2.4_ghz": nominal / 1e12, "frac_at_2.4_ghz": ach / nominal,
                "note": "whole step against the screening pass's 8-issue-cycles-per-cell roof (DESIGN.md 4.3)"}

    bufs = []

    def alloc(nbytes):
        q = L.sk_dev_alloc(nbytes)
        if not q:
            check(-4)
        bufs.append(q)
        return q

    try:
        # ---------------- float64 pA reads: segmenter and MotifSeq --------------------------------------------
        Rf = min(main.R, 500_000)                                    # (16 GB: enough wavefronts for the lane-per-read walk to fill the chip)
        Mf = M - 1                                                    # segmenter.py:207 with the default -n: sig[:-1]
        total = Rf * Mf
        d_pa, d_off = alloc(total * 8), alloc((Rf + 1) * 8)
        check(L.sk_synth_pa_dev(main.d_sig, main.stride, Rf, Mf, PA_OFFSET, PA_RANGE, PA_DIGITISATION, d_pa, d_off))
        d_segs, d_nsegs = alloc(Rf * MAX_SEGS * 2 * 4), alloc(Rf * 4)
        sp = SegParams()
        secs, ev = best_of(lambda: check(L.sk_segment_dev_f64(d_pa, d_off, Rf, total, Mf, C.byref(sp), d_segs, d_nsegs,
                                                              MAX_SEGS)))
        retried = L.sk_last_f64_retries()
        rows = strided_rows(Rf, 512)
        pa = download_rows(L, d_pa, Mf * 8, rows, np.float64, Mf)
        segs = np.empty((Rf, MAX_SEGS, 2), dtype=np.int32)
        nsegs = np.empty(Rf, dtype=np.int32)
        check(L.sk_dev_download(ptr(segs), d_segs, segs.nbytes))
        check(L.sk_dev_download(ptr(nsegs), d_nsegs, nsegs.nbytes))
        op = ora.SegParams(sp.error, sp.corrector, sp.window, sp.seg_dist, sp.std_scale, sp.stall_len)

        def seg_ok(k):
            want = ora.get_segs(ora.scale_outliers(pa[k], sp.lim_low, sp.lim_hi), op) or []
            r = rows[k]
            return nsegs[r] == len(want) and segs[r, :nsegs[r]].tolist() == want
        with ThreadPoolExecutor(T) as ex:
            ok = all(ex.map(seg_ok, range(len(rows))))
        alg = Rf * (8 * Mf + 4 + 16)
        out["segmenter_f64_pA"] = {
            "workload": "%d reads x %d float64 pA samples (2 decimals), default flags" % (Rf, Mf),
            "value": Rf / secs, "unit": "reads/s", "ms_per_step": secs * 1e3,
            "kernel_ms": {"statistics": ev[0], "walk": ev[1]}, "reads_redone_in_numpy_order": int(retried),
            "roofline": {"bound": "hbm", "achieved": alg / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / secs / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg,
                         "statistics_kernel_frac": (alg / (ev[0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ev[0] > 0 else None},
            "parity": {"reads_checked": int(len(rows)), "segments_bit_exact": bool(ok),
                       "segments_in_sample": int(nsegs[rows].sum())}}

        # ---------------- the same reads as raw int16 rows + channel constants: the raw-domain pA route (round 6) -------------
        def raw_pa_entry(d_raw, stride_, R_, M_, max_segs_, segs_f64, nsegs_f64, f64_secs):
            """sk_segment_dev_i16_pa over the rows the float64 batch was made from (segmenter.py:345-349 with fast5 / slow5
            input): EVERY record against the float64 path's (itself sampled against the oracle above)"""
            lens_ = np.full(R_, M_, dtype=np.int32)
            cal3 = np.tile(np.array([PA_DIGITISATION, PA_OFFSET, PA_RANGE]), (R_, 1))
            cal2 = np.empty((R_, 2))
            check(L.sk_pa_calib(ptr(cal3), R_, ptr(cal2)))
            d_len_, d_cal_ = alloc(R_ * 4), alloc(R_ * 16)
            check(L.sk_dev_upload(d_len_, ptr(lens_), lens_.nbytes))
            check(L.sk_dev_upload(d_cal_, ptr(cal2), cal2.nbytes))
            d_s, d_n = alloc(R_ * max_segs_ * 2 * 4), alloc(R_ * 4)
            secs_, ev_ = best_of(lambda: check(L.sk_segment_dev_i16_pa(d_raw, stride_, d_len_, R_, d_cal_, C.byref(sp), d_s, d_n,
                                                                      max_segs_)))
            redone = int(L.sk_last_pa_retries())
            s_ = np.empty((R_, max_segs_, 2), dtype=np.int32)
            n_ = np.empty(R_, dtype=np.int32)
            check(L.sk_dev_download(ptr(s_), d_s, s_.nbytes))
            check(L.sk_dev_download(ptr(n_), d_n, n_.nbytes))
            same = bool(np.array_equal(n_, nsegs_f64) and np.array_equal(s_, segs_f64))
            for q in (d_len_, d_cal_, d_s, d_n):
                L.sk_dev_free(q)
                bufs.remove(q)
            alg_ = R_ * (2 * M_ + 4 + 16 + 16)
            return {"workload": "%d raw int16 reads x %d samples + channel constants -> pA in the raw domain "
                                "(segmenter.py:345-349), default flags" % (R_, M_),
                    "value": R_ / secs_, "unit": "reads/s", "ms_per_step": secs_ * 1e3,
                    "speedup_vs_float64_route": f64_secs / secs_,
                    "kernel_ms": {"statistics": ev_[0], "walk": ev_[1]}, "reads_redone_from_float64": redone,
                    "roofline": {"bound": "hbm", "achieved": alg_ / secs_ / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": alg_ / secs_ / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg_,
                                 "bytes_note": "2 B per sample resident (the float64 route: 8)",
                                 "statistics_kernel_valu_issue_frac": counters_from_profiles(
                                     "k_seg_stats<8, 8, 8, true, true" if stride_ > 4096 else "k_seg_stats<8, 8, 8, false, true")["valu_issue_frac"],
                                 "walk_kernel_valu_issue_frac": counters_from_profiles("k_seg_walkL" if stride_ > 4096 else "k_seg_walk4<6, true")["valu_issue_frac"],
                                 "counters_source": "profiles/sq1_other_paths.json",
                                 "statistics_kernel_frac": (alg_ / (ev_[0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ev_[0] > 0 else None},
                    "parity": {"reads_checked": int(R_), "all_records_equal_float64_route": same,
                               "float64_route_vs_oracle": "sampled above"}}
        out["segmenter_raw_pA"] = raw_pa_entry(main.d_sig, main.stride, Rf, Mf, MAX_SEGS, segs, nsegs, secs)

        # ---------------- the same at real read lengths: 20 000 samples (C5-shaped) and 36 977 (the one measured read the
        # reference ships, example/slow5/0.blow5) -- segmenter AND MotifSeq: `MotifSeq.py --signal` parses every sample as a
        # float (MotifSeq.py:270), so this, not the int16 headline, is what the reference's default input looks like
        import gzip
        with gzip.open(os.path.join(ROOT, "tests", "golden", "motifseq_cli.json.gz"), "rt") as fh:
            model163 = np.array(json.load(fh)["model_expanded"]["values"], dtype=np.float64)
        for RL, ML, tag in ((50_000, 20_000, "20k"), (25_000, 36_978, "37k")):
            MLs = (ML + 7) // 8 * 8
            d_raw_l = alloc(RL * MLs * 2)
            check(L.sk_synth_squiggles_dev(d_raw_l, MLs, RL, ML, synth.SEED_C5, None, 0))
            MLf = ML - 1
            d_pa_l, d_off_l = alloc(RL * MLf * 8), alloc((RL + 1) * 8)
            check(L.sk_synth_pa_dev(d_raw_l, MLs, RL, MLf, PA_OFFSET, PA_RANGE, PA_DIGITISATION, d_pa_l, d_off_l))
            MAXS_L = 128
            d_segs_l, d_nsegs_l = alloc(RL * MAXS_L * 2 * 4), alloc(RL * 4)
            secs, ev = best_of(lambda: check(L.sk_segment_dev_f64(d_pa_l, d_off_l, RL, RL * MLf, MLf, C.byref(sp), d_segs_l,
                                                                  d_nsegs_l, MAXS_L)))
            retried_l = int(L.sk_last_f64_retries())
            rows_l = strided_rows(RL, 128)
            pa_l = download_rows(L, d_pa_l, MLf * 8, rows_l, np.float64, MLf)
            segs_l = np.empty((RL, MAXS_L, 2), dtype=np.int32)
            nsegs_l = np.empty(RL, dtype=np.int32)
            check(L.sk_dev_download(ptr(segs_l), d_segs_l, segs_l.nbytes))
            check(L.sk_dev_download(ptr(nsegs_l), d_nsegs_l, nsegs_l.nbytes))

            def seg_ok_l(k):
                want = ora.get_segs(ora.scale_outliers(pa_l[k], sp.lim_low, sp.lim_hi), op) or []
                r = rows_l[k]
                return nsegs_l[r] == len(want) and segs_l[r, :nsegs_l[r]].tolist() == want
            with ThreadPoolExecutor(T) as ex:
                ok_l = all(ex.map(seg_ok_l, range(len(rows_l))))
            alg = RL * (8 * MLf + 4 + 16)
            out["segmenter_f64_pA_%s" % tag] = {
                "workload": "%d reads x %d float64 pA samples (2 decimals), default flags" % (RL, MLf),
                "value": RL / secs, "unit": "reads/s", "ms_per_step": secs * 1e3,
                "kernel_ms": {"statistics": ev[0], "walk": ev[1]}, "reads_redone_in_numpy_order": retried_l,
                "roofline": {"bound": "hbm", "achieved": alg / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": alg / secs / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg,
                             "statistics_kernel_frac": (alg / (ev[0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ev[0] > 0 else None},
                "parity": {"reads_checked": int(len(rows_l)), "segments_bit_exact": bool(ok_l),
                           "segments_in_sample": int(nsegs_l[rows_l].sum())}}
            out["segmenter_raw_pA_%s" % tag] = raw_pa_entry(d_raw_l, MLs, RL, MLf, MAXS_L, segs_l, nsegs_l, secs)
            # MotifSeq, float64 medmad, against the example model (163 points)
            d_hits_l = alloc(RL * HIT_BYTES)
            secs, ev = best_of(lambda: check(L.sk_motifseq_dev_f64(d_pa_l, d_off_l, RL, RL * MLf, MLf, ptr(model163),
                                                                   model163.size, 0, 0, 1200, d_hits_l)))
            g = (C.c_int32 * 8)()
            check(L.sk_last_dtw_guard(g))
            hits_l = np.empty(RL, dtype=HIT_DTYPE)
            check(L.sk_dev_download(ptr(hits_l), d_hits_l, hits_l.nbytes))
            rows_m = rows_l[::4]
            pa_m = pa_l[::4]

            def want_long(k):
                y = ora.medmad(ora.scale_outliers(pa_m[k], 0, 1200))[0]
                return ora.dtw_subsequence(model163, y)
            with ThreadPoolExecutor(T) as ex:
                want_m = list(ex.map(want_long, range(len(rows_m))))
            got_m = hits_l[rows_m]
            alg = RL * (8 * MLf + HIT_BYTES)
            cells = float(model163.size) * float(np.mean(hits_l["n"]))
            out["motifseq_f64_medmad_%s" % tag] = {
                "workload": "%d reads x %d float64 pA samples vs the example model (%d points), medmad" % (RL, MLf, model163.size),
                "value": RL / secs, "unit": "reads/s", "ms_per_step": secs * 1e3,
                "kernel_ms": {"prep": ev[0], "dtw": ev[1]},
                "guard": {"premise_violations": int(g[0]), "audited_reads": int(g[1]), "audit_mismatches": int(g[2]),
                          "second_windows": int(g[6])},
                "roofline": {"bound": "hbm", "achieved": alg / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": alg / secs / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg,
                             "prep_kernel_frac": (alg / (ev[0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ev[0] > 0 else None,
                             "valu": dtw_view(RL, cells, secs),
                             "screening_pass_valu_issue_frac": counters_from_profiles(
                                 "k_sdtw_q<8, 21, 1" if RL >= 65536 else "k_sdtw_q<16, 11, 1")["valu_issue_frac"],
                             "counters_source": "profiles/sq1_other_paths.json"},
                "parity": {"reads_checked": int(len(rows_m)),
                           "dist_bit_identical": bool(all(got_m["dist"][k] == w[0] for k, w in enumerate(want_m))),
                           "start_end_exact": bool(all((got_m["start"][k], got_m["end"][k]) == (w[1], w[2])
                                                       for k, w in enumerate(want_m)))}}
            for q in (d_raw_l, d_pa_l, d_off_l, d_segs_l, d_nsegs_l, d_hits_l):
                L.sk_dev_free(q)
                bufs.remove(q)

        # ---------------- dRNA_segmenter.py, both branches, device resident at a size that fills the chip ----------------
        # (round 4 timed 20 000 reads = 313 wavefronts on 1 024 SIMDs: one wavefront's latency.  250 000 reads, 15 GB.)
        from squigglekit_amd import api
        from squigglekit_amd._lib import DrnaParams, RollParams
        # dRNA-shaped reads (synth.drna_reads: adapter stretch, poly(A) plateau, body; 6 000 .. 30 000 samples): 1 000
        # distinct ones, tiled -- the scans stop where the script stops ("adapter found"), which generic squiggles never reach
        base_reads = synth.drna_reads(1000, synth.SEED_C5 + 7, min_len=6000, max_len=30000)
        RD, MD = 250_000, 30_000
        NB_ = len(base_reads)
        host_b = api.pinned_empty((NB_, MD), np.int16)
        host_b[:] = 0
        lens_b = np.zeros(NB_, dtype=np.int32)
        for r, x in enumerate(base_reads):
            host_b[r, :x.size] = x
            lens_b[r] = x.size
        d_sig_d, d_len_d = alloc(RD * MD * 2), alloc(RD * 4)
        lens_d = np.tile(lens_b, RD // NB_)
        check(L.sk_dev_upload(d_len_d, ptr(lens_d), lens_d.nbytes))
        base_p = C.cast(d_sig_d, C.c_void_p).value
        for k in range(RD // NB_):                                    # the 1 000 distinct reads, 250 times
            check(L.sk_dev_upload(C.c_void_p(base_p + k * NB_ * MD * 2), ptr(host_b), host_b.nbytes))
        dp, rp = DrnaParams(), RollParams()
        d_dsegs, d_dn = alloc(RD * 32 * 2 * 4), alloc(RD * 4)
        d_xy, d_found = alloc(RD * 2 * 4), alloc(RD * 4)
        secs, ev = best_of(lambda: check(L.sk_drna_segment_dev_i16(d_sig_d, MD, d_len_d, RD, C.byref(dp), d_dsegs, d_dn, 32)))
        dsegs = np.zeros((RD, 32, 2), dtype=np.int32)
        dn = np.zeros(RD, dtype=np.int32)
        check(L.sk_dev_download(ptr(dsegs), d_dsegs, dsegs.nbytes))
        check(L.sk_dev_download(ptr(dn), d_dn, dn.nbytes))
        rows_d = strided_rows(RD, 96)
        odp = ora.DrnaParams()

        def drna_ok(r):
            b = r % NB_
            want = ora.drna_segs(ora.scale_outliers(host_b[b, :lens_b[b]].astype(float), dp.lim_low, dp.lim_hi), odp)[0]
            return dsegs[r, :dn[r]].tolist() == want
        with ThreadPoolExecutor(T) as ex:
            ok1 = all(ex.map(drna_ok, rows_d))
        kms = ev[0] + ev[1]
        alg = int(2 * lens_d.astype(np.int64).sum() + RD * 12)
        out["drna_slow5_branch"] = {
            "workload": "%d dRNA-shaped reads of 6 000 .. 30 000 int16 samples (mean %d), dRNA_segmenter.py:85-176 constants; "
                        "device resident" % (RD, int(lens_d.mean())),
            "value": RD / secs, "unit": "reads/s", "ms_per_step": secs * 1e3,
            "kernel_ms": {"statistics": ev[0], "scan": ev[1]},
            "roofline": dict(roof_with_counters(alg, secs, ev[0], "k_drna_stats"),
                             kernels_only_frac=alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS),
            "parity": {"reads_checked": int(len(rows_d)), "segments_bit_exact": bool(ok1)}}
        secs, ev = best_of(lambda: check(L.sk_drna_roll_dev_i16(d_sig_d, MD, d_len_d, RD, C.byref(rp), d_xy, d_found)))
        xy = np.zeros((RD, 2), dtype=np.int32)
        found = np.zeros(RD, dtype=np.int32)
        check(L.sk_dev_download(ptr(xy), d_xy, xy.nbytes))
        check(L.sk_dev_download(ptr(found), d_found, found.nbytes))
        orp = ora.RollParams()

        def roll_ok(r):
            b = r % NB_
            want = ora.drna_roll(ora.scale_outliers(host_b[b, :lens_b[b]].astype(float), rp.lim_low, rp.lim_hi), orp)
            got = (int(xy[r, 0]), int(xy[r, 1])) if found[r] else None
            return got == want
        with ThreadPoolExecutor(T) as ex:
            ok2 = all(ex.map(roll_ok, rows_d))
        kms = ev[0] + ev[1]
        out["drna_rolling_mean_branch"] = {
            "workload": "%d dRNA-shaped reads (mean %d samples), dRNA_segmenter.py:272-326, w = 2000; device resident"
                        % (RD, int(lens_d.mean())),
            "value": RD / secs, "unit": "reads/s", "ms_per_step": secs * 1e3,
            "kernel_ms": {"filter_prefix_sums_statistics": ev[0], "scan": ev[1]},
            "roofline": dict(roof_with_counters(alg, secs, ev[0], "k_roll_stream"),
                             kernels_only_frac=alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS),
            "parity": {"reads_checked": int(len(rows_d)), "pairs_exact": bool(ok2), "found_in_sample": int(found[rows_d].sum())}}
        del host_b
        for q in (d_sig_d, d_len_d, d_dsegs, d_dn, d_xy, d_found):
            L.sk_dev_free(q)
            bufs.remove(q)

        d_hits = alloc(max(Rf, main.R) * HIT_BYTES * 4)

        def hits_ok(rows_, got, want_fn):
            with ThreadPoolExecutor(T) as ex:
                want = list(ex.map(want_fn, range(len(rows_))))
            d_ok = all(got["dist"][k] == w[0] for k, w in enumerate(want))
            se_ok = all((got["start"][k], got["end"][k]) == (w[1], w[2]) for k, w in enumerate(want))
            return {"reads_checked": int(len(rows_)), "dist_bit_identical": bool(d_ok), "start_end_exact": bool(se_ok)}

        secs, ev = best_of(lambda: check(L.sk_motifseq_dev_f64(d_pa, d_off, Rf, total, Mf, ptr(main.motif), N, 0, 0, 1200,
                                                               d_hits)))
        hits = np.empty(Rf, dtype=HIT_DTYPE)
        check(L.sk_dev_download(ptr(hits), d_hits, hits.nbytes))
        rows2 = rows[::4]
        pa2 = pa[::4]

        def want_f64(k):
            y = ora.medmad(ora.scale_outliers(pa2[k], 0, 1200))[0]
            return ora.dtw_subsequence(main.motif, y)
        alg = Rf * (8 * Mf + HIT_BYTES)
        out["motifseq_f64_medmad"] = {
            "workload": "%d reads x %d float64 pA samples vs %d-pt motif, medmad" % (Rf, Mf, N),
            "value": Rf / secs, "unit": "reads/s", "ms_per_step": secs * 1e3,
            "kernel_ms": {"prep": ev[0], "dtw": ev[1]},
            "roofline": {"bound": "hbm", "achieved": alg / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / secs / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg,
                         "prep_kernel_frac": (alg / (ev[0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ev[0] > 0 else None,
                         "valu": dtw_view(Rf, float(N) * float(np.mean(hits["n"])), secs)},
            "parity": hits_ok(rows2, hits[rows2], want_f64)}
        for q in (d_pa, d_off, d_segs, d_nsegs):
            L.sk_dev_free(q)
            bufs.remove(q)

        # ---------------- int16, zscale ------------------------------------------------------------------------
        R = main.R
        secs, ev = best_of(lambda: check(L.sk_motifseq_dev_i16(main.d_sig, main.stride, main.d_len, R, ptr(main.motif), N,
                                                               1, 0, 1200, d_hits)))
        hits = np.empty(R, dtype=HIT_DTYPE)
        check(L.sk_dev_download(ptr(hits), d_hits, hits.nbytes))
        rows3 = strided_rows(R, 256)
        sample = download_rows(L, main.d_sig, main.stride * 2, rows3, np.int16, main.stride)
        parts = [(i, min(len(rows3), i + 8)) for i in range(0, len(rows3), 8)]

        def ora_i16(motif, mode):
            with ThreadPoolExecutor(T) as ex:
                return np.concatenate(list(ex.map(lambda ab: ora.motifseq_batch_i16(
                    sample[ab[0]:ab[1]], main.lens[rows3[ab[0]:ab[1]]], motif, scale_mode=mode), parts)))
        want = ora_i16(main.motif, 1)
        got = hits[rows3]
        alg = R * (2 * M + HIT_BYTES)
        out["motifseq_i16_zscale"] = {
            "workload": workload_name("motifseq", R, M, N, "weak", "zscale"),
            "value": R / secs, "unit": "reads/s", "ms_per_step": secs * 1e3,
            "kernel_ms": {"prep": ev[0], "dtw": ev[1]},
            "roofline": {"bound": "hbm", "achieved": alg / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / secs / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg,
                         "prep_kernel_frac": (alg / (ev[0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ev[0] > 0 else None,
                         "valu": dtw_view(R, float(N) * float(np.mean(hits["n"])), secs)},
            "parity": {"reads_checked": int(len(rows3)), "dist_bit_identical": bool(np.array_equal(got["dist"], want["dist"])),
                       "start_end_exact": bool(np.array_equal(got["start"], want["start"])
                                               and np.array_equal(got["end"], want["end"]))}}

        # ---------------- int16, medmad, four motifs ------------------------------------------------------------
        motifs = [main.motif] + [synth.synthetic_motif(n, seed=sd) for n, sd in ((N, 11), (max(8, N - 37), 12), (N + 40, 13))]
        flat = np.concatenate(motifs)
        moff = np.concatenate([[0], np.cumsum([m.size for m in motifs])]).astype(np.int32)
        K = len(motifs)
        secs, ev = best_of(lambda: check(L.sk_motifseq_multi_dev_i16(main.d_sig, main.stride, main.d_len, R, ptr(flat),
                                                                     ptr(moff), K, 0, 0, 1200, d_hits)))
        hits = np.empty((K, R), dtype=HIT_DTYPE)
        check(L.sk_dev_download(ptr(hits), d_hits, hits.nbytes))
        ok_d = ok_se = True
        for k in range(K):
            want = ora_i16(motifs[k], 0)
            got = hits[k][rows3]
            ok_d = ok_d and np.array_equal(got["dist"], want["dist"])
            ok_se = ok_se and np.array_equal(got["start"], want["start"]) and np.array_equal(got["end"], want["end"])
        cells = float(np.mean(hits[0]["n"])) * float(flat.size)
        out["motifseq_multi_k4"] = {
            "workload": "%d reads x %d int16 samples vs %d motifs (%s points), medmad" % (
                R, M, K, ", ".join(str(m.size) for m in motifs)),
            "value": R / secs, "unit": "reads/s", "read_motif_pairs_per_s": R * K / secs, "ms_per_step": secs * 1e3,
            "kernel_ms": {"prep": ev[0], "dtw_last_motif": ev[1]},
            "roofline": {"bound": "valu_issue", **dtw_view(R, cells, secs),
                         "hbm_frac": R * (2 * M + K * HIT_BYTES) / secs / 1e9 / HBM_PEAK_GBS},
            "parity": {"reads_checked": int(len(rows3)) * K, "dist_bit_identical": bool(ok_d), "start_end_exact": bool(ok_se)}}
    except Exception as e:                                            # noqa: BLE001 -- report, keep the line
        import traceback
        out["error"] = repr(e) + " | " + traceback.format_exc(limit=2).replace("\n", " / ")
    finally:
        for q in bufs:
            L.sk_dev_free(q)
    return out
