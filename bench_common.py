"""bench_common.py -- constants and small helpers shared by bench.py (the driver's contract: timed steps, roofline,
cpu_baseline, parity) and bench_extras.py (the N = 1 measurement programmes beyond it)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_F64_LANEOPS = 256 * 4 * 16 * 2.4e9   # 3.93e13 f64 add/min/cmp lane-ops per second
WAVE_ISSUE_SLOTS = 256 * 4 * 64 * 2.4e9   # lane-results per second if every SIMD issued a wave64 VALU op per cycle
HIT_BYTES = 24
MAX_SEGS = 16


def strided_rows(total, want, run=8):
    """About `want` row indices spread over the whole batch: runs of `run` consecutive rows at evenly spaced
    positions, first and last rows included."""
    nruns = max(2, want // run)
    starts = np.unique(np.linspace(0, max(0, total - run), nruns).astype(np.int64))
    idx = (starts[:, None] + np.arange(run)[None, :]).ravel()
    return np.unique(idx[(idx >= 0) & (idx < total)])


def download_rows(L, d_base, row_bytes, rows, dtype, row_items, run=8):
    """Rows `rows` (sorted) of a device array -> numpy; consecutive rows travel in one copy."""
    from squigglekit_amd._lib import check
    base = C.cast(d_base, C.c_void_p).value
    out = np.empty((len(rows), row_items), dtype=dtype)
    k = 0
    while k < len(rows):
        j = k
        while j + 1 < len(rows) and rows[j + 1] == rows[j] + 1 and j + 1 - k < 4096:
            j += 1
        view = out[k:j + 1]
        check(L.sk_dev_download(view.ctypes.data_as(C.c_void_p), C.c_void_p(base + int(rows[k]) * row_bytes),
                                view.nbytes))
        k = j + 1
    return out


def workload_name(kind, reads, samples, motif, scaling, scale="medmad"):
    """BASELINE.json's config label when the sizes are one of its configs, "custom" otherwise."""
    per = "per GPU" if scaling == "weak" else "in total"
    if kind == "motifseq":
        tag = {(1_000_000, 4000, 200): "C4", (10_000, 4000, 163): "C3", (100_000, 20_000, 500): "C5"}.get(
            (reads, samples, motif), "custom")
        return "MotifSeq %s: %d reads x %d int16 samples %s, %d-pt motif, %s" % (tag, reads, samples, per, motif, scale)
    tag = {(10_000, 4000): "C2", (1_000_000, 4000): "C2-1M"}.get((reads, samples), "custom")
    return "segmenter %s: %d reads x %d int16 samples %s, default flags" % (tag, reads, samples, per)


def counters_from_profiles(pattern, launch_pick=None):
    """The committed counter passes of the paths the headline does not take (tools/prof_other_sq.sh -> profiles/
    sq1_other_paths.json: SQ_INSTS_VALU / kernel cycles, profiles/traffic_other_paths.json: FETCH_SIZE + WRITE_SIZE per
    launch): a bench run cannot read hardware counters itself.  Returns {"valu_issue_frac", "hbm_bytes_per_launch", "source"}
    for the first kernel whose name contains `pattern` (None where a file or the kernel is missing)."""
    import json
    out = {"valu_issue_frac": None, "hbm_bytes_per_launch": None, "source": None}
    try:
        sq = json.load(open(os.path.join(ROOT, "profiles", "sq1_other_paths.json")))
        key = [k for k in sq["kernels"] if pattern in k]
        if key:
            out["valu_issue_frac"] = sq["kernels"][key[0]].get("valu_busy_at_4_cycles")
            out["source"] = "profiles/sq1_other_paths.json (%s): VALU instructions x 4 issue cycles / (SIMDs x kernel cycles)" % key[0]
    except Exception:                                                 # noqa: BLE001 -- no committed pass: fields stay None
        pass
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic_other_paths.json")))
        key = [k for k in tr["kernels"] if pattern in k]
        if key:
            kk = tr["kernels"][key[0]]
            out["hbm_bytes_per_launch"] = kk["fetch_bytes_per_launch"] + kk["write_bytes_per_launch"]
    except Exception:                                                 # noqa: BLE001
        pass
    return out


def roof_with_counters(alg_bytes, secs, kernel_ms, pattern):
    """roofline object of an other_paths entry: the HBM view of the whole step, and -- from the committed counter passes
    of the same command -- the dominant kernel's VALU issue fraction and its HBM traffic against the algorithmic bytes"""
    c = counters_from_profiles(pattern)
    hbm = alg_bytes / secs / 1e9 / HBM_PEAK_GBS
    out = {"bound": "hbm", "achieved": alg_bytes / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm,
           "algorithmic_bytes_per_step": alg_bytes, "dominant_kernel": pattern,
           "dominant_kernel_hbm_frac": (alg_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kernel_ms and kernel_ms > 0 else None,
           "valu_issue_frac": c["valu_issue_frac"],
           "traffic": c["hbm_bytes_per_launch"],
           "traffic_ratio": (c["hbm_bytes_per_launch"] / alg_bytes) if c["hbm_bytes_per_launch"] else None,
           "counters_source": c["source"]}
    if c["valu_issue_frac"] is not None:
        out["binding"] = "valu issue" if c["valu_issue_frac"] >= max(0.5, 2 * hbm) else ("hbm" if hbm >= 0.5 else "latency / occupancy: neither pipe is half busy")
    return out
