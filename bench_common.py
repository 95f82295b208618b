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
