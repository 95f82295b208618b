"""Host-side mirror of the reference's two in-process boundaries, over the HIP C ABI.

    get_segs(sig, args)            <->  segmenter.get_segs      (segmenter.py:399)
    dtw_subsequence(x, y)          <->  mlpy.dtw_subsequence    (MotifSeq.py:437)

plus the batch forms the GPU actually wants (many reads per call).  All arithmetic
on samples happens in the HIP kernels; this module only marshals numpy buffers.
Same names, argument meaning and "no segments -> False" behaviour as the
reference so the parity tests read like calls into the reference.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import HIT_DTYPE, SegParams, SquiggleKitError, check, ptr


# ----------------------------------------------------------------------------
# device selection
# ----------------------------------------------------------------------------
_default_devices = None


def set_devices(devices):
    """GPUs the batch calls shard their reads over when no `devices=` is passed (None / one entry: the
    calling thread's bound GPU, as before).  The CLIs' --gpus N sets range(N)."""
    global _default_devices
    devices = None if devices is None else [int(d) for d in devices]
    if devices is not None:
        n = _lib.load().sk_device_count()
        if not devices or len(set(devices)) != len(devices) or min(devices) < 0 or max(devices) >= max(n, 1):
            raise ValueError("devices %s: need distinct indices below the %d visible GPU(s)" % (devices, n))
    _default_devices = devices


def _devs(devices):
    return _default_devices if devices is None else devices


# ----------------------------------------------------------------------------
# pinned host buffers
# ----------------------------------------------------------------------------
def pinned_empty(shape, dtype=np.int16):
    """A numpy array in page-locked host memory (sk_host_alloc): the batch calls copy from it by DMA at PCIe
    speed, under the kernels of the previous sub-batch.  Freed when the array (and every view of it) is gone."""
    import weakref
    L = _lib.ensure_init()
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    p = L.sk_host_alloc(max(1, n))
    if not p:
        check(-4)
    raw = (C.c_char * max(1, n)).from_address(p)
    arr = np.frombuffer(raw, dtype=dt, count=int(np.prod(shape))).reshape(shape)
    weakref.finalize(raw, L.sk_host_free, C.c_void_p(p))
    return arr


# ----------------------------------------------------------------------------
# packing helpers
# ----------------------------------------------------------------------------
def pack_i16(reads):
    """list of 1-D integer arrays -> (int16 [R, stride] zero padded, int32 lens).
    stride is a multiple of 8 so rows are 16-byte aligned (vector loads)."""
    lens = np.array([len(r) for r in reads], dtype=np.int32)
    stride = int(max(8, (int(lens.max()) + 7) // 8 * 8)) if len(reads) else 8
    buf = np.zeros((len(reads), stride), dtype=np.int16)
    for i, r in enumerate(reads):
        buf[i, :len(r)] = r
    return buf, lens


def as_int16_exact(a):
    """The read as an int16 array if every value is an integer that fits int16, else None.
    (One cast and one comparison: a wrapped, rounded or NaN value fails the comparison.)"""
    a = np.asarray(a)
    if a.dtype == np.int16:
        return a
    with np.errstate(invalid="ignore", over="ignore"):
        b = a.astype(np.int16)
    return b if (a.size == 0 or np.array_equal(b, a)) else None


def _split_int16(reads):
    """indices + int16 arrays of the integer-valued reads, indices of the rest"""
    ints, arrs, flts = [], [], []
    for i, r in enumerate(reads):
        b = as_int16_exact(r)
        if b is None:
            flts.append(i)
        else:
            ints.append(i)
            arrs.append(b)
    return ints, arrs, flts


def is_int16_exact(a):
    """True if every value of the float/int array is an integer that fits int16."""
    a = np.asarray(a)
    if a.size == 0:
        return True
    if a.dtype.kind in "iu":
        return bool(a.min() >= -32768 and a.max() <= 32767)
    return bool(np.all(np.isfinite(a)) and np.all(a == np.rint(a))
                and a.min() >= -32768 and a.max() <= 32767)


def _too_wide_for_i16(lo, hi):
    """The int16 kernels keep a histogram of the values between the outlier limits in LDS (up to
    ~38 900 values); wider limits go through the float64 kernels (radix select, same results)."""
    return min(int(hi), 32768) - max(int(lo), -32769) - 1 > 38000


# ----------------------------------------------------------------------------
# segmenter path
# ----------------------------------------------------------------------------
def segment_batch(sig, lens=None, params=None, max_segs=64, devices=None):
    """scale_outliers + get_segs for every row of an int16 [R, stride] batch.

    Returns (segs int32 [R, max_segs, 2], nsegs int32 [R]); grows max_segs and
    retries on overflow.  Coordinates are in the FILTERED signal, like the
    reference's (segmenter.py:209-211).  devices=[d0, d1, ...]: the reads are block-sharded over those
    GPUs, one host thread each (multigpu.py); results land in the same arrays, in input order."""
    devices = _devs(devices)
    L = _lib.load() if devices else _lib.ensure_init()
    sig = np.ascontiguousarray(sig, dtype=np.int16)
    if sig.ndim != 2:
        raise ValueError("sig must be [reads, samples]")
    R, stride = sig.shape
    lens = (np.full(R, stride, dtype=np.int32) if lens is None
            else np.ascontiguousarray(lens, dtype=np.int32))
    params = params or SegParams()
    if _too_wide_for_i16(params.lim_low, params.lim_hi):
        per_read = segment_reads_f64([sig[r, :lens[r]].astype(np.float64) for r in range(R)], params)
        nsegs = np.array([len(x) if x else 0 for x in per_read], dtype=np.int32)
        segs = np.zeros((R, max(max_segs, int(nsegs.max()) if R else 0), 2), dtype=np.int32)
        for r, x in enumerate(per_read):
            if x:
                segs[r, :len(x)] = x
        return segs, nsegs
    sharded = devices is not None and len(devices) > 1 and R >= len(devices)
    while True:
        segs = np.zeros((R, max_segs, 2), dtype=np.int32)
        nsegs = np.zeros(R, dtype=np.int32)
        if sharded:
            from . import multigpu
            rcs = []

            def shard(lo, hi, comm, ms=max_segs):
                if hi > lo:
                    rc = L.sk_segment_batch_i16(ptr(sig[lo:hi]), stride, ptr(lens[lo:hi]), hi - lo, C.byref(params),
                                                ptr(segs[lo:hi]), ptr(nsegs[lo:hi]), ms)
                    if rc != _lib.SK_ERR_OVERFLOW:
                        check(rc)
                    rcs.append(rc)
            multigpu.run_sharded(devices, R, shard)
            rc = _lib.SK_ERR_OVERFLOW if _lib.SK_ERR_OVERFLOW in rcs else 0
        else:
            if devices:
                _lib.init(devices[0])
            rc = L.sk_segment_batch_i16(ptr(sig), stride, ptr(lens), R, C.byref(params),
                                        ptr(segs), ptr(nsegs), max_segs)
        if rc == _lib.SK_ERR_OVERFLOW:
            max_segs = int(nsegs.max()) + 8
            continue
        check(rc)
        return segs, nsegs


def segment_reads(reads, params=None):
    """Fused scale_outliers + get_segs on a list of raw integer reads.
    Returns, per read, a list of [start, end] or False (the reference's value)."""
    if not len(reads):
        return []
    buf, lens = pack_i16(reads)
    segs, nsegs = segment_batch(buf, lens, params)
    return [segs[i, :nsegs[i]].tolist() if nsegs[i] else False for i in range(len(reads))]


def pack_f64(reads):
    """list of 1-D float arrays -> (flat float64, int64 offsets[R+1])."""
    off = np.zeros(len(reads) + 1, dtype=np.int64)
    for i, r in enumerate(reads):
        off[i + 1] = off[i] + len(r)
    flat = (np.concatenate([np.asarray(r, dtype=np.float64) for r in reads])
            if len(reads) else np.zeros(0))
    return np.ascontiguousarray(flat, dtype=np.float64), off


def segment_reads_f64(reads, params=None, max_segs=64):
    """Fused scale_outliers + get_segs on float64 (pA) reads (segmenter.py:198-199)."""
    if not len(reads):
        return []
    L = _lib.ensure_init()
    flat, off = pack_f64(reads)
    params = params or SegParams()
    R = len(reads)
    while True:
        segs = np.zeros((R, max_segs, 2), dtype=np.int32)
        nsegs = np.zeros(R, dtype=np.int32)
        rc = L.sk_segment_batch_f64(ptr(flat), ptr(off), R, C.byref(params), ptr(segs), ptr(nsegs), max_segs)
        if rc == _lib.SK_ERR_OVERFLOW:
            max_segs = int(nsegs.max()) + 8
            continue
        check(rc)
        return [segs[i, :nsegs[i]].tolist() if nsegs[i] else False for i in range(R)]


def _over_devices(devices, R, call):
    """call(lo, hi) -> status on the calling thread's GPU, or -- with several devices -- on one host thread per GPU
    over the block split of the R reads (multigpu.run_sharded; every shard writes its slice of the caller's arrays).
    Returns SK_ERR_OVERFLOW if any shard overflowed, else 0; any other failure raises."""
    devices = _devs(devices)
    if devices is not None and len(devices) > 1 and R >= len(devices):
        from . import multigpu
        rcs = []

        def shard(lo, hi, comm):
            if hi > lo:
                rc = call(lo, hi)
                if rc != _lib.SK_ERR_OVERFLOW:
                    check(rc)
                rcs.append(rc)
        multigpu.run_sharded(devices, R, shard)
        return _lib.SK_ERR_OVERFLOW if _lib.SK_ERR_OVERFLOW in rcs else 0
    if devices:
        _lib.init(devices[0])
    else:
        _lib.ensure_init()
    rc = call(0, R)
    if rc != _lib.SK_ERR_OVERFLOW:
        check(rc)
    return rc


def segment_batch_pa(sig, lens, calib, params=None, max_segs=64, devices=None):
    """Raw int16 rows through the pA route (segmenter.py:345-349: fast5 / slow5 input without --raw_signal): the
    conversion np.round((raw + offset) * (float("%.2f" % range) / digitisation), 2) is a monotone map of the sample, so
    since round 6 the rows stay int16 on the GPU and limits, median, std and thresholds are found in the raw domain
    (k_seg_stats<.., PA>; reads it cannot certify are redone from their float64 values in numpy's order).
    calib: float64 [R, 3] = digitisation, offset, range per read.  Returns (segs, nsegs).
    devices (or api.set_devices / --gpus): the reads are block-sharded like segment_batch's."""
    L = _lib.load()
    sig = np.ascontiguousarray(sig, dtype=np.int16)
    R = sig.shape[0]
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    calib = np.ascontiguousarray(calib, dtype=np.float64).reshape(R, 3)
    params = params or SegParams()
    while True:
        segs = np.zeros((max(R, 1), max_segs, 2), dtype=np.int32)
        nsegs = np.zeros(max(R, 1), dtype=np.int32)
        rc = _over_devices(devices, R, lambda lo, hi, ms=max_segs: L.sk_segment_batch_i16_pa(
            ptr(sig[lo:hi]), sig.shape[1], ptr(lens[lo:hi]), hi - lo, ptr(calib[lo:hi]), C.byref(params),
            ptr(segs[lo:hi]), ptr(nsegs[lo:hi]), ms))
        if rc == _lib.SK_ERR_OVERFLOW:
            max_segs = int(nsegs.max()) + 8
            continue
        return segs[:R], nsegs[:R]


def last_pa_retries():
    """Reads of the most recent segment_batch_pa call on this thread's device (its last sub-batch) that took the
    numpy-order redo; -1 when the call expanded its rows to float64 instead of staying in the raw domain."""
    return int(_lib.load().sk_last_pa_retries())


def segment_ragged_f64(values, off, lens=None, params=None, max_segs=64, devices=None):
    """scale_outliers + get_segs for a ragged float64 batch as a tokenizer leaves it: read r is the first lens[r]
    (default: all) of values[off[r]:off[r+1]].  Returns (segs int32 [R, max_segs, 2], nsegs int32 [R]).
    devices (or api.set_devices / --gpus): block-sharded over the GPUs (a shard is a run of offsets into the same
    `values`: nothing is repacked)."""
    L = _lib.load()
    # int32 values are centi-units (tsvio.FloatBlock.centi: tokens with at most two decimals): sample = c / 100.0 on the GPU
    centi = isinstance(values, np.ndarray) and values.dtype == np.int32
    values = np.ascontiguousarray(values, dtype=np.int32 if centi else np.float64)
    entry = L.sk_segment_batch_centi_len if centi else L.sk_segment_batch_f64_len
    off = np.ascontiguousarray(off, dtype=np.int64)
    R = off.size - 1
    params = params or SegParams()
    ln = None if lens is None else np.ascontiguousarray(lens, dtype=np.int32)
    while True:
        segs = np.zeros((max(R, 1), max_segs, 2), dtype=np.int32)
        nsegs = np.zeros(max(R, 1), dtype=np.int32)
        rc = _over_devices(devices, R, lambda lo, hi, ms=max_segs: entry(
            ptr(values), ptr(off[lo:hi + 1]), None if ln is None else ptr(ln[lo:hi]), hi - lo, C.byref(params),
            ptr(segs[lo:hi]), ptr(nsegs[lo:hi]), ms))
        if rc == _lib.SK_ERR_OVERFLOW:
            max_segs = int(nsegs.max()) + 8
            continue
        return segs[:R], nsegs[:R]


def motifseq_multi_ragged_f64(values, off, motifs, scale="medmad", scale_low=0, scale_hi=1200, devices=None):
    """Every motif against a ragged float64 batch (read r = values[off[r]:off[r+1]]): one record array per motif.
    devices (or api.set_devices / --gpus): block-sharded over the GPUs."""
    L = _lib.load()
    centi = isinstance(values, np.ndarray) and values.dtype == np.int32      # centi-units, see segment_ragged_f64
    values = np.ascontiguousarray(values, dtype=np.int32 if centi else np.float64)
    entry = L.sk_motifseq_multi_batch_centi if centi else L.sk_motifseq_multi_batch_f64
    off = np.ascontiguousarray(off, dtype=np.int64)
    R = off.size - 1
    ms = [np.ascontiguousarray(m, dtype=np.float64) for m in motifs]
    out = [np.zeros(max(R, 1), dtype=HIT_DTYPE) for _ in ms]
    flat = np.ascontiguousarray(np.concatenate(ms)) if ms else np.zeros(0)
    moff = np.concatenate([[0], np.cumsum([m.size for m in ms])]).astype(np.int32)

    def call(lo, hi):
        # the shard is staged and filtered ONCE, every motif runs against it on the device (sk_motifseq_multi_batch_f64)
        part = np.zeros((len(ms), hi - lo), dtype=HIT_DTYPE)
        rc = entry(ptr(values), ptr(off[lo:hi + 1]), hi - lo, ptr(flat), ptr(moff), len(ms),
                   _lib.SK_SCALE[scale], int(scale_low), int(scale_hi), ptr(part))
        if rc == 0:
            for k, hits in enumerate(out):
                hits[lo:hi] = part[k]
        return rc
    if R and ms:
        _over_devices(devices, R, call)
    return [h[:R] for h in out]


def drna_segment_batch(sig, lens, params=None, max_segs=32):
    """dRNA_segmenter.py's slow5 branch for every row of an int16 [R, stride] batch: (segs int32 [R, max_segs, 2], nsegs)."""
    L = _lib.ensure_init()
    sig = np.ascontiguousarray(sig, dtype=np.int16)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    params = params or _lib.DrnaParams()
    R = sig.shape[0]
    while True:
        segs = np.zeros((max(R, 1), max_segs, 2), dtype=np.int32)
        nsegs = np.zeros(max(R, 1), dtype=np.int32)
        rc = L.sk_drna_segment_batch_i16(ptr(sig), sig.shape[1], ptr(lens), R, C.byref(params), ptr(segs), ptr(nsegs), max_segs)
        if rc == _lib.SK_ERR_OVERFLOW:
            max_segs = int(nsegs.max()) + 8
            continue
        check(rc)
        return segs[:R], nsegs[:R]


def drna_segment_reads(reads, params=None, max_segs=32):
    """dRNA_segmenter.py's slow5-branch per-read work (scale_outliers, window statistics, scan)
    for a list of raw integer reads: per read the list of [start, end] collected before the scan
    stopped (the script prints only the first, dRNA_segmenter.py:173-176)."""
    if not len(reads):
        return []
    L = _lib.ensure_init()
    buf, lens = pack_i16(reads)
    params = params or _lib.DrnaParams()
    R = len(reads)
    while True:
        segs = np.zeros((R, max_segs, 2), dtype=np.int32)
        nsegs = np.zeros(R, dtype=np.int32)
        rc = L.sk_drna_segment_batch_i16(ptr(buf), buf.shape[1], ptr(lens), R, C.byref(params),
                                         ptr(segs), ptr(nsegs), max_segs)
        if rc == _lib.SK_ERR_OVERFLOW:
            max_segs = int(nsegs.max()) + 8
            continue
        check(rc)
        return [segs[i, :nsegs[i]].tolist() for i in range(R)]


def segment_any(reads, params=None):
    """Route each read to the int16 kernels when it is integer valued and fits,
    else to the float64 kernels; results come back in input order."""
    ints, arrs, flts = _split_int16(reads)
    out = [None] * len(reads)
    if ints:
        for i, res in zip(ints, segment_reads(arrs, params)):
            out[i] = res
    if flts:
        for i, res in zip(flts, segment_reads_f64([reads[i] for i in flts], params)):
            out[i] = res
    return out


def get_segs(sig, args):
    """Drop-in for segmenter.get_segs(sig, args): `sig` is already filtered
    (segmenter.py:209), args carries error/corrector/window/seg_dist/std_scale/
    stall_len.  Returns [[start, end], ...] or False."""
    sig = np.asarray(sig)
    if sig.size == 0:
        return False
    # limits that keep every sample: the caller has already run scale_outliers
    lo = int(np.floor(float(sig.min()))) - 1
    hi = int(np.ceil(float(sig.max()))) + 1
    p = SegParams(args.error, args.corrector, args.window, args.seg_dist, args.std_scale,
                  args.stall_len, lo, hi)
    return segment_any([sig], p)[0]


def test_segs(segs, args, err=None):
    """segmenter.test_segs (segmenter.py:473-494): host-side acceptance filter.
    Messages go to `err` (a file object) exactly as the reference writes them."""
    import sys
    import traceback
    err = err or sys.stderr
    try:
        if args.stall:
            if segs[0][0] > args.stall_start:
                err.write("start seg too late!")
                return False
        if args.gap:
            if segs[1][0] > segs[0][1] + args.gap_dist:
                err.write("second seg too far!")
                return False
    except Exception:                                   # reference: bare except, read passes
        err.write("something went wrong test_segs()")
        traceback.print_exc(file=err)
    return segs


def drna_roll_reads(reads, params=None):
    """dRNA_segmenter.py's --signal branch (:272-326) on raw integer reads: per read (x, y) -- the first
    low rolling-mean segment of acceptable length, both ends shifted as the script prints them -- or None."""
    if not len(reads):
        return []
    from ._lib import RollParams
    L = _lib.ensure_init()
    params = params or RollParams()
    buf, lens = pack_i16(reads)
    R = len(reads)
    xy = np.zeros((R, 2), dtype=np.int32)
    found = np.zeros(R, dtype=np.int32)
    check(L.sk_drna_roll_batch_i16(ptr(buf), buf.shape[1], ptr(lens), R, C.byref(params), ptr(xy), ptr(found)))
    return [(int(xy[i, 0]), int(xy[i, 1])) if found[i] else None for i in range(R)]


# ----------------------------------------------------------------------------
# MotifSeq path
# ----------------------------------------------------------------------------
def motifseq_batch(sig, lens, motif, scale="medmad", scale_low=0, scale_hi=1200, devices=None, gather="host"):
    """scale_outliers + medmad/zscale + dtw_subsequence for every row of an
    int16 [R, stride] batch.  Returns a HIT_DTYPE record array (dist, start,
    end, n, flags); start/end index the FILTERED signal (MotifSeq.py:438-439).
    devices=[d0, d1, ...]: reads block-sharded over those GPUs, one host thread each; gather="host" writes
    every shard's records straight into the result, gather="rccl" all-gathers them GPU to GPU first (RCCL) and
    downloads the complete result from the first device (multigpu.motifseq_sharded)."""
    devices = _devs(devices)
    L = _lib.load() if devices else _lib.ensure_init()
    sig = np.ascontiguousarray(sig, dtype=np.int16)
    R, stride = sig.shape
    lens = (np.full(R, stride, dtype=np.int32) if lens is None
            else np.ascontiguousarray(lens, dtype=np.int32))
    motif = np.ascontiguousarray(motif, dtype=np.float64)
    if _too_wide_for_i16(scale_low, scale_hi):
        return motifseq_reads_f64([sig[r, :lens[r]].astype(np.float64) for r in range(R)], motif, scale,
                                  scale_low, scale_hi)
    out = np.zeros(R, dtype=HIT_DTYPE)
    if devices is not None and len(devices) > 1 and R >= len(devices):
        from . import multigpu
        if gather == "rccl":
            return multigpu.motifseq_sharded(sig, lens, motif, _lib.SK_SCALE[scale], scale_low, scale_hi,
                                             devices, gather="rccl")[0]

        def shard(lo, hi, comm):
            if hi > lo:
                check(L.sk_motifseq_batch_i16(ptr(sig[lo:hi]), stride, ptr(lens[lo:hi]), hi - lo, ptr(motif),
                                              motif.size, _lib.SK_SCALE[scale], int(scale_low), int(scale_hi),
                                              ptr(out[lo:hi])))
        multigpu.run_sharded(devices, R, shard)
        return out
    if devices:
        _lib.init(devices[0])
    check(L.sk_motifseq_batch_i16(ptr(sig), stride, ptr(lens), R, ptr(motif), motif.size,
                                  _lib.SK_SCALE[scale], int(scale_low), int(scale_hi), ptr(out)))
    return out


GUARD_FIELDS = ("premise_violations", "audited", "audit_mismatches", "image_rejects", "alarm", "exact_fallback",
                "second_windows")


def last_dtw_guard():
    """Run-time guard counters of the most recent DTW call on this thread's device (sk_last_dtw_guard): premise
    violations the window pass found, reads audited by the exact pass and how many of them differed, reads kept from
    the screening because their sample image could not be bounded, the alarm count, and whether the whole call was
    redone by the exact pass.  A healthy build reports 0 violations / 0 mismatches, always."""
    g = (C.c_int32 * 8)()
    check(_lib.load().sk_last_dtw_guard(g))
    return dict(zip(GUARD_FIELDS, (int(v) for v in g)))


def motifseq_reads_f64(reads, motif, scale="medmad", scale_low=0, scale_hi=1200):
    """Same as motifseq_batch for float64 (pA) reads given as a list (MotifSeq.py:270)."""
    L = _lib.ensure_init()
    flat, off = pack_f64(reads)
    motif = np.ascontiguousarray(motif, dtype=np.float64)
    out = np.zeros(len(reads), dtype=HIT_DTYPE)
    check(L.sk_motifseq_batch_f64(ptr(flat), ptr(off), len(reads), ptr(motif), motif.size,
                                  _lib.SK_SCALE[scale], int(scale_low), int(scale_hi), ptr(out)))
    return out


def motifseq_any(reads, motif, scale="medmad", scale_low=0, scale_hi=1200):
    """Integer-valued reads go through the int16 kernels, the rest through the
    float64 kernels (bit-identical results either way); input order is kept."""
    out = np.zeros(len(reads), dtype=HIT_DTYPE)
    ints, arrs, flts = _split_int16(reads)
    if ints:
        buf, lens = pack_i16(arrs)
        out[ints] = motifseq_batch(buf, lens, motif, scale, scale_low, scale_hi)
    if flts:
        out[flts] = motifseq_reads_f64([reads[i] for i in flts], motif, scale, scale_low, scale_hi)
    return out


def stall_cuts(reads, seg_params=None):
    """[extension -- the author's TODO "integration with MotifSeq", segmenter.py:35]  Per read, the raw
    index right after the first segment the segmenter finds (the stall at the start of a read), 0 when it
    finds none: the filtered coordinate get_segs reports is mapped back through scale_outliers' mask."""
    params = seg_params or SegParams()
    cuts = np.zeros(len(reads), dtype=np.int64)
    for i, (r, s) in enumerate(zip(reads, segment_any(reads, params))):
        if s:
            a = np.asarray(r)
            kept = np.flatnonzero((a > params.lim_low) & (a < params.lim_hi))      # segmenter.py:311-318
            e = s[0][1]
            cuts[i] = int(kept[e]) if e < kept.size else int(a.size)
    return cuts


def motifseq_after_stall(reads, motif, scale="medmad", scale_low=0, scale_hi=1200, seg_params=None):
    """[extension]  MotifSeq on what follows the stall: reads[i][cut:] goes through the usual filter,
    normalisation and subsequence DTW.  Returns (hits, cuts); hit coordinates index the filtered slice."""
    cuts = stall_cuts(reads, seg_params)
    return motifseq_any([np.asarray(r)[c:] for r, c in zip(reads, cuts)], motif, scale, scale_low, scale_hi), cuts


def motifseq_multi_batch(sig, lens, motifs, scale="medmad", scale_low=0, scale_hi=1200):
    """Every motif of `motifs` against every row of an int16 [R, stride] batch (one filter / statistics pass):
    list (one per motif) of HIT_DTYPE arrays in read order.  The block form of motifseq_multi."""
    sig = np.ascontiguousarray(sig, dtype=np.int16)
    R = sig.shape[0]
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    if _too_wide_for_i16(scale_low, scale_hi):
        return motifseq_multi([sig[r, :lens[r]] for r in range(R)], motifs, scale, scale_low, scale_hi)
    return motifseq_multi([], motifs, scale, scale_low, scale_hi, _packed=(sig, lens))


def motifseq_multi(reads, motifs, scale="medmad", scale_low=0, scale_hi=1200, _packed=None):
    """Every motif of `motifs` (list of float vectors) against every read: list (one per motif,
    in order) of HIT_DTYPE arrays in read order -- the double loop of MotifSeq.py:261-298,436.
    Integer reads share one filter/statistics pass across motifs."""
    L = _lib.ensure_init()
    motifs = [np.ascontiguousarray(m, dtype=np.float64) for m in motifs]
    if _packed is not None:                                   # an int16 batch as it is (motifseq_multi_batch)
        buf, lens = _packed
        ints, flts = list(range(buf.shape[0])), []
        outs = [np.zeros(len(ints), dtype=HIT_DTYPE) for _ in motifs]
    else:
        outs = [np.zeros(len(reads), dtype=HIT_DTYPE) for _ in motifs]
        ints, arrs, flts = _split_int16(reads)
        if ints and motifs:
            buf, lens = pack_i16(arrs)
    if ints and motifs:
        moff = np.zeros(len(motifs) + 1, dtype=np.int32)
        moff[1:] = np.cumsum([m.size for m in motifs])
        flat = np.ascontiguousarray(np.concatenate(motifs))
        res = np.zeros((len(motifs), len(ints)), dtype=HIT_DTYPE)
        devices = _devs(None)
        if devices is not None and len(devices) > 1 and len(ints) >= len(devices):
            from . import multigpu

            def shard(lo, hi, comm):
                if hi > lo:
                    part = np.zeros((len(motifs), hi - lo), dtype=HIT_DTYPE)
                    check(L.sk_motifseq_multi_batch_i16(ptr(buf[lo:hi]), buf.shape[1], ptr(lens[lo:hi]), hi - lo,
                                                        ptr(flat), ptr(moff), len(motifs), _lib.SK_SCALE[scale],
                                                        int(scale_low), int(scale_hi), ptr(part)))
                    res[:, lo:hi] = part
            multigpu.run_sharded(devices, len(ints), shard)
        else:
            check(L.sk_motifseq_multi_batch_i16(ptr(buf), buf.shape[1], ptr(lens), len(ints), ptr(flat), ptr(moff),
                                                len(motifs), _lib.SK_SCALE[scale], int(scale_low), int(scale_hi),
                                                ptr(res)))
        for k in range(len(motifs)):
            outs[k][ints] = res[k]
    if flts:
        sub = [reads[i] for i in flts]
        for k, m in enumerate(motifs):
            outs[k][flts] = motifseq_reads_f64(sub, m, scale, scale_low, scale_hi)
    return outs


def normalise(sig, scale="medmad", scale_low=0, scale_hi=1200):
    """Filtered + normalised signal of one read, as MotifSeq hands it to
    dtw_subsequence (MotifSeq.py:274-289)."""
    L = _lib.ensure_init()
    n = C.c_int32(0)
    if is_int16_exact(sig):
        s = np.ascontiguousarray(np.asarray(sig).astype(np.int16))
        out = np.empty(max(1, s.size), dtype=np.float64)
        check(L.sk_normalise_i16(ptr(s), s.size, _lib.SK_SCALE[scale], int(scale_low), int(scale_hi),
                                 ptr(out), C.byref(n)))
    else:
        s = np.ascontiguousarray(sig, dtype=np.float64)
        out = np.empty(max(1, s.size), dtype=np.float64)
        check(L.sk_normalise_f64(ptr(s), s.size, _lib.SK_SCALE[scale], int(scale_low), int(scale_hi),
                                 ptr(out), C.byref(n)))
    return out[:n.value].copy()


class LastRowCost:
    """What MotifSeq reads from mlpy's cost matrix: only `cost[-1, :]`
    (MotifSeq.py:507-509).  The N x n matrix itself is never materialised."""

    def __init__(self, last_row, nrows):
        self._last = last_row
        self.shape = (nrows, last_row.size)

    def __getitem__(self, key):
        if isinstance(key, tuple):
            row = key[0]
            rest = key[1] if len(key) > 1 else slice(None)
        else:
            row, rest = key, slice(None)
        if row in (-1, self.shape[0] - 1):
            return self._last[rest]
        raise IndexError("only the last row of the DTW cost matrix is kept on this path")


def dtw_subsequence(x, y, last_row=False):
    """Drop-in for mlpy.dtw_subsequence(x, y) as MotifSeq.py:437-439 consumes it:
    returns (dist, cost, path) with path[1][0] == start and path[1][-1] == end.
    `cost` supports cost[-1, :] when last_row=True, else it is None."""
    L = _lib.ensure_init()
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    dist, s, e = C.c_double(), C.c_int32(), C.c_int32()
    row = np.empty(y.size, dtype=np.float64) if last_row else None
    check(L.sk_dtw_subsequence(ptr(x), x.size, ptr(y), y.size, C.byref(dist), C.byref(s),
                               C.byref(e), ptr(row) if last_row else None))
    path = (np.array([0, x.size - 1]), np.array([s.value, e.value]))
    return dist.value, (LastRowCost(row, x.size) if last_row else None), path


def dtw_subsequence_cref(x, y):
    """mlpy.dtw_subsequence(x, y) -> (dist, start, end) in the reference's own C arithmetic for inputs that hold inf / nan
    (medmad of a read whose MAD is 0): one GPU lane, full cost matrix (sk_dtw_subsequence_cref)."""
    L = _lib.ensure_init()
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    dist, s, e = C.c_double(), C.c_int32(), C.c_int32()
    check(L.sk_dtw_subsequence_cref(ptr(x), x.size, ptr(y), y.size, C.byref(dist), C.byref(s), C.byref(e)))
    return dist.value, s.value, e.value


def dtw_subsequence_batch(x, ys):
    """dtw_subsequence(x, y) for a list of already-normalised float64 signals."""
    L = _lib.ensure_init()
    x = np.ascontiguousarray(x, dtype=np.float64)
    off = np.zeros(len(ys) + 1, dtype=np.int64)
    for i, y in enumerate(ys):
        off[i + 1] = off[i] + len(y)
    flat = (np.concatenate([np.asarray(y, dtype=np.float64) for y in ys])
            if len(ys) else np.zeros(0))
    flat = np.ascontiguousarray(flat, dtype=np.float64)
    out = np.zeros(len(ys), dtype=HIT_DTYPE)
    check(L.sk_dtw_subsequence_batch(ptr(x), x.size, ptr(flat), ptr(off), len(ys), ptr(out)))
    return out
