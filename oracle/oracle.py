"""ctypes view of oracle/libsk_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (as the checker / the timed CPU baseline).  The product package
`squigglekit_amd` never does.

Pinning: segmenter + normalisation pinned by goldens minted from the reference
(tools/gen_golden.py); the DTW core restates third-party mlpy 3.5.0
(mlpy/dtw/cdtw.c, absent from /root/reference) -- "parity unpinned".
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsk_oracle.so")


def build(force=False):
    """Compile the C restatement with gcc (no GPU needed)."""
    src = os.path.join(_HERE, "sk_oracle.c")
    if (force or not os.path.exists(_SO)
            or os.path.getmtime(_SO) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class SegParams(C.Structure):
    """Mirror of ora_seg_params; defaults are segmenter.py:67-88."""
    _fields_ = [("error", C.c_int32), ("corrector", C.c_int32), ("window", C.c_int32),
                ("seg_dist", C.c_int32), ("std_scale", C.c_double), ("stall_len", C.c_double)]

    def __init__(self, error=5, corrector=50, window=150, seg_dist=50,
                 std_scale=0.75, stall_len=0.25):
        super().__init__(error, corrector, window, seg_dist, std_scale, stall_len)


class RollParams(C.Structure):
    """Mirror of ora_roll_params: the constants of dRNA_segmenter.py:291-295,322 and the window `w`
    the script forgot to define (its commented default, :81, is 2000)."""
    _fields_ = [("w", C.c_int32), ("seg_dist", C.c_int32), ("hi_thresh", C.c_int32), ("lo_thresh", C.c_int32),
                ("shift", C.c_int32), ("std_scale", C.c_double)]

    def __init__(self, w=2000, seg_dist=1500, hi_thresh=200000, lo_thresh=2000, shift=1000, std_scale=0.5):
        super().__init__(w, seg_dist, hi_thresh, lo_thresh, shift, std_scale)


class DrnaParams(C.Structure):
    """Mirror of ora_drna_params; defaults are dRNA_segmenter.py:80-104's constants."""
    _fields_ = [("error", C.c_int32), ("no_err_thresh", C.c_int32), ("w", C.c_int32),
                ("window", C.c_int32), ("seg_dist", C.c_int32), ("t_start", C.c_int32),
                ("t_end", C.c_int32), ("std_scale", C.c_double)]

    def __init__(self, error=5, no_err_thresh=2500, w=1200, window=100, seg_dist=1200,
                 t_start=1000, t_end=5000, std_scale=0.8):
        super().__init__(error, no_err_thresh, w, window, seg_dist, t_start, t_end, std_scale)


class Hit(C.Structure):
    _fields_ = [("dist", C.c_double), ("start", C.c_int32), ("end", C.c_int32),
                ("n", C.c_int32), ("flags", C.c_int32)]


HIT_DTYPE = np.dtype([("dist", "<f8"), ("start", "<i4"), ("end", "<i4"),
                      ("n", "<i4"), ("flags", "<i4")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp, ip, i16p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int16)
        L.ora_pairwise_sum.restype = C.c_double
        L.ora_pairwise_sum.argtypes = [dp, C.c_int64]
        L.ora_np_sum.restype = C.c_double
        L.ora_np_sum.argtypes = [dp, C.c_int64]
        for f in (L.ora_mean, L.ora_std, L.ora_median):
            f.restype = C.c_double
            f.argtypes = [dp, C.c_int64]
        L.ora_scale_outliers.restype = C.c_int64
        L.ora_scale_outliers.argtypes = [dp, C.c_int64, C.c_double, C.c_double, dp]
        L.ora_get_segs.restype = C.c_int32
        L.ora_get_segs.argtypes = [dp, C.c_int64, C.POINTER(SegParams), ip, C.c_int32, dp, dp]
        L.ora_drna_roll.restype = C.c_int
        L.ora_drna_roll.argtypes = [dp, C.c_int64, C.POINTER(RollParams), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int64), dp, dp]
        L.ora_drna_segs.restype = C.c_int32
        L.ora_drna_segs.argtypes = [dp, C.c_int64, C.POINTER(DrnaParams), ip, C.c_int32, dp]
        L.ora_medmad.restype = None
        L.ora_medmad.argtypes = [dp, C.c_int64, dp, dp, dp]
        L.ora_zscale.restype = C.c_int
        L.ora_zscale.argtypes = [dp, C.c_int64, dp, dp, dp]
        L.ora_dtw_subsequence.restype = C.c_int
        L.ora_dtw_subsequence.argtypes = [dp, C.c_int32, dp, C.c_int32, dp, ip, ip, dp, dp]
        L.ora_dtw_subsequence_fwd.restype = C.c_int
        L.ora_dtw_subsequence_fwd.argtypes = [dp, C.c_int32, dp, C.c_int32, dp, ip, ip]
        L.ora_dtw_subsequence_path.restype = C.c_int32
        L.ora_dtw_subsequence_path.argtypes = [dp, C.c_int32, dp, C.c_int32, ip, ip, C.c_int32]
        L.ora_motifseq_batch_i16.restype = C.c_int
        L.ora_motifseq_batch_i16.argtypes = [i16p, C.c_int64, ip, C.c_int32, dp, C.c_int32,
                                             C.c_int, C.c_int32, C.c_int32, C.POINTER(Hit)]
        L.ora_segment_batch_i16.restype = C.c_int
        L.ora_segment_batch_i16.argtypes = [i16p, C.c_int64, ip, C.c_int32, C.POINTER(SegParams),
                                            C.c_int32, C.c_int32, ip, ip, C.c_int32]
        _lib = L
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def pairwise_sum(x):
    a, p = _d(x)
    return lib().ora_pairwise_sum(p, a.size)


def np_sum(x):
    a, p = _d(x)
    return lib().ora_np_sum(p, a.size)


def mean(x):
    a, p = _d(x)
    return lib().ora_mean(p, a.size)


def std(x):
    a, p = _d(x)
    return lib().ora_std(p, a.size)


def median(x):
    a, p = _d(x)
    return lib().ora_median(p, a.size)


def scale_outliers(x, lo, hi):
    a, p = _d(x)
    out = np.empty_like(a)
    k = lib().ora_scale_outliers(p, a.size, float(lo), float(hi),
                                 out.ctypes.data_as(C.POINTER(C.c_double)))
    return out[:k]


def get_segs(sig, params=None, max_segs=4096, return_thresholds=False):
    """Restated segmenter.get_segs: list of [start, end] or False."""
    a, p = _d(sig)
    params = params or SegParams()
    segs = np.zeros(2 * max_segs, dtype=np.int32)
    top, bot = C.c_double(), C.c_double()
    k = lib().ora_get_segs(p, a.size, C.byref(params), _i(segs), max_segs,
                           C.byref(top), C.byref(bot))
    if k < 0:
        raise ValueError("invalid segmenter parameters")
    if k > max_segs:
        raise OverflowError("max_segs too small")
    out = segs[:2 * k].reshape(-1, 2).tolist() if k else False
    if return_thresholds:
        return out, top.value, bot.value
    return out


def drna_segs(sig, params=None, max_segs=256):
    """dRNA_segmenter's slow5-branch scan on an already filtered signal: list of [start, end]."""
    a, p = _d(sig)
    params = params or DrnaParams()
    segs = np.zeros(2 * max_segs, dtype=np.int32)
    top = C.c_double()
    k = lib().ora_drna_segs(p, a.size, C.byref(params), _i(segs), max_segs, C.byref(top))
    if k < 0:
        raise ValueError("invalid dRNA parameters")
    return segs[:2 * min(k, max_segs)].reshape(-1, 2).tolist(), top.value


def drna_roll(sig, params=None, want_t=False):
    """dRNA_segmenter's --signal branch on an already filtered signal: (x, y) or None; with
    want_t also the rolling mean and (mn, std, bot)."""
    a, p = _d(sig)
    params = params or RollParams()
    x, y = C.c_int64(), C.c_int64()
    t = np.empty(max(1, a.size), dtype=np.float64)
    stats = np.zeros(3, dtype=np.float64)
    found = lib().ora_drna_roll(p, a.size, C.byref(params), C.byref(x), C.byref(y),
                                t.ctypes.data_as(C.POINTER(C.c_double)), stats.ctypes.data_as(C.POINTER(C.c_double)))
    if found < 0:
        raise ValueError("invalid rolling parameters")
    res = (x.value, y.value) if found else None
    return (res, t[:a.size], tuple(stats)) if want_t else res


def medmad(x):
    a, p = _d(x)
    out = np.empty_like(a)
    med, smad = C.c_double(), C.c_double()
    lib().ora_medmad(p, a.size, out.ctypes.data_as(C.POINTER(C.c_double)),
                     C.byref(med), C.byref(smad))
    return out, med.value, smad.value


def zscale(x):
    a, p = _d(x)
    out = np.empty_like(a)
    mean_, sc = C.c_double(), C.c_double()
    fired = lib().ora_zscale(p, a.size, out.ctypes.data_as(C.POINTER(C.c_double)),
                             C.byref(mean_), C.byref(sc))
    return out, mean_.value, sc.value, fired


def dtw_subsequence(x, y, want_cost=False):
    """mlpy.dtw_subsequence restated: returns (dist, start, end[, cost])."""
    xa, xp = _d(x)
    ya, yp = _d(y)
    dist, s, e = C.c_double(), C.c_int32(), C.c_int32()
    cost = np.empty((xa.size, ya.size)) if want_cost else None
    cp = cost.ctypes.data_as(C.POINTER(C.c_double)) if want_cost else None
    rc = lib().ora_dtw_subsequence(xp, xa.size, yp, ya.size, C.byref(dist), C.byref(s),
                                   C.byref(e), cp, None)
    if rc:
        raise ValueError("empty input")
    if want_cost:
        return dist.value, s.value, e.value, cost
    return dist.value, s.value, e.value


def dtw_subsequence_fwd(x, y):
    xa, xp = _d(x)
    ya, yp = _d(y)
    dist, s, e = C.c_double(), C.c_int32(), C.c_int32()
    rc = lib().ora_dtw_subsequence_fwd(xp, xa.size, yp, ya.size, C.byref(dist),
                                       C.byref(s), C.byref(e))
    if rc:
        raise ValueError("empty input")
    return dist.value, s.value, e.value


def dtw_subsequence_path(x, y):
    xa, xp = _d(x)
    ya, yp = _d(y)
    cap = xa.size + ya.size
    px = np.zeros(cap, dtype=np.int32)
    py = np.zeros(cap, dtype=np.int32)
    k = lib().ora_dtw_subsequence_path(xp, xa.size, yp, ya.size, _i(px), _i(py), cap)
    return px[:k].copy(), py[:k].copy()


def motifseq_batch_i16(sig, lens, motif, scale_mode=0, lo=0, hi=1200):
    """filter -> normalise -> dtw for every row of an int16 [R, M] batch."""
    sig = np.ascontiguousarray(sig, dtype=np.int16)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    ma, mp = _d(motif)
    out = np.zeros(sig.shape[0], dtype=HIT_DTYPE)
    rc = lib().ora_motifseq_batch_i16(sig.ctypes.data_as(C.POINTER(C.c_int16)), sig.shape[1],
                                      _i(lens), sig.shape[0], mp, ma.size, scale_mode,
                                      lo, hi, out.ctypes.data_as(C.POINTER(Hit)))
    if rc:
        raise MemoryError
    return out


def segment_batch_i16(sig, lens, params=None, lo=0, hi=900, max_segs=64):
    sig = np.ascontiguousarray(sig, dtype=np.int16)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    params = params or SegParams()
    R = sig.shape[0]
    segs = np.zeros((R, max_segs, 2), dtype=np.int32)
    nsegs = np.zeros(R, dtype=np.int32)
    rc = lib().ora_segment_batch_i16(sig.ctypes.data_as(C.POINTER(C.c_int16)), sig.shape[1],
                                     _i(lens), R, C.byref(params), lo, hi,
                                     _i(segs), _i(nsegs), max_segs)
    if rc:
        raise ValueError("oracle segment batch failed")
    return segs, nsegs


# ------------------------------------------------------------------------------------------------
# Python-speed restatements: what the reference costs "as shipped" (BASELINE.md section 3, items 2 and 4a).
# Test infrastructure like everything else here -- bench.py times them, tests pin them to the C oracle.
# ------------------------------------------------------------------------------------------------
def medmad_python_loop(sig):
    """MotifSeq.py:192-200 at the reference's own speed: numpy medians, then one Python-level
    (x - med) / scaled_mad and one list append per sample, then a list -> ndarray conversion."""
    sig = np.asarray(sig)
    centre = np.median(sig)
    spread = np.median(np.abs(sig - centre)) * 1.4826
    acc = []
    for x in sig:
        acc.append((x - centre) / spread)
    return np.array(acc)


def get_segs_python(sig, params=None):
    """segmenter.py:399-470 at interpreter speed (one Python iteration per sample).  Same decisions as
    ora_get_segs; written as an explicit in/tolerated/closing classification per sample."""
    p = params or SegParams()
    sig = np.asarray(sig)
    mid = np.median(sig)
    sd = np.std(sig)
    hi_t = mid + sd * p.std_scale                                    # :413
    lo_t = mid - sd * p.std_scale                                    # :414
    inside = False
    run = errs = tail = 0
    period = p.corrector                                             # :424, never reset
    first = 0
    found = []
    for i in range(len(sig)):
        v = sig[i]
        in_band = lo_t < v < hi_t                                    # :431
        if in_band or (inside and errs < p.error):                   # :431 / :442 -- the sample extends the run
            if in_band:
                if not inside:
                    inside, first = True, i
                period += 1
                tail = 0
            else:
                errs += 1
                tail += 1
            run += 1
            if run >= p.window and run >= period and run % period == 0:   # :439 / :446
                errs -= 1
            continue
        if not inside:
            continue
        long_enough = run >= p.window or (not found and run >= p.window * p.stall_len)   # :448
        if long_enough:
            stop = i - tail                                          # :449
            if found and first - found[-1][1] < p.seg_dist:          # :451
                found[-1][1] = stop
            else:
                found.append([first, stop])
        inside = False
        run = errs = tail = 0
    return found if found else False
