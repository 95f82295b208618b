#!/usr/bin/env python3
"""The raw-domain pA segmenter alone on long reads (what tools/profile and rocprofv3 --pmc runs want: few kernels):
    python tools/bench_pa_long.py [reads=50000] [samples=20000] [steps=5] [route=pa|i16|f64]
route pa : sk_segment_dev_i16_pa (k_seg_stats<.., PA>, k_seg_walkL)      -- segmenter.py:345-349 input
      i16: sk_segment_dev_i16 on the same rows (--raw_signal)
      f64: the float64 image of the rows through sk_segment_dev_f64      -- what round 5 did for pa"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SK_TUNING", "1")
from squigglekit_amd import _lib, synth            # noqa: E402
from squigglekit_amd._lib import SegParams, check, ptr   # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    route = sys.argv[4] if len(sys.argv) > 4 else "pa"
    _lib.init(0)
    L = _lib.load()
    S = (M + 7) // 8 * 8
    d_raw = L.sk_dev_alloc(R * S * 2)
    check(L.sk_synth_squiggles_dev(d_raw, S, R, M, synth.SEED_C5, None, 0))
    lens = np.full(R, M - 1, dtype=np.int32)
    d_len = L.sk_dev_alloc(R * 4)
    check(L.sk_dev_upload(d_len, ptr(lens), lens.nbytes))
    cal3 = np.tile(np.array([8192.0, 16.0, 1493.94]), (R, 1))
    cal2 = np.empty((R, 2))
    check(L.sk_pa_calib(ptr(cal3), R, ptr(cal2)))
    d_cal = L.sk_dev_alloc(R * 16)
    check(L.sk_dev_upload(d_cal, ptr(cal2), cal2.nbytes))
    MS = 128
    d_segs, d_n = L.sk_dev_alloc(R * MS * 8), L.sk_dev_alloc(R * 4)
    sp = SegParams()
    if route == "f64":
        d_pa, d_off = L.sk_dev_alloc(R * (M - 1) * 8), L.sk_dev_alloc((R + 1) * 8)
        check(L.sk_synth_pa_dev(d_raw, S, R, M - 1, 16.0, 1493.94, 8192.0, d_pa, d_off))

    def call():
        if route == "pa":
            check(L.sk_segment_dev_i16_pa(d_raw, S, d_len, R, d_cal, C.byref(sp), d_segs, d_n, MS))
        elif route == "i16":
            check(L.sk_segment_dev_i16(d_raw, S, d_len, R, C.byref(sp), d_segs, d_n, MS))
        else:
            check(L.sk_segment_dev_f64(d_pa, d_off, R, R * (M - 1), M - 1, C.byref(sp), d_segs, d_n, MS))
        check(L.sk_sync())
    call()
    ts, ev = [], None
    for _ in range(steps):
        t0 = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t0)
        p_, m_ = C.c_float(), C.c_float()
        check(L.sk_last_kernel_ms(C.byref(p_), C.byref(m_)))
        ev = (p_.value, m_.value)
    t = min(ts)
    bps = 8 if route == "f64" else 2
    alg = R * (bps * (M - 1) + 36)
    print("%s: %d reads x %d samples: %.3f ms/step (statistics %.3f, walk %.3f ms); %.0f GB/s on %d B/sample = %.3f of 8 TB/s; "
          "statistics kernel alone %.3f" % (route, R, M - 1, t * 1e3, ev[0], ev[1], alg / t / 1e9, bps, alg / t / 8e12,
                                           alg / (ev[0] * 1e-3) / 8e12))


if __name__ == "__main__":
    main()
