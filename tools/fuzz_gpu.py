#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep over shapes the unit tests do not pin: random read counts, ragged
lengths, strides, motif lengths, outlier limits, both scalings, segmenter parameters.

    python tools/fuzz_gpu.py [seconds=120] [seed=1]

Prints one line per mismatch and exits non-zero if there was any.  (Test infrastructure.)"""
import os
import sys
import time

import numpy as np

os.environ["SK_TUNING"] = "1"      # this tool flips tuning switches

sys.path.insert(0, ".")
from squigglekit_amd import api, synth               # noqa: E402
from squigglekit_amd._lib import SegParams           # noqa: E402
from oracle import oracle as ora                      # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t_end = time.time() + budget
    bad = rounds = 0
    while time.time() < t_end:
        rounds += 1
        R = int(rng.choice([1, 3, 64, 255, 256, 300, 700, 1100]))
        M = int(rng.choice([8, 63, 512, 1000, 2047, 4000, 4001, 6000]))
        sig = synth.squiggle_batch(R, M, int(rng.integers(1 << 30)))
        if rng.random() < 0.3:
            sig = np.clip(sig, 300, 700).astype(np.int16)
        elif rng.random() < 0.3 and M >= 512:
            sig = synth.pattern_reads(rng, R, M)          # in-band masks built to defeat the jumping walk (round 4)
        k = int(rng.integers(0, 40))
        sig[rng.integers(0, R, k), rng.integers(0, M, k)] = rng.choice([-9, 0, 899, 900, 1199, 1200, 3000], k)
        lens = rng.integers(0, M + 1, R).astype(np.int32)
        lens[rng.integers(0, R)] = M
        # ---- MotifSeq ----
        # (round 3: every rows-per-lane instantiation of the screening kernels can come up -- 8 lanes x 1..32 rows up to
        # 256 points, 16 x 17..32 up to 512, 64 x 9..16 beyond -- not only the lengths the unit tests pin)
        N = int(rng.choice([1, 7, 16, 17, 100, 163, 200, 256, 257, 400, 512, 513, 1030])) if rng.random() < 0.5 \
            else int(rng.integers(1, 1025))
        motif = synth.synthetic_motif(N, seed=int(rng.integers(1000)))
        lo, hi = [(0, 1200), (0, 900), (-50, 2500), (400, 650)][int(rng.integers(4))]
        scale = ["medmad", "zscale"][int(rng.integers(2))]
        # the lanes-per-read layout of the screening scheme (8 is what large batches get by themselves: these batches
        # are small, so it is asked for), the fused / separate filter + statistics, the early / late exact retry
        import os
        ql = [None, "8", "16", "64"][int(rng.integers(4))]
        for key, val in (("SK_DTW_QL", ql), ("SK_DTW_NOFUSE", "1" if rng.random() < 0.3 else None),
                         ("SK_DTW_NO_EARLY", "1" if rng.random() < 0.3 else None),
                         ("SK_DTW_NO_SIBLINGS", "1" if rng.random() < 0.2 else None),  # two clusters: exact pass / second window
                         ("SK_DTW_SORT_MIN", "1" if rng.random() < 0.5 else None)):     # window passes in sorted order
            if val is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = val
        got = api.motifseq_batch(sig, lens, motif, scale=scale, scale_low=lo, scale_hi=hi)
        want = ora.motifseq_batch_i16(sig, lens, motif, scale_mode=0 if scale == "medmad" else 1, lo=lo, hi=hi)
        ok = ((got["flags"] & 2) == 0) & np.isfinite(want["dist"]) | (want["n"] == 0)   # MAD == 0 / std == 0 rows aside
        nan = np.isnan(got["dist"]) & np.isnan(want["dist"])
        same = (got["start"] == want["start"]) & (got["end"] == want["end"]) & (got["n"] == want["n"]) & \
               ((got["dist"] == want["dist"]) | nan)
        if not np.all(same[ok]):
            bad += 1
            r = int(np.nonzero(~same & ok)[0][0])
            print("MOTIFSEQ mismatch R=%d M=%d N=%d %s lo=%d hi=%d QL=%s read %d: got %s want %s"
                  % (R, M, N, scale, lo, hi, ql, r, got[r], want[r]))
        # ---- segmenter ----
        # (round 4: the default walk jumps between stretches of quiet mask entries, k_seg_walk4 -- window >= 127 and
        # error < min(corrector, 32) take it, with the statistics kernel's hints up to 4 096 samples and a pass of its
        # own beyond or under SK_WALK_OWNPASS; the other draws take the older walks)
        kw = [dict(), dict(error=10, corrector=3), dict(window=20, seg_dist=5), dict(std_scale=1.5, stall_len=0.9),
              dict(lim_low=300, lim_hi=800), dict(error=3), dict(window=127, stall_len=0.05), dict(error=0, seg_dist=0),
              dict(window=400, error=8, corrector=20), dict(std_scale=0.4, stall_len=1.2)][int(rng.integers(10))]
        if rng.random() < 0.25:
            os.environ["SK_WALK_OWNPASS"] = "1"
        p = SegParams(**kw)
        segs, nsegs = api.segment_batch(sig, lens, p, max_segs=64)
        okw = {a: b for a, b in kw.items() if a not in ("lim_low", "lim_hi")}
        osegs, onsegs = ora.segment_batch_i16(sig, lens, ora.SegParams(**okw), lo=p.lim_low, hi=p.lim_hi,
                                              max_segs=segs.shape[1])
        if not np.array_equal(nsegs, onsegs) or any(
                not np.array_equal(segs[r, :nsegs[r]], osegs[r, :nsegs[r]]) for r in range(R)):
            bad += 1
            print("SEGMENTER mismatch R=%d M=%d %s ownpass=%s" % (R, M, kw, os.environ.get("SK_WALK_OWNPASS")))
        os.environ.pop("SK_WALK_OWNPASS", None)
        # ---- float64 reads (pA-like), per-read oracle composition ----
        # (round 4: reads of up to 4 096 samples take the streaming statistics kernel, sk_f64stat.hip -- histogram median,
        # certified comparisons, numpy-order redo list; longer ones, and everything under SK_F64_OLD, the numpy-order
        # kernel.  Drawn per round: the data kind, segmenter parameters and limits, the certification margin blown up so
        # that every read takes the redo, the old kernel, and once in a while a read too long for the streaming path.)
        kinds = int(rng.integers(6))
        freads = []
        for _ in range(int(rng.choice([1, 3, 12, 70, 300]))):
            n = int(rng.choice([1, 2, 5, 63, 64, 65, 257, 1500, 4000, 4095, 4096]))
            if kinds == 0:
                x = np.round(rng.normal(90.0, 14.0, n), 1)                      # 0.1 pA steps: many ties
            elif kinds == 1:
                x = rng.normal(90.0, 14.0, n)
            elif kinds == 2:
                x = 500.0 + rng.integers(0, 3, n) * 2.0 ** -40                  # near constant
            elif kinds == 3:
                x = np.exp(rng.normal(4.0, 1.0, n))
            elif kinds == 4:                                                    # SquigglePull's pA image of a squiggle
                x = np.round((synth.squiggle_batch(1, n, int(rng.integers(1 << 30)))[0].astype(np.int64) + 16.0)
                             * (1493.94 / 8192.0), 2)
            else:                                                               # gridded, a far outlier, NaN / inf inside
                x = np.round(rng.normal(96.0, 9.0, n), 2)
                x[rng.integers(0, n)] = rng.choice([899.99, 0.01, np.nan, np.inf, -np.inf, 1199.0])
            freads.append(x)
        if rng.random() < 0.25:
            # a longer read decides the batch's kernel: the workgroup-per-read one (4 / 8 / 12 wavefronts by length, round 5),
            # the window-by-window one under SK_F64_LONG_LOOKS or beyond 41 472 samples
            top = int(rng.choice([9000, 10240, 12000, 20480, 30000, 41472, 45000]))
            freads.append(np.round(rng.normal(96.0, 15.0, int(rng.integers(4097, top + 1))), 2))
            if rng.random() < 0.5:
                freads.append(rng.normal(96.0, 15.0, int(rng.integers(4097, top + 1))))
        for key, val in (("SK_F64_OLD", "1" if rng.random() < 0.15 else None),
                         ("SK_F64_LONG_LOOKS", "1" if rng.random() < 0.2 else None),
                         ("SK_SEG_DELTA_SCALE", "1e13" if rng.random() < 0.2 else None)):
            if val is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = val
        fm = synth.synthetic_motif(int(rng.choice([3, 50, 163])), seed=int(rng.integers(1000)))
        fscale = ["medmad", "zscale"][int(rng.integers(2))]
        flo, fhi = [(0, 1200), (0, 900), (60, 130), (-1000, 1000)][int(rng.integers(4))]
        fgot = api.motifseq_reads_f64(freads, fm, scale=fscale, scale_low=flo, scale_hi=fhi)
        for i, x in enumerate(freads):
            f = ora.scale_outliers(x, flo, fhi)
            if f.size == 0:
                if not (fgot["flags"][i] & 1):
                    bad += 1
                    print("F64 empty read not flagged, kind %d" % kinds)
                continue
            y = ora.medmad(f)[0] if fscale == "medmad" else ora.zscale(f)[0]
            if not np.all(np.isfinite(y)):
                continue                                                         # MAD == 0: flagged, not compared
            d, s0, e0 = ora.dtw_subsequence(fm, y)
            if (fgot["dist"][i], fgot["start"][i], fgot["end"][i], fgot["n"][i]) != (d, s0, e0, f.size):
                bad += 1
                print("F64 mismatch kind %d %s n=%d lo=%d hi=%d: got %s want %s" % (kinds, fscale, len(x), flo, fhi, fgot[i], (d, s0, e0)))
        fkw = [dict(), dict(error=10, corrector=3), dict(window=20, seg_dist=5), dict(std_scale=0.2, stall_len=0.9),
               dict(lim_low=60, lim_hi=130), dict(std_scale=-0.3)][int(rng.integers(6))]
        fp = SegParams(**fkw)
        fokw = {a: b for a, b in fkw.items() if a not in ("lim_low", "lim_hi")}
        fsegs = api.segment_reads_f64(freads, fp)
        for x, gsegs in zip(freads, fsegs):
            f = ora.scale_outliers(x, fp.lim_low, fp.lim_hi)
            wsegs = ora.get_segs(f, ora.SegParams(**fokw)) if f.size else False
            if gsegs != wsegs:
                bad += 1
                print("F64 SEGMENTER mismatch kind %d n=%d %s" % (kinds, len(x), fkw))
        os.environ.pop("SK_F64_OLD", None)
        os.environ.pop("SK_F64_LONG_LOOKS", None)
        os.environ.pop("SK_SEG_DELTA_SCALE", None)
        # ---- round 6: raw rows through the pA conversion in the RAW domain (k_seg_stats<.., PA> / k_seg_stats_wg), the
        # wavefront-per-read walk, centi-unit batches -- against the oracle on the float64 values numpy makes the reference's way
        # (segmenter.py:345-349, :385) ----
        if rounds % 2 == 1:
            Rp = int(rng.choice([1, 5, 64, 200]))
            Mp = int(rng.choice([8, 512, 4000, 4096, 4104, 9000, 16384, 20000, 33000, 50000, 66000]))
            Sp = (Mp + 7) // 8 * 8
            praw = synth.squiggle_batch(Rp, Sp, int(rng.integers(1 << 30)))
            kind = int(rng.integers(5))
            if kind == 1:                                         # PromethION-like: raw window starts high
                praw = np.clip(np.rint(praw * 0.6 + 237), -32768, 32767).astype(np.int16)
            elif kind == 2 and Mp >= 512:
                praw = synth.pattern_reads(rng, Rp, Sp)
            elif kind == 3:                                       # spikes: kept ones above the window, dropped ones, negatives
                k2 = int(rng.integers(1, 60))
                praw[rng.integers(0, Rp, k2), rng.integers(0, Sp, k2)] = rng.choice([2500, 3000, 6000, 32767, -5, -300], k2)
            plens = rng.integers(0, Mp + 1, Rp).astype(np.int32)
            plens[rng.integers(0, Rp)] = Mp
            cal = np.empty((Rp, 3))
            cal[:, 0] = rng.choice([8192.0, 2048.0, 4096.0], Rp)
            cal[:, 1] = np.round(rng.uniform(-250, 40, Rp), 1) if kind == 1 else np.round(rng.uniform(-40, 40, Rp), 1)
            cal[:, 2] = rng.uniform(600, 1600, Rp)
            if rng.random() < 0.2:
                cal[rng.integers(0, Rp)] = [[8192.0, 16.0, 8192.0 * 5], [8192.0, 16.0, -1493.94], [8192.0, np.nan, 1493.94],
                                            [8192.0, 16.0, 8192.0 * 0.012]][int(rng.integers(4))]
            pkw = [dict(), dict(lim_low=60, lim_hi=140), dict(std_scale=0.3), dict(window=40, error=2), dict(error=40, corrector=10),
                   dict(lim_low=-50, lim_hi=2000), dict(std_scale=2.5, stall_len=0.9)][int(rng.integers(7))]
            for key, val in (("SK_SEG_DELTA_SCALE", "1e13" if rng.random() < 0.15 else None),
                             ("SK_WALK_NOWAVE", "1" if rng.random() < 0.3 else None),
                             ("SK_SEG_WG_ALL", "1" if rng.random() < 0.4 else None),
                             ("SK_SEG_NO_WG", "1" if rng.random() < 0.2 else None)):
                if val is None:
                    os.environ.pop(key, None)
                else:
                    os.environ[key] = val
            pp = SegParams(**pkw)
            psegs, pn = api.segment_batch_pa(praw, plens, cal, pp)
            opp = ora.SegParams(**{a: b for a, b in pkw.items() if a not in ("lim_low", "lim_hi")})
            for r in range(Rp):
                unit = float("{0:.2f}".format(cal[r, 2])) / cal[r, 0]
                pa = np.round((praw[r, :plens[r]].astype(np.int64) + cal[r, 1]) * unit, 2)
                f = ora.scale_outliers(pa, pp.lim_low, pp.lim_hi)
                w = (ora.get_segs(f, opp) or []) if f.size else []
                if psegs[r, :pn[r]].tolist() != w:
                    bad += 1
                    print("PA mismatch R=%d M=%d kind %d read %d cal %s %s env %s" % (Rp, Mp, kind, r, cal[r].tolist(), pkw,
                          {k: os.environ.get(k) for k in ("SK_SEG_DELTA_SCALE", "SK_WALK_NOWAVE", "SK_SEG_WG_ALL", "SK_SEG_NO_WG")}))
                    break
            # the same rows as --raw_signal input (int16 route) through the same kernel choices
            if Mp <= 50000:
                isegs, inn = api.segment_batch(praw, plens, SegParams(**{a: b for a, b in pkw.items() if a not in ("lim_low", "lim_hi")}), max_segs=64)
                osegs2, onn2 = ora.segment_batch_i16(praw, plens, opp, lo=0, hi=900, max_segs=isegs.shape[1])
                if not np.array_equal(inn, onn2) or any(not np.array_equal(isegs[r, :inn[r]], osegs2[r, :inn[r]]) for r in range(Rp)):
                    bad += 1
                    print("LONG I16 SEGMENTER mismatch R=%d M=%d kind %d %s" % (Rp, Mp, kind, pkw))
            for key in ("SK_SEG_DELTA_SCALE", "SK_WALK_NOWAVE", "SK_SEG_WG_ALL", "SK_SEG_NO_WG"):
                os.environ.pop(key, None)
            # centi-unit batch == float64 batch of the same tokens (both tools)
            cn = [rng.integers(-2000, 120000, int(rng.choice([1, 50, 3000, 4097]))).astype(np.int32) for _ in range(int(rng.choice([1, 7, 40])))]
            cflat = np.concatenate(cn)
            coff = np.concatenate([[0], np.cumsum([c.size for c in cn])]).astype(np.int64)
            ca, cb = api.segment_ragged_f64(cflat, coff), api.segment_ragged_f64(cflat / 100.0, coff)
            cm = synth.synthetic_motif(int(rng.choice([5, 60])), seed=3)
            ha, hb = api.motifseq_multi_ragged_f64(cflat, coff, [cm]), api.motifseq_multi_ragged_f64(cflat / 100.0, coff, [cm])
            if not (np.array_equal(ca[0], cb[0]) and np.array_equal(ca[1], cb[1]) and ha[0].tobytes() == hb[0].tobytes()):
                bad += 1
                print("CENTI mismatch: %d reads" % len(cn))
        # ---- dRNA_segmenter: both branches on a few long ragged reads ----
        if rounds % 2 == 0:
            from squigglekit_amd._lib import DrnaParams, RollParams
            dreads = synth.drna_reads(int(rng.integers(1, 9)), int(rng.integers(1 << 30)),
                                      min_len=int(rng.choice([300, 3000, 9000])), max_len=26000)
            if rng.random() < 0.5:                              # masks with structure (alternating, trains, near-miss gaps)
                dreads = dreads + [x for x in synth.pattern_reads(rng, int(rng.integers(1, 5)), int(rng.choice([3000, 9000])))]
            dkw = [dict(), dict(error=2, no_err_thresh=0, w=50, window=30, seg_dist=100),
                   dict(t_start=0, t_end=2000, std_scale=0.2), dict(error=0), dict(w=64, window=500, seg_dist=50),
                   dict(no_err_thresh=100000, error=1), dict(error=9, w=100, window=250, seg_dist=10, std_scale=1.5),
                   dict(no_err_thresh=777, w=128, window=64, error=3, t_start=0, t_end=9000)][int(rng.integers(8))]
            if rng.random() < 0.2:
                os.environ["SK_DRNA_STEP"] = "1"
            for x, g in zip(dreads, api.drna_segment_reads(dreads, DrnaParams(**dkw))):
                if g != ora.drna_segs(ora.scale_outliers(x.astype(float), 0, 1200), ora.DrnaParams(**dkw))[0]:
                    bad += 1
                    print("DRNA mismatch n=%d %s step=%s" % (len(x), dkw, os.environ.get("SK_DRNA_STEP")))
            rkw = [dict(), dict(w=int(rng.choice([3, 64, 999, 2000, 5000]))),
                   dict(w=800, lo_thresh=200, seg_dist=int(rng.choice([1, 300, 5000])), std_scale=0.25),
                   dict(w=int(rng.integers(1, 9000)), lo_thresh=int(rng.choice([5, 500, 2000])),
                        std_scale=float(rng.choice([-0.5, 0.0, 0.1, 0.5, 2.0])))][int(rng.integers(4))]
            # (round 5: rows of up to ~35 000 samples take the one-look kernel, k_roll_one; the two-kernel path on request)
            rsw = rng.random()
            rkey = "SK_ROLL_TWO_KERNELS" if rsw < 0.12 else "SK_ROLL_ONE_LOOK" if rsw < 0.3 else \
                   "SK_ROLL_DELTA_SCALE" if rsw < 0.4 else None
            if rkey:
                os.environ[rkey] = "1e13" if rkey == "SK_ROLL_DELTA_SCALE" else "1"
            for x, g in zip(dreads, api.drna_roll_reads(dreads, RollParams(**rkw))):
                if g != ora.drna_roll(ora.scale_outliers(x.astype(float), 0, 1200), ora.RollParams(**rkw)):
                    bad += 1
                    print("DRNA ROLL mismatch n=%d %s step=%s switch=%s" % (len(x), rkw, os.environ.get("SK_DRNA_STEP"), rkey))
            os.environ.pop("SK_DRNA_STEP", None)
            for key in ("SK_ROLL_TWO_KERNELS", "SK_ROLL_ONE_LOOK", "SK_ROLL_DELTA_SCALE"):
                os.environ.pop(key, None)
    print("fuzz: %d rounds, %d mismatching configurations" % (rounds, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
