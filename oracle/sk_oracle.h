/*
 * sk_oracle.h -- CPU restatement of the SquiggleKit signal-analysis hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * Pinning status
 *   segmenter path (filter, median, std, get_segs, test_segs): PINNED against
 *     golden vectors minted by importing /root/reference/segmenter.py
 *     (tools/gen_golden.py -> tests/golden/).
 *   MotifSeq normalisation (medmad / zscale): PINNED the same way
 *     (reference code + real numpy / sklearn).
 *   DTW core (mlpy 3.5.0 mlpy/dtw/cdtw.c, third party, NOT in /root/reference
 *     and not installable here): "PARITY UNPINNED" -- restated from mlpy's
 *     published algorithm; anchored on the reference's call site
 *     MotifSeq.py:437-439 and cross-checked against independent DP
 *     formulations in tests/.
 */
#ifndef SK_ORACLE_H
#define SK_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* numpy add.reduce pairwise summation (numpy/_core/src/umath/loops_utils.h). */
double ora_pairwise_sum(const double *a, int64_t n);
/* np.add.reduce as seen by np.mean/np.std: serial over 8192-element chunks of pairwise sums. */
double ora_np_sum(const double *a, int64_t n);
/* np.mean / np.std (ddof 0) / np.median on a float64 vector. */
double ora_mean(const double *x, int64_t n);
double ora_std(const double *x, int64_t n);
double ora_median(const double *x, int64_t n);

/* scale_outliers: segmenter.py:311-318, MotifSeq.py:317-324.
 * keeps lo < x < hi (strict), order preserving.  Returns survivors. */
int64_t ora_scale_outliers(const double *x, int64_t n, double lo, double hi, double *out);

typedef struct {
    int32_t error;      /* -e  segmenter.py:67  */
    int32_t corrector;  /* -c  segmenter.py:69  */
    int32_t window;     /* -w  segmenter.py:71  */
    int32_t seg_dist;   /* -d  segmenter.py:73  */
    double  std_scale;  /* -t  segmenter.py:75  */
    double  stall_len;  /* -l  segmenter.py:87  */
} ora_seg_params;

/* get_segs: segmenter.py:399-470.  sig is the already-filtered signal.
 * Writes up to max_segs [start,end] pairs; returns the number of segments
 * found (0 == the reference's `False`), or -1 on invalid parameters.
 * top/bot (may be NULL) receive the thresholds of segmenter.py:413-414. */
int32_t ora_get_segs(const double *sig, int64_t n, const ora_seg_params *p,
                     int32_t *segs, int32_t max_segs, double *top, double *bot);

/* medmad: MotifSeq.py:192-200.  out[i] = (x[i]-med)/(mad*1.4826). */
void ora_medmad(const double *x, int64_t n, double *out, double *med, double *smad);
/* zscale: MotifSeq.py:186-191 -> sklearn.preprocessing.scale (1.7.2).
 * Returns a bit mask of which "re-centre" guards fired (0 on sane data). */
int ora_zscale(const double *x, int64_t n, double *out, double *mean, double *scale);

/* mlpy 3.5.0 dtw_subsequence(x, y) restated: cdtw.c subsequence() fills the
 * full nx*ny float64 matrix (allocated per call like mlpy does), dtw.pyx takes
 * argmin of the last row, cdtw.c subsequence_path() back-traces.
 * cost (may be NULL) receives the nx*ny matrix; last_row (may be NULL) the
 * final row.  Returns 0, or -1 on allocation failure / empty input. */
int ora_dtw_subsequence(const double *x, int32_t nx, const double *y, int32_t ny,
                        double *dist, int32_t *start, int32_t *end,
                        double *cost, double *last_row);
/* Same result computed in O(nx) memory by forward start propagation. */
int ora_dtw_subsequence_fwd(const double *x, int32_t nx, const double *y, int32_t ny,
                            double *dist, int32_t *start, int32_t *end);
/* Full path (px, py) as mlpy returns it; returns path length or -1. */
int32_t ora_dtw_subsequence_path(const double *x, int32_t nx, const double *y, int32_t ny,
                                 int32_t *px, int32_t *py, int32_t cap);

/* dRNA_segmenter.py slow5 branch; defaults are its hard-coded values (dRNA_segmenter.py:80-104). */
typedef struct {
    int32_t error;          /* 5    */
    int32_t no_err_thresh;  /* 2500 */
    int32_t w;              /* 1200 (constant corrector) */
    int32_t window;         /* 100  */
    int32_t seg_dist;       /* 1200 */
    int32_t t_start;        /* 1000 */
    int32_t t_end;          /* 5000 */
    double  std_scale;      /* 0.8  */
} ora_drna_params;
int32_t ora_drna_segs(const double *sig, int64_t n, const ora_drna_params *p,
                      int32_t *segs, int32_t max_segs, double *top);

/* dRNA_segmenter.py:272-326, the --signal branch: rolling mean of the filtered signal
 * (pandas Series.rolling(window=w).mean(), restated from pandas/_libs/window/aggregations.pyx
 * roll_mean: Kahan-compensated add / remove, "consecutive same value" shortcut, min_periods = w),
 * mn = t.mean(), std = t.std() (pandas nanops: NaN -> 0, numpy sums, ddof = 1), bot = mn - 0.5 std,
 * the begin/end scan and the first segment whose length lies in [lo_thresh, hi_thresh].
 * `w` is undefined in the reference (the branch stops with NameError); the caller supplies it.
 * Returns 1 and (x, y) = (a - 1000, b - 1000) when a segment is printed, else 0.
 * t_out (n doubles, optional) receives the rolling mean; stats_out[3] = mn, std, bot. */
typedef struct {
    int32_t w;              /* rolling window: the script's commented default is 2000 */
    int32_t seg_dist;       /* 1500   */
    int32_t hi_thresh;      /* 200000 */
    int32_t lo_thresh;      /* 2000   */
    int32_t shift;          /* 1000: subtracted from both ends when printing */
    double  std_scale;      /* 0.5    */
} ora_roll_params;
int ora_drna_roll(const double *sig, int64_t n, const ora_roll_params *p, int64_t *x, int64_t *y,
                  double *t_out, double *stats_out);

/* 24-byte hit record shared with the product ABI. */
typedef struct { double dist; int32_t start, end, n, flags; } ora_hit;

/* Batch drivers over int16 squiggles (what bench.py times as cpu_baseline):
 * filter -> medmad(0)/zscale(1) -> dtw_subsequence, one read after another. */
int ora_motifseq_batch_i16(const int16_t *sig, int64_t stride, const int32_t *len, int32_t R,
                           const double *motif, int32_t nmotif, int scale_mode,
                           int32_t lo, int32_t hi, ora_hit *out);
/* [:-1] is NOT applied here; caller passes the length it wants. */
int ora_segment_batch_i16(const int16_t *sig, int64_t stride, const int32_t *len, int32_t R,
                          const ora_seg_params *p, int32_t lo, int32_t hi,
                          int32_t *segs, int32_t *nsegs, int32_t max_segs);

#ifdef __cplusplus
}
#endif
#endif
