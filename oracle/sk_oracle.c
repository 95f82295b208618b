/*
 * sk_oracle.c -- CPU restatement (plain C) of the SquiggleKit hot path.
 * TEST INFRASTRUCTURE ONLY -- see sk_oracle.h for who may use it and for the
 * pinning status (segmenter + normalisation pinned by reference goldens;
 * DTW core "parity unpinned": mlpy 3.5.0 is third party and absent).
 *
 * Build:  make -C oracle      (gcc -O2 -ffp-contract=off, no -ffast-math:
 * every double operation below must be one correctly rounded IEEE op).
 */
#include "sk_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* numpy reductions                                                     */
/* ------------------------------------------------------------------ */

/* numpy pairwise_sum (numpy/_core/src/umath/loops_utils.h.src, PW_BLOCKSIZE
 * 128): <8 serial from 0.0; <=128 eight strided accumulators combined as
 * ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) then the tail serially; otherwise split
 * at n/2 rounded down to a multiple of 8.  This is what np.mean / np.std use
 * at segmenter.py:409,412 and inside sklearn.scale (MotifSeq.py:187). */
double ora_pairwise_sum(const double *a, int64_t n)
{
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (int k = 0; k < 8; k++) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return ora_pairwise_sum(a, n2) + ora_pairwise_sum(a + n2, n - n2);
    }
}

/* np.add.reduce over a 1-D float64 vector as np.mean/np.std reach it: the
 * ufunc machinery feeds the inner loop at most NPY_BUFSIZE = 8192 elements at a
 * time, so the result is the SERIAL accumulation (from 0.0) of the pairwise sum
 * of each 8192-element chunk.  Verified bit-for-bit against numpy 2.2.6 for
 * n up to 60 000 (tests/test_oracle_golden.py). */
#define ORA_NPY_BUFSIZE 8192
double ora_np_sum(const double *a, int64_t n)
{
    double res = 0.0;
    for (int64_t i = 0; i < n; i += ORA_NPY_BUFSIZE) {
        int64_t m = n - i < ORA_NPY_BUFSIZE ? n - i : ORA_NPY_BUFSIZE;
        res += ora_pairwise_sum(a + i, m);
    }
    return res;
}

double ora_mean(const double *x, int64_t n)
{
    return ora_np_sum(x, n) / (double)n;
}

/* np.std, ddof 0 (numpy/_core/_methods.py _var/_std): mean = sum/n;
 * d = x-mean; d*d; pairwise sum; /n; sqrt.  segmenter.py:412. */
double ora_std(const double *x, int64_t n)
{
    if (n <= 0) return NAN;
    double mean = ora_mean(x, n);
    double *d = (double *)malloc(sizeof(double) * (size_t)n);
    if (!d) return NAN;
    for (int64_t i = 0; i < n; i++) {
        double t = x[i] - mean;
        d[i] = t * t;
    }
    double v = ora_np_sum(d, n) / (double)n;
    free(d);
    return sqrt(v);
}

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* np.median (segmenter.py:410, MotifSeq.py:194-195): middle element, or the
 * mean of the two middle elements (a+b)/2 for even n. */
double ora_median(const double *x, int64_t n)
{
    if (n <= 0) return NAN;
    double *s = (double *)malloc(sizeof(double) * (size_t)n);
    if (!s) return NAN;
    memcpy(s, x, sizeof(double) * (size_t)n);
    qsort(s, (size_t)n, sizeof(double), cmp_double);
    double m;
    if (n & 1) m = s[n / 2];
    else       m = (s[n / 2 - 1] + s[n / 2]) / 2.0;
    free(s);
    return m;
}

/* ------------------------------------------------------------------ */
/* scale_outliers  (segmenter.py:311-318 ; MotifSeq.py:317-324)         */
/* ------------------------------------------------------------------ */
int64_t ora_scale_outliers(const double *x, int64_t n, double lo, double hi, double *out)
{
    int64_t k = 0;
    for (int64_t i = 0; i < n; i++)
        if (x[i] > lo && x[i] < hi) out[k++] = x[i];
    return k;
}

/* ------------------------------------------------------------------ */
/* get_segs  (segmenter.py:399-470)                                     */
/* ------------------------------------------------------------------ */
int32_t ora_get_segs(const double *sig, int64_t n, const ora_seg_params *p,
                     int32_t *segs, int32_t max_segs, double *top_out, double *bot_out)
{
    if (!p || p->corrector < 0) return -1;   /* reference would hit c % 0 / negative w */
    if (n <= 0) return 0;                    /* reference raises in sig.min(); we say "none" */

    /* segmenter.py:410-414 */
    double median = ora_median(sig, n);
    double stdev  = ora_std(sig, n);
    double top = median + (stdev * p->std_scale);
    double bot = median - (stdev * p->std_scale);
    if (top_out) *top_out = top;
    if (bot_out) *bot_out = bot;

    /* segmenter.py:420-428 */
    int     prev = 0;
    int64_t err = 0, prev_err = 0, c = 0;
    int64_t w = p->corrector;
    int64_t start = 0, end = 0;
    int32_t nseg = 0;
    int64_t last_end = 0;                    /* segs[-1][1] */
    const double first_len = (double)p->window * p->stall_len;   /* :448 */

    for (int64_t i = 0; i < n; i++) {        /* :429 */
        double a = sig[i];
        if (a < top && a > bot) {            /* :431 */
            if (!prev) { start = i; prev = 1; }
            c += 1;
            w += 1;
            if (prev_err) prev_err = 0;
            if (c >= p->window && c >= w && (c % w) == 0) err -= 1;      /* :439 */
        } else {
            if (prev && err < p->error) {    /* :442 */
                c += 1; err += 1; prev_err += 1;
                if (c >= p->window && c >= w && (c % w) == 0) err -= 1;  /* :446 */
            } else if (prev && (c >= p->window || (nseg == 0 && (double)c >= first_len))) { /* :448 */
                end = i - prev_err;          /* :449 */
                prev = 0;
                if (nseg > 0 && start - last_end < p->seg_dist) {        /* :451 */
                    if (nseg <= max_segs) segs[2 * (nseg - 1) + 1] = (int32_t)end;
                } else {
                    if (nseg < max_segs) {
                        segs[2 * nseg] = (int32_t)start;
                        segs[2 * nseg + 1] = (int32_t)end;
                    }
                    nseg++;
                }
                last_end = end;
                c = 0; err = 0; prev_err = 0;
            } else if (prev) {               /* :458 */
                prev = 0; c = 0; err = 0; prev_err = 0;
            }
        }
    }
    /* :466 -- a segment still open at EOF is dropped */
    return nseg;
}

/* ------------------------------------------------------------------ */
/* MotifSeq normalisation                                               */
/* ------------------------------------------------------------------ */

/* MotifSeq.py:192-200 */
void ora_medmad(const double *x, int64_t n, double *out, double *med_out, double *smad_out)
{
    double med = ora_median(x, n);
    double *dev = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) dev[i] = fabs(x[i] - med);
    double mad = ora_median(dev, n);
    free(dev);
    double smad = mad * 1.4826;
    if (out) for (int64_t i = 0; i < n; i++) out[i] = (x[i] - med) / smad;
    if (med_out) *med_out = med;
    if (smad_out) *smad_out = smad;
}

/* np.allclose(m, 0) with default rtol=1e-5, atol=1e-8: |m - 0| <= atol + rtol*|0| */
static int allclose0(double m) { return isfinite(m) && fabs(m) <= 1e-8; }

/* MotifSeq.py:186-191 -> sklearn.preprocessing.scale(axis=0, with_mean, with_std, copy)
 * (sklearn 1.7.2 preprocessing/_data.py): Xr -= mean; [guard 1]; Xr /= std (0 -> 1);
 * [guard 2]. */
int ora_zscale(const double *x, int64_t n, double *out, double *mean_out, double *scale_out)
{
    int fired = 0;
    double mean = ora_mean(x, n);
    double sd = ora_std(x, n);
    for (int64_t i = 0; i < n; i++) out[i] = x[i] - mean;
    double m1 = ora_mean(out, n);
    if (!allclose0(m1)) { fired |= 1; for (int64_t i = 0; i < n; i++) out[i] -= m1; }
    double sc = (sd == 0.0) ? 1.0 : sd;
    for (int64_t i = 0; i < n; i++) out[i] /= sc;
    double m2 = ora_mean(out, n);
    if (!allclose0(m2)) { fired |= 2; for (int64_t i = 0; i < n; i++) out[i] -= m2; }
    if (mean_out) *mean_out = mean;
    if (scale_out) *scale_out = sc;
    return fired;
}

/* ------------------------------------------------------------------ */
/* mlpy 3.5.0 dtw_subsequence  (call site MotifSeq.py:437)              */
/* ------------------------------------------------------------------ */

/* cdtw.c min3: a; if b<a; if c<that */
static inline double min3(double a, double b, double c)
{
    double m = a;
    if (b < m) m = b;
    if (c < m) m = c;
    return m;
}

/* cdtw.c subsequence(): Manhattan local cost, free start along y. */
static void fill_cost(const double *x, int32_t n, const double *y, int32_t m, double *cost)
{
    cost[0] = fabs(x[0] - y[0]);
    for (int32_t i = 1; i < n; i++)
        cost[(size_t)i * m] = fabs(x[i] - y[0]) + cost[(size_t)(i - 1) * m];
    for (int32_t j = 1; j < m; j++)
        cost[j] = fabs(x[0] - y[j]);
    for (int32_t i = 1; i < n; i++) {
        const double *up = cost + (size_t)(i - 1) * m;
        double *row = cost + (size_t)i * m;
        for (int32_t j = 1; j < m; j++)
            row[j] = fabs(x[i] - y[j]) + min3(up[j], up[j - 1], row[j - 1]);
    }
}

/* dtw.pyx: idx = np.argmin(cost[-1, :]) -- first minimum. */
static int32_t argmin_first(const double *row, int32_t m)
{
    int32_t idx = 0;
    double best = row[0];
    if (isnan(best)) return 0;
    for (int32_t j = 1; j < m; j++) {
        if (isnan(row[j])) return j;         /* np.argmin returns the first NaN */
        if (row[j] < best) { best = row[j]; idx = j; }
    }
    return idx;
}

/* cdtw.c subsequence_path(): from (n-1, starty) while i > 0; j==0 -> i--;
 * else diag==min -> (i--,j--); elif left(j-1)==min -> j--; else i--.
 * Returns the path length; px/py (may be NULL) are filled in reverse order
 * (end of path first) up to cap entries; *j_at_row0 gets path_y[0]. */
static int32_t backtrace(const double *cost, int32_t n, int32_t m, int32_t starty,
                         int32_t *px, int32_t *py, int32_t cap, int32_t *j_at_row0)
{
    int32_t i = n - 1, j = starty, k = 0;
    if (px && k < cap) { px[k] = i; py[k] = j; }
    k++;
    while (i > 0) {
        if (j == 0) {
            i--;
        } else {
            double up = cost[(size_t)(i - 1) * m + j];
            double dg = cost[(size_t)(i - 1) * m + (j - 1)];
            double lf = cost[(size_t)i * m + (j - 1)];
            double mc = min3(up, dg, lf);
            if (dg == mc)      { i--; j--; }
            else if (lf == mc) { j--; }
            else               { i--; }
        }
        if (px && k < cap) { px[k] = i; py[k] = j; }
        k++;
    }
    if (j_at_row0) *j_at_row0 = j;
    return k;
}

int ora_dtw_subsequence(const double *x, int32_t nx, const double *y, int32_t ny,
                        double *dist, int32_t *start, int32_t *end,
                        double *cost_out, double *last_row)
{
    if (nx <= 0 || ny <= 0) return -1;
    /* mlpy allocates the full matrix on every call (np.empty((n, m))). */
    double *cost = cost_out ? cost_out : (double *)malloc(sizeof(double) * (size_t)nx * (size_t)ny);
    if (!cost) return -1;
    fill_cost(x, nx, y, ny, cost);
    const double *last = cost + (size_t)(nx - 1) * ny;
    int32_t idx = argmin_first(last, ny);
    int32_t s = 0;
    backtrace(cost, nx, ny, idx, NULL, NULL, 0, &s);
    if (dist) *dist = last[idx];
    if (start) *start = s;          /* path[1][0]  MotifSeq.py:438 */
    if (end) *end = idx;            /* path[1][-1] MotifSeq.py:439 */
    if (last_row) memcpy(last_row, last, sizeof(double) * (size_t)ny);
    if (!cost_out) free(cost);
    return 0;
}

int32_t ora_dtw_subsequence_path(const double *x, int32_t nx, const double *y, int32_t ny,
                                 int32_t *px, int32_t *py, int32_t cap)
{
    if (nx <= 0 || ny <= 0) return -1;
    double *cost = (double *)malloc(sizeof(double) * (size_t)nx * (size_t)ny);
    if (!cost) return -1;
    fill_cost(x, nx, y, ny, cost);
    int32_t idx = argmin_first(cost + (size_t)(nx - 1) * ny, ny);
    int32_t k = backtrace(cost, nx, ny, idx, px, py, cap, NULL);
    free(cost);
    /* mlpy reverses so the path runs from row 0 to row n-1 */
    int32_t kk = k < cap ? k : cap;
    for (int32_t a = 0, b = kk - 1; a < b; a++, b--) {
        int32_t t = px[a]; px[a] = px[b]; px[b] = t;
        t = py[a]; py[a] = py[b]; py[b] = t;
    }
    return k;
}

/* O(nx) memory: one column at a time, each cell carries the column where its
 * back-trace would reach row 0 (tie order of subsequence_path: diag, left, up). */
int ora_dtw_subsequence_fwd(const double *x, int32_t nx, const double *y, int32_t ny,
                            double *dist, int32_t *start, int32_t *end)
{
    if (nx <= 0 || ny <= 0) return -1;
    double  *D = (double *)malloc(sizeof(double) * (size_t)nx);
    int32_t *S = (int32_t *)malloc(sizeof(int32_t) * (size_t)nx);
    if (!D || !S) { free(D); free(S); return -1; }
    double best = INFINITY; int32_t bj = 0, bs = 0; int seen = 0;
    for (int32_t j = 0; j < ny; j++) {
        double dgD = 0.0; int32_t dgS = 0;     /* (i-1, j-1) */
        for (int32_t i = 0; i < nx; i++) {
            double c = fabs(x[i] - y[j]);
            double newD; int32_t newS;
            if (i == 0) { newD = c; newS = j; }
            else if (j == 0) { newD = c + D[i - 1]; newS = S[i - 1]; }
            else {
                double up = D[i - 1], lf = D[i];
                int32_t upS = S[i - 1], lfS = S[i];
                double mc = min3(up, dgD, lf);
                if (dgD == mc)     newS = dgS;
                else if (lf == mc) newS = lfS;
                else               newS = upS;
                newD = c + mc;
            }
            dgD = D[i]; dgS = S[i];            /* old (i, j-1) becomes diag of (i+1, j) */
            D[i] = newD; S[i] = newS;
        }
        double v = D[nx - 1];
        if (!seen || v < best) { best = v; bj = j; bs = S[nx - 1]; seen = 1; }
    }
    if (dist) *dist = best;
    if (start) *start = bs;
    if (end) *end = bj;
    free(D); free(S);
    return 0;
}

/* ------------------------------------------------------------------ */
/* batch drivers (cpu_baseline legs of bench.py; parity oracles)        */
/* ------------------------------------------------------------------ */

int ora_motifseq_batch_i16(const int16_t *sig, int64_t stride, const int32_t *len, int32_t R,
                           const double *motif, int32_t nmotif, int scale_mode,
                           int32_t lo, int32_t hi, ora_hit *out)
{
    for (int32_t r = 0; r < R; r++) {
        const int16_t *s = sig + (size_t)r * stride;
        int32_t m = len[r];
        double *raw = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
        double *nrm = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
        if (!raw || !nrm) { free(raw); free(nrm); return -1; }
        int64_t n = 0;
        for (int32_t i = 0; i < m; i++)
            if (s[i] > lo && s[i] < hi) raw[n++] = (double)s[i];   /* MotifSeq.py:274 */
        ora_hit h; h.dist = NAN; h.start = -1; h.end = -1; h.n = (int32_t)n; h.flags = 0;
        if (n > 0) {
            if (scale_mode == 0) ora_medmad(raw, n, nrm, NULL, NULL);
            else h.flags = ora_zscale(raw, n, nrm, NULL, NULL);
            ora_dtw_subsequence(motif, nmotif, nrm, (int32_t)n, &h.dist, &h.start, &h.end, NULL, NULL);
        }
        out[r] = h;
        free(raw); free(nrm);
    }
    return 0;
}

int ora_segment_batch_i16(const int16_t *sig, int64_t stride, const int32_t *len, int32_t R,
                          const ora_seg_params *p, int32_t lo, int32_t hi,
                          int32_t *segs, int32_t *nsegs, int32_t max_segs)
{
    for (int32_t r = 0; r < R; r++) {
        const int16_t *s = sig + (size_t)r * stride;
        int32_t m = len[r];
        double *raw = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
        if (!raw) return -1;
        int64_t n = 0;
        for (int32_t i = 0; i < m; i++)
            if (s[i] > lo && s[i] < hi) raw[n++] = (double)s[i];   /* segmenter.py:209 */
        int32_t k = ora_get_segs(raw, n, p, segs + (size_t)r * 2 * max_segs, max_segs, NULL, NULL);
        free(raw);
        if (k < 0) return -1;
        nsegs[r] = k;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* dRNA_segmenter.py, slow5 branch (dRNA_segmenter.py:85-176)           */
/* ------------------------------------------------------------------ */
/* sig: already filtered by its scale_outliers (0 < x < 1200, :329-332).  Thresholds come from
 * the slice sig[t_start:t_end] (:109-111); the scan is one-sided (a < top), errors only count
 * from sample no_err_thresh on, `w` is a constant, and the scan stops once the signal has left
 * the last segment by more than seg_dist (:152-159).  Returns the number of segments collected
 * before the scan stopped (the script prints only the first one, :173-176). */
int32_t ora_drna_segs(const double *sig, int64_t n, const ora_drna_params *p,
                      int32_t *segs, int32_t max_segs, double *top_out)
{
    if (!p || p->w <= 0) return -1;
    int64_t a0 = p->t_start < n ? p->t_start : n, a1 = p->t_end < n ? p->t_end : n;
    if (a0 < 0) a0 = 0;
    if (a1 < a0) a1 = a0;
    double median = ora_median(sig + a0, a1 - a0);      /* NaN on an empty slice, like numpy */
    double stdev = ora_std(sig + a0, a1 - a0);
    double top = median + (stdev * p->std_scale);
    if (top_out) *top_out = top;

    int prev = 0;
    int64_t err = 0, prev_err = 0, c = 0, start = 0, last_end = 0;
    int32_t nseg = 0;
    for (int64_t i = 0; i < n; i++) {
        double a = sig[i];
        if (a < top) {
            if (!prev) { start = i; prev = 1; err = 0; }
            c += 1;
            if (prev_err) prev_err = 0;
            if (c >= p->window && c >= p->w && (c % p->w) == 0) err -= 1;
        } else {
            if (prev && err < p->error) {
                c += 1;
                if (i >= p->no_err_thresh) { err += 1; prev_err += 1; }
                if (c >= p->window && c >= p->w && (c % p->w) == 0) err -= 1;
            } else if (prev) {
                if (c >= p->window) {
                    int64_t end = i - prev_err;
                    if (nseg > 0 && start - last_end < p->seg_dist) {
                        if (nseg <= max_segs) segs[2 * (nseg - 1) + 1] = (int32_t)end;
                    } else {
                        if (nseg < max_segs) { segs[2 * nseg] = (int32_t)start; segs[2 * nseg + 1] = (int32_t)end; }
                        nseg++;
                    }
                    last_end = end;
                }
                prev = 0; c = 0; err = 0; prev_err = 0;
            } else if (nseg > 0 && i - last_end > p->seg_dist) {
                break;                                   /* adapter found */
            }
        }
    }
    return nseg;
}


/* ------------------------------------------------------------------------------------------------
 * dRNA_segmenter.py:272-326 -- the --signal branch (rolling mean)
 * ------------------------------------------------------------------------------------------------ */
/* pandas/_libs/window/aggregations.pyx: add_mean / remove_mean / calc_mean */
typedef struct { int64_t nobs, neg_ct, same; double sum_x, comp_add, comp_rem, prev; } roll_state;
static void roll_add(roll_state *s, double val)
{
    if (val == val) {
        s->nobs += 1;
        double y = val - s->comp_add;
        double t = s->sum_x + y;
        s->comp_add = t - s->sum_x - y;
        s->sum_x = t;
        if (signbit(val)) s->neg_ct += 1;
        if (val == s->prev) s->same += 1; else s->same = 1;
        s->prev = val;
    }
}
static void roll_remove(roll_state *s, double val)
{
    if (val == val) {
        s->nobs -= 1;
        double y = -val - s->comp_rem;
        double t = s->sum_x + y;
        s->comp_rem = t - s->sum_x - y;
        s->sum_x = t;
        if (signbit(val)) s->neg_ct -= 1;
    }
}
static double roll_calc(const roll_state *s, int64_t minp)
{
    if (s->nobs >= minp && s->nobs > 0) {
        double r = s->sum_x / (double)s->nobs;
        if (s->same >= s->nobs) r = s->prev;
        else if (s->neg_ct == 0 && r < 0) r = 0;
        else if (s->neg_ct == s->nobs && r > 0) r = 0;
        return r;
    }
    return NAN;
}

int ora_drna_roll(const double *sig, int64_t n, const ora_roll_params *p, int64_t *x_out, int64_t *y_out,
                  double *t_out, double *stats_out)
{
    if (!p || p->w <= 0) return -1;
    const int64_t w = p->w;
    double *t = t_out ? t_out : (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    /* fixed window, closed on the right: start = max(0, i - w + 1), end = i + 1 (monotonic bounds) */
    roll_state st;
    memset(&st, 0, sizeof st);
    for (int64_t i = 0; i < n; i++) {
        const int64_t s0 = (i - w + 1 > 0) ? i - w + 1 : 0;
        if (i == 0) {
            memset(&st, 0, sizeof st);
            st.prev = sig[s0];
            st.same = 0;
            roll_add(&st, sig[0]);
        } else {
            const int64_t sp = (i - w > 0) ? i - w : 0;          /* start[i-1] */
            for (int64_t j = sp; j < s0; j++) roll_remove(&st, sig[j]);
            roll_add(&st, sig[i]);                                /* end[i-1] = i .. end[i] = i + 1 */
        }
        t[i] = roll_calc(&st, w);
    }
    /* t.mean(), t.std(): pandas.core.nanops (no bottleneck): NaN -> 0 copies, numpy sums */
    double *v = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; i++) { if (t[i] == t[i]) { v[i] = t[i]; cnt++; } else v[i] = 0.0; }
    const double mn = ora_np_sum(v, n) / (double)cnt;             /* count 0 -> NaN like pandas */
    for (int64_t i = 0; i < n; i++) {
        if (t[i] == t[i]) { const double d = mn - t[i]; v[i] = d * d; } else v[i] = 0.0;
    }
    const double var = ora_np_sum(v, n) / (double)(cnt - 1);      /* ddof = 1 */
    const double sd = sqrt(var);
    const double bot = mn - (sd * p->std_scale);
    free(v);
    if (stats_out) { stats_out[0] = mn; stats_out[1] = sd; stats_out[2] = bot; }

    /* the scan (dRNA_segmenter.py:296-316) */
    int begin = 0, found = 0;
    int64_t start = 0, end = 0, last_end = 0, nseg = 0;
    int64_t cap = 64, *segs = (int64_t *)malloc((size_t)cap * 2 * sizeof(int64_t));
    for (int64_t c = 0; c < n; c++) {
        const double i = t[c];
        if (i < bot && !begin) { start = c; begin = 1; }
        else if (i < bot) { end = c; }
        else if (i > bot && begin) {
            if (nseg > 0 && start - last_end < p->seg_dist) segs[2 * (nseg - 1) + 1] = end;
            else {
                if (nseg == cap) { cap *= 2; segs = (int64_t *)realloc(segs, (size_t)cap * 2 * sizeof(int64_t)); }
                segs[2 * nseg] = start; segs[2 * nseg + 1] = end; nseg++;
            }
            last_end = end;
            start = 0; end = 0; begin = 0;
        }
    }
    for (int64_t k = 0; k < nseg; k++) {                          /* :318-326 */
        const int64_t a = segs[2 * k], b = segs[2 * k + 1];
        if (b - a > p->hi_thresh) continue;
        if (b - a < p->lo_thresh) continue;
        if (x_out) *x_out = a - p->shift;
        if (y_out) *y_out = b - p->shift;
        found = 1;
        break;
    }
    free(segs);
    if (!t_out) free(t);
    return found;
}
