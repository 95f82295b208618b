// sk_segment.hip -- the get_segs state machine (segmenter.py:420-470) on gfx950.
//
// The scan is inherently sequential inside a read (state: prev, err, prev_err, c, w, start,
// segs), so the parallelism is ACROSS reads: one lane per read, 64 reads per wavefront.  The
// per-sample threshold test was already done by the prep kernel, which left one bit per
// filtered sample in a transposed mask (word wi of read r at maskT[wi * rows + r]) so that the
// 64 lanes of a wave fetch 64 consecutive 8-byte words -- a coalesced 512-byte load per 64
// samples of 64 reads.  Integer state only; the two float compares of segmenter.py:431 live in
// the mask, and `c >= window * stall_len` (:448) is an integer compare against
// ceil(window * stall_len) because c is an integer.
#include "sk_common.h"
#include <math.h>

namespace {

struct WalkParams {
    int error, corrector, window, seg_dist, first_len;   // first_len = ceil(window * stall_len)
};

// Per-sample update written as straight-line selects: the 64 lanes of a wave are 64 different
// reads in 64 different states, so `if` ladders would execute every arm at every step.  Only the
// two rare events leave the straight line: closing a segment that is long enough to be reported
// (a handful per read) and the corrector test (dead unless error >= corrector).
__global__ __launch_bounds__(64)
void k_segment_walk(const uint64_t *__restrict__ maskT, int64_t mask_rows,
                    const sk_prep *__restrict__ prep, int nreads, WalkParams p,
                    int32_t *__restrict__ segs, int32_t *__restrict__ nsegs, int max_segs)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < nreads;
    const int n = live ? prep[r].n : 0;
    int32_t *my = segs + (int64_t)(live ? r : 0) * 2 * max_segs;

    int prev = 0;                                 // inside a candidate segment
    int err = 0, prev_err = 0, c = 0;
    int w = p.corrector;                          // segmenter.py:424 -- never reset inside a read
    int start = 0, nseg = 0, last_end = 0;

    int nmax = n;                                 // wave-uniform trip count (lanes past their n idle)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, __shfl_xor(nmax, d));
    const int nwords = (nmax + 63) >> 6;

    uint64_t next = (n > 0) ? maskT[r] : 0ull;
    for (int wi = 0; wi < nwords; wi++) {
        const uint64_t word = next;
        if ((wi + 1) * 64 < n) next = maskT[(int64_t)(wi + 1) * mask_rows + r];     // prefetch
        const unsigned half[2] = {(unsigned)word, (unsigned)(word >> 32)};
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned bits = half[h];
#pragma unroll 4
            for (int b = 0; b < 32; b++) {
                const int i = wi * 64 + h * 32 + b;
                const int valid = i < n;
                const int inb = (int)((bits >> b) & 1u) & valid;                   // :431 in band
                const int tol = (inb ^ 1) & prev & (int)(err < p.error) & valid;   // :442 tolerated
                const int act = inb | tol;
                const int closing = prev & (act ^ 1) & valid;                      // :448 / :458
                if (closing && (c >= p.window || (nseg == 0 && c >= p.first_len))) {
                    const int end = i - prev_err;                                  // :449
                    if (nseg > 0 && start - last_end < p.seg_dist) {               // :451 merge
                        if (nseg <= max_segs) my[2 * (nseg - 1) + 1] = end;
                    } else {
                        if (nseg < max_segs) { my[2 * nseg] = start; my[2 * nseg + 1] = end; }
                        nseg++;
                    }
                    last_end = end;
                }
                start = (inb & (prev ^ 1)) ? i : start;
                c = act ? c + 1 : (valid ? 0 : c);
                w += inb;
                err = tol ? err + 1 : (act ? err : (valid ? 0 : err));
                prev_err = tol ? prev_err + 1 : (valid ? 0 : prev_err);
                prev = valid ? act : prev;
                if (act && c >= p.window && c >= w) {                              // :439 / :446
                    if ((c % w) == 0) err--;
                }
            }
        }
    }
    if (live) nsegs[r] = nseg;                    // a segment still open at EOF is dropped (:466)
}

// dRNA_segmenter.py's slow5-branch scan (dRNA_segmenter.py:112-165): same skeleton, but the
// error budget is re-armed when a segment opens, errors only count from sample no_err_thresh
// on, `w` is a constant, and the scan stops ("adapter found") once the signal has been out of
// band for more than seg_dist samples after the last segment.
struct DrnaWalk { int error, no_err_thresh, w, window, seg_dist; };

__global__ __launch_bounds__(64)
void k_drna_walk(const uint64_t *__restrict__ maskT, int64_t mask_rows,
                 const sk_prep *__restrict__ prep, int nreads, DrnaWalk p,
                 int32_t *__restrict__ segs, int32_t *__restrict__ nsegs, int max_segs)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nreads) return;
    const int n = prep[r].n;
    int32_t *my = segs + (int64_t)r * 2 * max_segs;
    bool prev = false, done = false;
    int err = 0, prev_err = 0, c = 0, start = 0, nseg = 0, last_end = 0;
    for (int wi = 0; wi * 64 < n && !done; wi++) {
        const uint64_t word = maskT[(int64_t)wi * mask_rows + r];
        const int lim = min(64, n - wi * 64);
        for (int b = 0; b < lim; b++) {
            const int i = wi * 64 + b;
            if ((word >> b) & 1) {                                         // a < top  (:114)
                if (!prev) { start = i; prev = true; err = 0; }
                c++; prev_err = 0;
                if (c >= p.window && c >= p.w && (c % p.w) == 0) err--;
            } else if (prev) {
                if (err < p.error) {                                       // :129
                    c++;
                    if (i >= p.no_err_thresh) { err++; prev_err++; }
                    if (c >= p.window && c >= p.w && (c % p.w) == 0) err--;
                } else {
                    if (c >= p.window) {                                   // :137 close
                        const int end = i - prev_err;
                        if (nseg > 0 && start - last_end < p.seg_dist) {
                            if (nseg <= max_segs) my[2 * (nseg - 1) + 1] = end;
                        } else {
                            if (nseg < max_segs) { my[2 * nseg] = start; my[2 * nseg + 1] = end; }
                            nseg++;
                        }
                        last_end = end;
                    }
                    prev = false; c = 0; err = 0; prev_err = 0;
                }
            } else if (nseg > 0 && i - last_end > p.seg_dist) {            // :152 adapter found
                done = true;
                break;
            }
        }
    }
    nsegs[r] = nseg;
}

} // namespace

int sk_launch_drna_walk(sk_ctx *c, const uint64_t *d_mask, int64_t mask_rows, const sk_prep *d_prep,
                        int32_t nreads, const sk_drna_params *p, int32_t *d_segs, int32_t *d_nsegs,
                        int32_t max_segs)
{
    if (nreads <= 0) return SK_OK;
    DrnaWalk wp;
    wp.error = p->error; wp.no_err_thresh = p->no_err_thresh; wp.w = p->w; wp.window = p->window;
    wp.seg_dist = p->seg_dist;
    const int grid = (nreads + 63) / 64;
    SK_HIP(hipEventRecord(c->ev[2], c->stream));
    hipLaunchKernelGGL(k_drna_walk, dim3(grid), dim3(64), 0, c->stream, d_mask, mask_rows, d_prep, nreads, wp,
                       d_segs, d_nsegs, max_segs);
    SK_HIP(hipGetLastError());
    SK_HIP(hipEventRecord(c->ev[3], c->stream));
    return SK_OK;
}

int sk_launch_segment_walk(sk_ctx *c, const uint64_t *d_mask, int64_t mask_rows,
                           const int64_t *, const sk_prep *d_prep, int32_t nreads,
                           const sk_seg_params *p, int32_t *d_segs, int32_t *d_nsegs, int32_t max_segs)
{
    if (nreads <= 0) return SK_OK;
    WalkParams wp;
    wp.error = p->error; wp.corrector = p->corrector; wp.window = p->window; wp.seg_dist = p->seg_dist;
    const double fl = (double)p->window * p->stall_len;            // segmenter.py:448
    if (!(fl == fl))           wp.first_len = 0x7fffffff;          // NaN: never true
    else if (fl > 2147483000.) wp.first_len = 0x7fffffff;
    else if (fl < -2147483000.) wp.first_len = -0x7fffffff;
    else                       wp.first_len = (int)ceil(fl);
    const int grid = (nreads + 63) / 64;
    SK_HIP(hipEventRecord(c->ev[2], c->stream));
    hipLaunchKernelGGL(k_segment_walk, dim3(grid), dim3(64), 0, c->stream, d_mask, mask_rows, d_prep, nreads,
                       wp, d_segs, d_nsegs, max_segs);
    SK_HIP(hipGetLastError());
    SK_HIP(hipEventRecord(c->ev[3], c->stream));
    return SK_OK;
}
