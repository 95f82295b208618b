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
#include <stdlib.h>

namespace {

struct WalkParams {
    int error, corrector, window, seg_dist, first_len;   // first_len = ceil(window * stall_len)
};

// Per-sample update written as straight-line arithmetic: the 64 lanes of a wave are 64 different
// reads in 64 different states, so `if` ladders would execute every arm at every step.  Only the
// rare events leave the straight line: closing a segment that is long enough to be reported
// (a handful per read) and, in the general variant, the corrector test.
//
// FAST variant (error < corrector, the defaults included): the corrector test of
// segmenter.py:439/446 can never fire -- c <= (in-band samples of this segment) + err and
// w = corrector + (all in-band samples so far), so c >= w needs err >= corrector -- hence err
// never decreases, `w` needs no tracking, and every state variable is a product/sum of 0/1 flags:
//     c' = c*act + act      err' = err*act + tol      prev_err' = prev_err*tol + tol
// A segment's `start` is not stored either: from the opening sample on every step is `act`
// until the one that closes it, so start == i - c there.  11 vector instructions per sample.
struct WalkState {
    int prev, err, prev_err, c, w, start, nseg, last_end;
};

__device__ __forceinline__ void report_segment(WalkState &st, int start, int end, const WalkParams &p,
                                               int32_t *my, int max_segs)
{
    if (st.nseg > 0 && start - st.last_end < p.seg_dist) {                         // :451 merge
        if (st.nseg <= max_segs) my[2 * (st.nseg - 1) + 1] = end;
    } else {
        if (st.nseg < max_segs) { my[2 * st.nseg] = start; my[2 * st.nseg + 1] = end; }
        st.nseg++;
    }
    st.last_end = end;
}

// v_mad_u32_u24 / v_mul_u32_u24 (one issue slot each; the compiler will not pick them on its own)
__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned mul24(unsigned a, unsigned b)
{
    unsigned r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// one 32-sample half word, every lane's samples valid, corrector dead
__device__ __forceinline__ void walk_fast32(WalkState &st, unsigned bits, int i0, const WalkParams &p,
                                            int thr_first, int32_t *my, int max_segs)
{
    unsigned prev = (unsigned)st.prev, err = (unsigned)st.err, perr = (unsigned)st.prev_err, c = (unsigned)st.c;
    // report threshold: window, or min(window, first_len) until the first segment (:448);
    // 0x7fffffff for lanes that own no read
    unsigned thr = (st.nseg == 0) ? (unsigned)thr_first : (unsigned)p.window;
#pragma unroll 16
    for (int b = 0; b < 32; b++) {
        const unsigned inb = (bits >> b) & 1u;                                     // :431 in band
        const unsigned ltm = (unsigned)(((int)err - p.error) >> 31);               // all ones: err < error
        const unsigned tol = prev & ~inb & ltm;                                    // :442 tolerated
        const unsigned closing = prev & ~inb & ~ltm;                               // :448 / :458
        const unsigned act = inb | tol;
        if (mul24(closing, c) >= thr) {                                            // thr >= 1
            report_segment(st, i0 + b - (int)c, i0 + b - (int)perr, p, my, max_segs);   // :449
            thr = (unsigned)p.window;
        }
        c = mad24(c, act, act);
        err = mad24(err, act, tol);
        perr = mad24(perr, tol, tol);
        prev = act;
    }
    st.prev = (int)prev; st.err = (int)err; st.prev_err = (int)perr; st.c = (int)c;
}

// general step (any parameters, samples past a lane's n ignored); keeps `start` and `w`
__device__ __forceinline__ void walk_general32(WalkState &st, unsigned bits, int i0, int n, const WalkParams &p,
                                               int32_t *my, int max_segs)
{
    int prev = st.prev, err = st.err, prev_err = st.prev_err, c = st.c, w = st.w, start = st.start;
#pragma unroll 4
    for (int b = 0; b < 32; b++) {
        const int i = i0 + b;
        const int valid = i < n;
        const int inb = (int)((bits >> b) & 1u) & valid;                           // :431 in band
        const int tol = (inb ^ 1) & prev & (int)(err < p.error) & valid;           // :442 tolerated
        const int act = inb | tol;
        const int closing = prev & (act ^ 1) & valid;                              // :448 / :458
        if (closing && (c >= p.window || (st.nseg == 0 && c >= p.first_len)))
            report_segment(st, start, i - prev_err, p, my, max_segs);              // :449
        start = (inb & (prev ^ 1)) ? i : start;
        c = act ? c + 1 : (valid ? 0 : c);
        w += inb;
        err = tol ? err + 1 : (act ? err : (valid ? 0 : err));
        prev_err = tol ? prev_err + 1 : (valid ? 0 : prev_err);
        prev = valid ? act : prev;
        if (act && c >= p.window && c >= w) {                                      // :439 / :446
            if ((c % w) == 0) err--;
        }
    }
    st.prev = prev; st.err = err; st.prev_err = prev_err; st.c = c; st.w = w; st.start = start;
}

template <bool FAST>
__global__ __launch_bounds__(64)
void k_segment_walk(const uint64_t *__restrict__ maskT, int64_t mask_rows,
                    const sk_prep *__restrict__ prep, int nreads, WalkParams p,
                    int32_t *__restrict__ segs, int32_t *__restrict__ nsegs, int max_segs)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < nreads;
    const int n = live ? prep[r].n : 0;
    int32_t *my = segs + (int64_t)(live ? r : 0) * 2 * max_segs;

    WalkState st;
    st.prev = 0;                                  // inside a candidate segment
    st.err = 0; st.prev_err = 0; st.c = 0;
    st.w = p.corrector;                           // segmenter.py:424 -- never reset inside a read
    st.start = 0; st.nseg = 0; st.last_end = 0;

    int nmax = n;                                 // wave-uniform trip count (lanes past their n idle)
    int nmin = (n > 0) ? n : 0x7fffffff;          // shortest read of the wave: words all lanes own
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        nmax = max(nmax, __shfl_xor(nmax, d));
        nmin = min(nmin, __shfl_xor(nmin, d));
    }
    const int nwords = (nmax + 63) >> 6;
    // the fast steps use 24-bit multiplies on counters bounded by n
    const int nfast = (FAST && nmax < (1 << 24)) ? min(nwords, nmin >> 6) : 0;
    const int thr_first = (n > 0) ? min(p.window, p.first_len) : 0x7fffffff;

    uint64_t next = (n > 0) ? maskT[r] : 0ull;
    for (int wi = 0; wi < nwords; wi++) {
        const uint64_t word = next;
        if ((wi + 1) * 64 < n) next = maskT[(int64_t)(wi + 1) * mask_rows + r];     // prefetch
        if (wi < nfast) {
            walk_fast32(st, (unsigned)word, wi * 64, p, thr_first, my, max_segs);
            walk_fast32(st, (unsigned)(word >> 32), wi * 64 + 32, p, thr_first, my, max_segs);
            if (wi + 1 == nfast) {                // hand over to the general steps
                st.start = (wi + 1) * 64 - st.c;
                st.w = 0x7fffffff;                // the corrector test stays dead
            }
        } else {
            walk_general32(st, (unsigned)word, wi * 64, n, p, my, max_segs);
            walk_general32(st, (unsigned)(word >> 32), wi * 64 + 32, n, p, my, max_segs);
        }
    }
    if (live) nsegs[r] = st.nseg;                 // a segment still open at EOF is dropped (:466)
}

// dRNA_segmenter.py's slow5-branch scan (dRNA_segmenter.py:112-165): same skeleton, but the
// error budget is re-armed when a segment opens, errors only count from sample no_err_thresh
// on, `w` is a constant, and the scan stops ("adapter found") once the signal has been out of
// band for more than seg_dist samples after the last segment.
struct DrnaWalk { int error, no_err_thresh, w, window, seg_dist; };

// Straight-line per-sample update (the 64 lanes are 64 reads in different states): flags are 0/1
// integers, `cm` tracks c mod w so that the corrector test `c % w == 0` (:121 / :133) needs no division,
// and only "close a segment that is long enough" branches.  A lane whose scan has stopped ("adapter
// found", :152) or whose read has ended just stops changing; the wave skips words in which no lane
// can change state.
__global__ __launch_bounds__(64)
void k_drna_walk(const uint64_t *__restrict__ maskT, int64_t mask_rows,
                 const sk_prep *__restrict__ prep, int nreads, DrnaWalk p,
                 int32_t *__restrict__ segs, int32_t *__restrict__ nsegs, int max_segs)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < nreads;
    const int n = live ? prep[r].n : 0;
    int32_t *my = segs + (int64_t)(live ? r : 0) * 2 * max_segs;
    int prev = 0, done = 0;
    int err = 0, prev_err = 0, c = 0, cm = 0, start = 0, nseg = 0, last_end = 0;
    int nmax = n;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, __shfl_xor(nmax, d));
    for (int wi = 0; wi * 64 < nmax; wi++) {
        const bool mine = !done && wi * 64 < n;
        const uint64_t word = mine ? maskT[(int64_t)wi * mask_rows + r] : 0ull;
        if (__all(!mine || (!prev && word == 0ull))) {
            // no lane can open or continue a segment in these 64 samples: only the "adapter found" test
            // (:152) can fire, and it fires somewhere in the word iff it fires at its last sample
            const int last = min(wi * 64 + 63, n - 1);
            if (mine && nseg > 0 && last - last_end > p.seg_dist) done = 1;
            if (__all(done || (wi + 1) * 64 >= n)) break;
            continue;
        }
        for (int h = 0; h < 2; h++) {
        const unsigned bits = h ? (unsigned)(word >> 32) : (unsigned)word;
#pragma unroll 8
        for (int b = 0; b < 32; b++) {
            const int i = wi * 64 + h * 32 + b;
            const int valid = (int)(i < n) & (done ^ 1);
            const int inb = valid & (int)((bits >> b) & 1u);                    // a < top (:114)
            const int opening = inb & (prev ^ 1);
            start = opening ? i : start;
            err = opening ? 0 : err;                                            // :117 re-arms the budget
            const int tol = valid & (inb ^ 1) & prev & (int)(err < p.error);    // :129
            const int act = inb | tol;
            const int closing = valid & prev & (act ^ 1);
            if (closing && c >= p.window) {                                     // :137 close
                const int end = i - prev_err;
                if (nseg > 0 && start - last_end < p.seg_dist) {
                    if (nseg <= max_segs) my[2 * (nseg - 1) + 1] = end;
                } else {
                    if (nseg < max_segs) { my[2 * nseg] = start; my[2 * nseg + 1] = end; }
                    nseg++;
                }
                last_end = end;
            }
            const int adapter = valid & (inb ^ 1) & (prev ^ 1) & (int)(nseg > 0) &
                                (int)(i - last_end > p.seg_dist);               // :152
            const int e1 = tol & (int)(i >= p.no_err_thresh);                   // :131
            c = act ? c + 1 : (closing ? 0 : c);
            cm = act ? ((cm + 1 == p.w) ? 0 : cm + 1) : (closing ? 0 : cm);
            err = closing ? 0 : err + e1;
            prev_err = (inb | closing) ? 0 : prev_err + e1;
            err -= act & (int)(cm == 0) & (int)(c >= p.window);                 // c % w == 0 (then c >= w too)
            prev = closing ? 0 : (prev | opening);
            done |= adapter;
        }
        }
        if (__all(done || (wi + 1) * 64 >= n)) break;
    }
    if (live) nsegs[r] = nseg;
}

// The same scan by RUNS, every lane at its own position (w >= 64; as k_seg_walk4 does for the segmenter, sk_segstat.hip).
// A lane takes 64 samples of its read's in-band mask at its own bit position and cuts a PIECE off it that ends where
// the rules change: at the sample whose count re-arms the budget (c = k w, known from the run's start: no division),
// at no_err_thresh (out-of-band samples before it are tolerated but not counted), at the window's or the read's end.
// Inside a piece the out-of-band samples are all free or all counted, so the run either closes at the (budget + 1)-th
// of them -- bit scans -- or swallows the piece whole.  Idle, a lane looks for the next in-band sample and for
// "adapter found" (:152: an out-of-band sample more than seg_dist behind the last segment) in one step.  One trip per
// piece instead of 25 instructions per sample: the scan of the bench's 20 000 dRNA-shaped reads goes from 4.7 to
// ~0.9 ms (it is the latency of one wavefront's chain of dependent loads now).
__global__ __launch_bounds__(64)
void k_drna_walk_runs(const uint64_t *__restrict__ maskT, int64_t mask_rows,
                      const sk_prep *__restrict__ prep, int nreads, DrnaWalk p, int kw0,
                      int32_t *__restrict__ segs, int32_t *__restrict__ nsegs, int max_segs)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < nreads;
    const int n = live ? prep[r].n : 0;
    const uint64_t *mcol = maskT + (live ? r : 0);
    int32_t *my = segs + (int64_t)(live ? r : 0) * 2 * max_segs;
    int prev = 0, err = 0, prev_err = 0, start = 0, nseg = 0, last_end = 0, refill = 0;
    int pos = 0;
    bool done = false;
    while (true) {
        const bool act = !done && pos < n;
        if (__builtin_amdgcn_ballot_w64(act) == 0ull) break;
        if (!act) continue;
        const int wi = pos >> 6, sh = pos & 63;
        const uint64_t w0 = mcol[(int64_t)wi * mask_rows];
        const uint64_t w1 = ((wi + 1) * 64 < n) ? mcol[(int64_t)(wi + 1) * mask_rows] : 0ull;
        uint64_t W = (w0 >> sh) | ((w1 << 1) << (63 - sh));
        int V = min(64, n - pos);
        if (!prev) {
            const int o = W ? min((int)__builtin_ctzll(W), V) : V;           // out-of-band samples before the next opening
            if (nseg > 0 && o > 0 && pos + o - 1 - last_end > p.seg_dist) { done = true; continue; }   // :152
            if (o == V) { pos += V; continue; }
            pos += o; W >>= o; V -= o;
            prev = 1; start = pos; err = 0; prev_err = 0;                    // :116-118
            refill = start + kw0 - 1;                                        // the sample that makes c the first multiple of w >= window
        }
        // the piece: up to the re-arming sample, up to no_err_thresh, at most the window
        int L = min(V, refill - pos + 1);
        const bool counted = pos >= p.no_err_thresh;
        if (!counted) L = min(L, p.no_err_thresh - pos);
        const uint64_t pm = (L == 64) ? ~0ull : ((1ull << L) - 1ull);
        const uint64_t Z = ~W & pm, O = W & pm;
        const int nz = __builtin_popcountll(Z);
        const int tol = counted ? max(p.error - err, 0) : (err < p.error ? 64 : 0);   // out-of-band samples the run still takes
        if (nz > tol) {                                                      // the (tol + 1)-th closes the run (:137)
            uint64_t Zk = Z;
            for (int i = 0; i < tol; i++) Zk &= Zk - 1ull;
            const int q = __builtin_ctzll(Zk);
            const uint64_t before = O & ((1ull << q) - 1ull);
            if (before) prev_err = counted ? tol - __builtin_popcountll(Z & ((2ull << (63 - __builtin_clzll(before))) - 1ull)) : 0;
            else prev_err += counted ? tol : 0;
            const int i = pos + q;
            if (i - start >= p.window) {
                const int end = i - prev_err;
                if (nseg > 0 && start - last_end < p.seg_dist) {
                    if (nseg <= max_segs) my[2 * (nseg - 1) + 1] = end;
                } else {
                    if (nseg < max_segs) { my[2 * nseg] = start; my[2 * nseg + 1] = end; }
                    nseg++;
                }
                last_end = end;
            }
            prev = 0; err = 0; prev_err = 0;
            pos = i + 1;
        } else {                                                             // the run takes the whole piece
            if (O) prev_err = counted ? __builtin_popcountll((Z >> (63 - __builtin_clzll(O))) >> 1) : 0;
            else prev_err += counted ? nz : 0;
            err += counted ? nz : 0;
            pos += L;
            if (pos - 1 == refill) { err -= 1; refill += p.w; }              // :121 / :133: c is a multiple of w (and >= window)
        }
    }
    if (live) nsegs[r] = nseg;
}

// dRNA_segmenter.py --signal branch, the scan over the rolling mean (:296-326): runs of t < bot
// (mask `below`), closed by the first t > bot (mask `above`; NaN or t == bot change nothing), merged into
// the previous segment when they start less than seg_dist after its end, and the first segment whose
// length lies in [lo_thresh, hi_thresh] is reported, both ends shifted.  One lane per read; a segment
// can only be judged once the next one has been appended (or the read ends), because merges extend it.
struct RollWalk { int seg_dist, hi_thresh, lo_thresh, shift; };

template <bool BY_RUNS>
__global__ __launch_bounds__(64)
void k_roll_walk(const uint64_t *__restrict__ below, const uint64_t *__restrict__ above, int64_t mask_rows, int64_t read_stride,
                 const sk_prep *__restrict__ prep, int nreads, RollWalk p,
                 int32_t *__restrict__ xy, int32_t *__restrict__ found)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nreads) return;
    const int n = prep[r].n;
    bool begin = false, done = false;
    int start = 0, end = 0, nseg = 0, sa = 0, sb = 0, fx = 0, fy = 0;
    auto judge = [&]() {                                      // :318-326 on the segment (sa, sb)
        const int len = sb - sa;
        if (!done && len <= p.hi_thresh && len >= p.lo_thresh) { fx = sa - p.shift; fy = sb - p.shift; done = true; }
    };
    // (the next pair of words is requested before this one is looked at: a lane's loop is a chain of dependent loads otherwise)
    const uint64_t *bp = below + (int64_t)r * read_stride, *ap = above + (int64_t)r * read_stride;
    // (read-major masks are interleaved, {below, above} side by side: one 16-byte load per step)
    const bool pairs = mask_rows == 2 && above == below + 1;
    auto fetch = [&](int wi, uint64_t &B, uint64_t &A) {
        if (pairs) { const ulonglong2 v = *(const ulonglong2 *)(bp + 2 * (int64_t)wi); B = v.x; A = v.y; }
        else { B = bp[(int64_t)wi * mask_rows]; A = ap[(int64_t)wi * mask_rows]; }
    };
    uint64_t Bn = 0ull, An = 0ull;
    if (n > 0) fetch(0, Bn, An);
    for (int wi = 0; wi * 64 < n; wi++) {
        const uint64_t B = Bn, A = An;
        if ((wi + 1) * 64 < n) fetch(wi + 1, Bn, An);
        if (!begin && B == 0ull) continue;                    // nothing opens in this word
        const int lim = min(64, n - wi * 64);
        if (BY_RUNS) {
            // by transitions instead of by samples (round 4): a word without one costs a dozen instructions -- the
            // rolling mean crosses `bot` a few times per read, the words are 270 per read
            const uint64_t valid = (lim == 64) ? ~0ull : ((1ull << lim) - 1ull);
            const uint64_t Bm = B & valid, Am = A & ~B & valid;
            int b = 0;
            while (b < 64) {
                const uint64_t from = ~0ull << b;
                if (!begin) {
                    const uint64_t t = Bm & from;
                    if (!t) break;
                    const int s1 = __builtin_ctzll(t);
                    start = wi * 64 + s1; begin = true;                      // :297-299
                    b = s1 + 1;
                } else {
                    const uint64_t t = Am & from;
                    const uint64_t upto = t ? ((1ull << __builtin_ctzll(t)) - 1ull) : ~0ull;
                    const uint64_t mid = Bm & from & upto;                   // `below` samples of the open run: end = the last (:300-301)
                    if (mid) end = wi * 64 + 63 - __builtin_clzll(mid);
                    if (!t) break;
                    if (nseg > 0 && start - sb < p.seg_dist) sb = end;       // :302-309
                    else {
                        if (nseg > 0) judge();
                        sa = start; sb = end; nseg++;
                    }
                    start = 0; end = 0; begin = false;
                    b = __builtin_ctzll(t) + 1;
                }
            }
            continue;
        }
        for (int b = 0; b < lim; b++) {
            const int i = wi * 64 + b;
            if ((B >> b) & 1) {
                if (!begin) { start = i; begin = true; }      // :297-299
                else end = i;                                 // :300-301
            } else if (begin && ((A >> b) & 1)) {             // :302-309
                if (nseg > 0 && start - sb < p.seg_dist) sb = end;
                else {
                    if (nseg > 0) judge();                    // the previous segment is final now
                    sa = start; sb = end; nseg++;
                }
                start = 0; end = 0; begin = false;
            }
        }
    }
    if (nseg > 0) judge();
    found[r] = done ? 1 : 0;
    xy[2 * r] = fx; xy[2 * r + 1] = fy;
}

} // namespace

int sk_launch_roll_walk(sk_ctx *c, const uint64_t *d_below, const uint64_t *d_above, const sk_prep *d_prep,
                        int32_t nreads, const sk_roll_params *p, int32_t *d_xy, int32_t *d_found, int64_t row_words)
{
    // row_words: 0 -- the masks are word-major ([word][read]); otherwise read-major rows of that many words with the two
    // masks interleaved, d_above = d_below + 1 (k_roll_stream)
    if (nreads <= 0) return SK_OK;
    const int64_t ws = row_words ? 2 : (int64_t)nreads, rs = row_words ? 2 * row_words : 1;
    RollWalk wp;
    wp.seg_dist = p->seg_dist; wp.hi_thresh = p->hi_thresh; wp.lo_thresh = p->lo_thresh; wp.shift = p->shift;
    const int grid = (nreads + 63) / 64;
    SK_HIP(hipEventRecord(c->ev[2], c->stream));
    if (sk_tune("SK_DRNA_STEP") == nullptr)
        hipLaunchKernelGGL(k_roll_walk<true>, dim3(grid), dim3(64), 0, c->stream, d_below, d_above, ws, rs, d_prep,
                           nreads, wp, d_xy, d_found);
    else
        hipLaunchKernelGGL(k_roll_walk<false>, dim3(grid), dim3(64), 0, c->stream, d_below, d_above, ws, rs, d_prep,
                           nreads, wp, d_xy, d_found);
    SK_HIP(hipGetLastError());
    SK_HIP(hipEventRecord(c->ev[3], c->stream));
    return SK_OK;
}

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
    if (wp.w >= 64 && wp.w <= (1 << 24) && wp.window >= 0 && wp.window <= (1 << 24) && sk_tune("SK_DRNA_STEP") == nullptr) {
        // the first count at which the budget is re-armed: the smallest multiple of w that is >= max(window, w)
        const int kw0 = ((wp.window > wp.w ? wp.window : wp.w) + wp.w - 1) / wp.w * wp.w;
        hipLaunchKernelGGL(k_drna_walk_runs, dim3(grid), dim3(64), 0, c->stream, d_mask, mask_rows, d_prep, nreads, wp, kw0,
                           d_segs, d_nsegs, max_segs);
    } else
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
    // the straight-line variant needs a dead corrector test and positive report thresholds
    const bool fast = wp.error < wp.corrector && wp.window >= 1 && wp.first_len >= 1 &&
                      sk_tune("SK_WALK_GENERAL") == nullptr;
    if (fast)
        hipLaunchKernelGGL(k_segment_walk<true>, dim3(grid), dim3(64), 0, c->stream, d_mask, mask_rows, d_prep,
                           nreads, wp, d_segs, d_nsegs, max_segs);
    else
        hipLaunchKernelGGL(k_segment_walk<false>, dim3(grid), dim3(64), 0, c->stream, d_mask, mask_rows, d_prep,
                           nreads, wp, d_segs, d_nsegs, max_segs);
    SK_HIP(hipGetLastError());
    SK_HIP(hipEventRecord(c->ev[3], c->stream));
    return SK_OK;
}
