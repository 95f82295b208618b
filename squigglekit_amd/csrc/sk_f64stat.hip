// sk_f64stat.hip -- filter + statistics of FLOAT64 reads as one streaming pass (gfx950).
//
// The float64 twin of sk_segstat.hip (segmenter) and sk_prepw.hip (MotifSeq medmad) for the reads the reference
// parses as floats: pA TSVs (segmenter.py:198-201; SquigglePull.py:183-189 writes np.round(pA, 2)) and every
// MotifSeq --signal read (MotifSeq.py:270).  Per read:
//   scale_outliers      segmenter.py:311-318 / MotifSeq.py:317-324    strict lo < x < hi
//   np.median           segmenter.py:410 / MotifSeq.py:194            exact (a sample, or the mean of two)
//   segmenter:  np.std -> top / bot (:412-414), `a < top and a > bot` (:431) as two bit masks in RAW coordinates
//               ({in band, kept}, the layout k_seg_walk3 of sk_segstat.hip walks)
//   MotifSeq:   MAD = median(|x - med|) (:195-196), the filtered samples in order for the DTW feed
//
// One wavefront owns one read of up to 64 NJ samples from its first load to its last store: lane l keeps samples
// 64 j + l (j < NJ) in registers -- 8-byte loads, 512 contiguous bytes per instruction, rows need no alignment
// beyond the doubles' own -- so every v_cmp over slot j yields, as its lane mask, the mask word of samples
// 64 j .. 64 j + 63.  No workgroup barrier, no second look at memory.
//
// Median of arbitrary doubles without sorting: a 2048-bin LDS histogram over [min, max] of the kept samples
// (bin = (x - min) * scale, monotone in x), a rank select on it, then the members of the selected bin (a handful:
// for pA data on the 0.01 grid one bin holds at most one distinct value) are gathered and ranked exactly.  Reads
// whose selected bin holds more than 64 members that are not all equal go to the retry list.
//
// Segmenter thresholds WITHOUT numpy's summation order (the float64 analogue of sk_segstat.hip's certificate):
// the state machine only sees the comparisons x < top, x > bot.  std is computed from shifted sums in any order
// with a rigorous error bound; numpy's top / bot lie within delta of ours; a sample is classified from
// u = |x - median| as "in band" when u < spread - delta and "out" when u > spread + delta.  If some kept sample
// falls between the two the read is UNCERTIFIED and goes to the retry list, which the numpy-order kernel
// (k_prep_f64<LISTED>, sk_prep.hip) redoes, rewriting the read's masks in place.  For pA reads that happens about
// once per 10^6 reads; all-equal reads (std = 0, where numpy's own rounding decides) always take it.
//
// Algorithmic HBM traffic per read: 8 M in; segmenter: M / 4 out (masks) + 52 B; MotifSeq: 8 n out (the filtered
// samples for the DTW kernels) + 48 B.
#include "sk_common.h"
#include <math.h>
#include <stdlib.h>

// v_writelane_b32: this clang has no builtin for it; the LLVM intrinsic is reachable through an asm label
extern "C" __device__ int sk_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

namespace {

constexpr int WPB = 4;                  // wavefronts (= reads in flight) per workgroup
constexpr int NB = 2048;                // histogram bins per round
constexpr int PER = NB / 64;            // bins a lane owns in the first sweep of the rank select
constexpr int CAP = 64;                 // members of the selected bin that are ranked exactly (one per lane)
constexpr int MODE_SEG = 0, MODE_MEDMAD = 1;
constexpr double U53 = 1.1102230246251565e-16;   // 2^-53

struct F64StatArgs {
    const double  *sig;
    const int64_t *off;                 // zero based, nreads + 1
    const int32_t *len;                 // optional: read r is its first len[r] samples (the caller's sig[:Num] cut)
    int            nreads;
    double         lo, hi;              // outlier limits (strict)
    double         std_scale;           // segmenter
    double         delta_scale;         // 1.0; tests raise it to push reads onto the retry list
    sk_prep       *prep;
    uint4         *mask2;               // segmenter: [nreads][row16] of {in band lo, hi, kept lo, hi}
    int            row16;
    int32_t       *len_out;             // segmenter: the read's length (the walk's `len`)
    int32_t       *retry;               // [0] = count, [1 ..] = reads the numpy-order kernel has to redo
    double        *comp;                // MotifSeq: filtered samples of read r at comp + off[r]
};

__device__ __forceinline__ double vmin64(double a, double b)    // one v_min_f64 (operands are never NaN here)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmax64(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2, 3
    return v;
}
// inclusive scan inside each half of 32 lanes
__device__ __forceinline__ int half_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
    return v;
}
__device__ __forceinline__ double wave_min64(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = vmin64(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ double wave_max64(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = vmax64(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ double wave_sum64(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ double readlane64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// Bin, count and exclusive prefix of the bin that holds rank k (0 <= k < sum of the counts); b < 0: inconsistent.
struct RankSel { int b, pre, c; };

__device__ __forceinline__ void rank_select2(const unsigned *hist, int lane, int k1, int k2, RankSel &r1, RankSel &r2)
{
    const int hb0 = lane * PER;
    int local = 0;
#pragma unroll
    for (int q = 0; q < PER / 4; q += 2) {               // two 16-byte reads in flight (all eight: 32 registers)
        const uint4 v = *(const uint4 *)(hist + hb0 + 4 * q);
        const uint4 u = *(const uint4 *)(hist + hb0 + 4 * q + 4);
        local += (int)(v.x + v.y + v.z + v.w) + (int)(u.x + u.y + u.z + u.w);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int inc = wave_incl_scan(local), pre = inc - local;
    // the lane whose 32 bins hold the rank is found from the scan; then ALL lanes look at that lane's bins, one
    // bin per lane (both halves of the wave do the same work), instead of every lane scanning its own 32
    auto one = [&](int k) -> RankSel {
        const unsigned long long own = __ballot(k >= pre && k < inc);
        const int ol = own ? (int)__builtin_ctzll(own) : 0;
        const int opre = __builtin_amdgcn_readlane(pre, ol);
        const int c = (int)hist[ol * PER + (lane & (PER - 1))];
        const int acc = opre + half_incl_scan(c);
        const unsigned hit = (unsigned)__ballot(acc > k);           // (low half)
        const int bi = hit ? (int)__builtin_ctz(hit) : 0;
        RankSel rs;
        rs.b = (own != 0ull && hit != 0u) ? ol * PER + bi : -1;
        rs.c = __builtin_amdgcn_readlane(c, bi);
        rs.pre = __builtin_amdgcn_readlane(acc - c, bi);
        return rs;
    };
    r1 = one(k1);
    r2 = r1;
    if (k2 != k1) r2 = one(k2);
}

template <int NJ, int MODE, int OCC>
__global__ __launch_bounds__(64 * WPB, OCC)
void k_f64_stats(const F64StatArgs a)
{
    __shared__ __align__(16) unsigned hist_all[WPB][NB + 64];       // [NB]: dump bin (dropped samples)
    __shared__ __align__(16) double list_all[WPB][CAP];
    __shared__ unsigned cnt_all[WPB][2];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned *hist = hist_all[w];
    double *list = list_all[w];
    unsigned *cnt = cnt_all[w];
    const double INF = __builtin_huge_val();

    auto clear_hist = [&]() {
#pragma unroll
        for (int q = 0; q < PER / 4; q++) *(uint4 *)(hist + lane * PER + 4 * q) = make_uint4(0u, 0u, 0u, 0u);
        hist[NB + lane] = 0u;
    };
    clear_hist();

    const int nwaves = gridDim.x * WPB;
    for (int r = blockIdx.x * WPB + w; r < a.nreads; r += nwaves) {
        const int64_t o0 = a.off[r];
        int64_t Mfull = max(a.off[r + 1] - o0, (int64_t)0);
        if (a.len) Mfull = min(Mfull, (int64_t)max(a.len[r], 0));
        const int M = __builtin_amdgcn_readfirstlane((int)min(Mfull, (int64_t)(64 * NJ)));
        const double *row = a.sig + o0;

        // ---- the read into registers: lane l holds samples 64 j + l; slots past the end read as +inf ----------
        // (the validity test is `lane < M - 64 j` with a scalar right-hand side: written as `64 j + lane < M` the
        // compiler keeps all NJ lane indices in vector registers across the whole kernel)
        double x[NJ];
        const double *prow = row + lane;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            int rem = M - 64 * j;
            asm("" : "+s"(rem));                        // (opaque: else it is folded back into 64 j + lane < M)
            x[j] = (lane < rem) ? prow[64 * j] : INF;
        }

        // ---- first look: filter (lane mask of the compare = the "kept" word), extremes, shifted sums -----------
        // dropped samples are overwritten with +inf: every later pass sees them fall out by themselves
        unsigned kplo = 0u, kphi = 0u;                  // lane j: the kept word of slot j
        unsigned long long anydrop = 0ull;              // bit j: slot j holds dropped samples (or the read's end)
        int n = 0;
        double mn = INF, mx = -INF, S1 = 0.0, S2 = 0.0;
        double K = readlane64(x[0], 0);                 // shift of the sums (any finite value near the data)
        if (!(K > a.lo && K < a.hi)) K = 0.5 * (a.lo + a.hi);
        if (!(fabs(K) < 1e300)) K = 0.0;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            if (64 * j >= M) continue;                  // (wave-uniform)
            const bool k = x[j] > a.lo && x[j] < a.hi;
            const unsigned long long km = __ballot(k);
            kplo = (unsigned)sk_writelane_i32((int)(unsigned)km, j, (int)kplo);
            kphi = (unsigned)sk_writelane_i32((int)(unsigned)(km >> 32), j, (int)kphi);
            n += __popcll(km);
            if (km != ~0ull) { anydrop |= 1ull << j; x[j] = k ? x[j] : INF; }
            if (k) {
                mn = vmin64(mn, x[j]);
                mx = vmax64(mx, x[j]);
                if (MODE == MODE_SEG) {
                    const double d = x[j] - K;
                    S1 += d;
                    S2 = fma(d, d, S2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);          // one slot after the other: interleaved, the slots' temporaries add up
        }
        // (wave-uniform from here on: into scalar registers, the vector ones are needed for the samples)
        mn = readlane64(wave_min64(mn), 0);
        mx = readlane64(wave_max64(mx), 0);
        if (MODE == MODE_SEG) {
            S1 = readlane64(wave_sum64(S1), 0);
            S2 = readlane64(wave_sum64(S2), 0);
        }

        sk_prep pr;
        pr.n = n; pr.flags = 0; pr.center = 0.0; pr.scale = 1.0; pr.top = 0.0; pr.bot = 0.0;
        bool ok = true;
        unsigned inlo = 0u, inhi = 0u;

        // exact order statistics k1 <= k2 <= k1 + 1 of val(j) over the kept samples, val in [vlo, vhi], vlo < vhi:
        // histogram over [vlo, vhi], rank select, members of the selected bin ranked exactly
        auto select2 = [&](auto val, double vlo, double vhi, int k1, int k2, double &v1, double &v2) -> bool {
            const double sc = ((double)NB - 0.5) / (vhi - vlo);
            if (!(sc > 0.0 && sc < 1e300)) return false;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                if (64 * j >= M) continue;
                unsigned b = (unsigned)((val(j) - vlo) * sc);               // v_cvt_u32_f64 saturates: +inf -> 2^32 - 1
                if ((anydrop >> j) & 1ull) b = min(b, (unsigned)NB);        // dropped -> the dump bin
                atomicAdd(&hist[b], 1u);
                __builtin_amdgcn_sched_barrier(0);
            }
            RankSel r1, r2;
            rank_select2(hist, lane, k1, k2, r1, r2);
            clear_hist();
            if (lane == 0) cnt[0] = 0u;
            if (r1.b < 0 || r2.b < 0) return false;
            const bool two = r2.b != r1.b;
            double mnm = INF, mxm = -INF, mn2 = INF;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                if (64 * j >= M) continue;
                const double v = val(j);
                const unsigned b = (unsigned)((v - vlo) * sc);
                if (b == (unsigned)r1.b) {
                    const unsigned slot = atomicAdd(&cnt[0], 1u);
                    if (slot < (unsigned)CAP) list[slot] = v;
                    mnm = vmin64(mnm, v);
                    mxm = vmax64(mxm, v);
                }
                if (two && b == (unsigned)r2.b) mn2 = vmin64(mn2, v);
                __builtin_amdgcn_sched_barrier(0);
            }
            mnm = readlane64(wave_min64(mnm), 0);
            mxm = readlane64(wave_max64(mxm), 0);
            if (two) mn2 = readlane64(wave_min64(mn2), 0);
            const int j1 = k1 - r1.pre;                 // rank of k1 inside its bin
            if (mnm == mxm) {                           // one distinct value in the bin (the usual case on gridded data)
                v1 = mnm;
                v2 = two ? mn2 : mnm;
                return true;
            }
            if (r1.c > CAP) return false;
            // exact ranks among the <= 64 members: lane l takes member l and counts the members ordered before it
            const double m = (lane < r1.c) ? list[lane] : INF;
            int rk = 0;
            for (int k = 0; k < r1.c; k++) {
                const double mk = list[k];
                rk += (mk < m || (mk == m && k < lane)) ? 1 : 0;
            }
            const unsigned long long h1 = __ballot(lane < r1.c && rk == j1);
            if (h1 == 0ull) return false;
            v1 = readlane64(m, (int)__builtin_ctzll(h1));
            if (two) v2 = mn2;                          // k2 is the first element of the next occupied bin
            else if (k2 != k1) {
                const unsigned long long h2 = __ballot(lane < r1.c && rk == j1 + 1);
                if (h2 == 0ull) return false;
                v2 = readlane64(m, (int)__builtin_ctzll(h2));
            } else v2 = v1;
            return true;
        };

        if (n == 0) {
            pr.flags = SK_FLAG_EMPTY;
            const double qnan = __builtin_nan("");
            pr.center = qnan; pr.scale = qnan; pr.top = qnan; pr.bot = qnan;
        } else {
            // ---- median ------------------------------------------------------------------------------------------
            const int k1 = (n - 1) / 2, k2 = n / 2;
            double v1 = mn, v2 = mn;
            if (mn != mx) ok = select2([&](int j) { return x[j]; }, mn, mx, k1, k2, v1, v2);
            const double median = (k1 == k2) ? v1 : (v1 + v2) / 2.0;         // np.median: mean of the two middle elements

            if (MODE == MODE_MEDMAD) {
                // ---- MAD = median(|x - med|)   MotifSeq.py:195 ---------------------------------------------------
                double w1 = 0.0, w2 = 0.0;
                const double umax = vmax64(fabs(mn - median), fabs(mx - median));
                if (ok && umax > 0.0)
                    ok = select2([&](int j) { return fabs(x[j] - median); }, 0.0, umax, k1, k2, w1, w2);
                const double mad = (k1 == k2) ? w1 : (w1 + w2) / 2.0;
                pr.center = median;
                pr.scale = mad * 1.4826;                                     // MotifSeq.py:196
                if (mad == 0.0) pr.flags |= SK_FLAG_DEGENERATE;
            } else {
                // ---- thresholds + certificate (file header) ------------------------------------------------------
                const double dn = (double)n;
                const double md = S1 / dn, Q = S2 / dn;
                const double var = Q - md * md;
                const double Ev = 8.0 * (dn + 8.0) * U53 * Q;                // |var - var_true|
                const double sd = sqrt(var);
                const double A = vmax64(fabs(mn), fabs(mx));
                const double dstd = Ev / sd + dn * U53 * A + sd * (dn + 8.0) * U53;     // |sd - numpy's std|
                const double spread = sd * a.std_scale;                      // segmenter.py:413-414
                const double dlt = a.delta_scale * 4.0 *
                                   (fabs(a.std_scale) * dstd + U53 * (4.0 * fabs(spread) + fabs(median) + 2.0 * A));
                if (!(var > 4.0 * Ev)) ok = false;                           // (also NaN; all-equal reads)
                const double s_lo = spread - dlt, s_hi = spread + dlt;
                pr.center = median; pr.scale = sd; pr.top = median + spread; pr.bot = median - spread;
                // ---- second look: in band / out of band / undecided -------------------------------------------
                unsigned long long unc = 0ull;
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    if (64 * j >= M) continue;
                    const double u = fabs(x[j] - median);                    // dropped samples: +inf, out of band
                    const unsigned long long im = __ballot(u < s_lo);
                    unc |= ~(im | __ballot(u > s_hi));
                    inlo = (unsigned)sk_writelane_i32((int)(unsigned)im, j, (int)inlo);
                    inhi = (unsigned)sk_writelane_i32((int)(unsigned)(im >> 32), j, (int)inhi);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (unc != 0ull) ok = false;
            }
        }

        if (MODE == MODE_MEDMAD) {
            // the filtered samples, in order, for the DTW feed
            double *crow = a.comp + o0;
            if (n == M) {                               // nothing dropped: the DTW feed reads the input itself
                pr.flags |= SK_IFLAG_INPLACE;
            } else {
                int base = 0;
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    if (64 * j >= M) continue;
                    const unsigned klo = (unsigned)__builtin_amdgcn_readlane((int)kplo, j);
                    const unsigned khi = (unsigned)__builtin_amdgcn_readlane((int)kphi, j);
                    const unsigned long long km = ((unsigned long long)khi << 32) | klo;
                    const int pos = base + (int)__builtin_amdgcn_mbcnt_hi(khi, __builtin_amdgcn_mbcnt_lo(klo, 0u));
                    if ((km >> lane) & 1ull) crow[pos] = x[j];
                    base += __popcll(km);
                }
            }
        } else {
            const int nent = (M + 63) >> 6;
            if (lane < nent) a.mask2[(int64_t)r * a.row16 + lane] = make_uint4(inlo, inhi, kplo, kphi);
        }
        if (lane == 0) {
            a.prep[r] = pr;
            if (MODE == MODE_SEG) a.len_out[r] = M;
            if (!ok) a.retry[1 + atomicAdd(&a.retry[0], 1)] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Reads longer than 4 096 samples (real reads are tens of thousands of samples long): the same algorithm, one wavefront
// per read, but the read no longer fits the registers -- it is looked at several times, window by window (2 048 samples),
// each look re-reading it (L2 / Infinity Cache / HBM):
//   look A   filter, n, extremes, shifted sums -- and the median's first histogram, over a provisional range taken from
//            the first window (values outside clamp into the edge bins: any monotone binning serves the rank select)
//   look B   (later rounds / the MAD) histogram over the range           look C   members of the selected bin (+ their extremes)
//            -- repeated on the members' own range while the bin holds more than 512 values that are not all equal;
//               up to 512 members are resolved in LDS (re-histogrammed there until at most 64 are left, then ranked)
//   segmenter: look D   classification, the {in band, kept} entries of each window stored as it goes
//   MotifSeq:  looks B', C' for the MAD; the filtered samples are copied (one more look) only when the filter dropped some
// ------------------------------------------------------------------------------------------------------------------
constexpr int LNJ = 32;                 // 64-sample slots per window
constexpr int LWIN = 64 * LNJ;
constexpr int CAPL = 512;               // members resolved in LDS

template <int MODE>
__global__ __launch_bounds__(64 * WPB, 3)
void k_f64_long(const F64StatArgs a)
{
    __shared__ __align__(16) unsigned hist_all[WPB][NB + 64];
    __shared__ __align__(16) double list_all[WPB][CAPL];
    __shared__ unsigned cnt_all[WPB][2];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned *hist = hist_all[w];
    double *list = list_all[w];
    unsigned *cnt = cnt_all[w];
    const double INF = __builtin_huge_val();

    auto clear_hist = [&]() {
#pragma unroll
        for (int q = 0; q < PER / 4; q++) *(uint4 *)(hist + lane * PER + 4 * q) = make_uint4(0u, 0u, 0u, 0u);
        hist[NB + lane] = 0u;
    };
    clear_hist();

    // exact ranks j1 (and j1 + 1 when two_ranks) among list[0 .. c), c <= CAPL, not all equal: re-histogram in LDS until
    // at most 64 are left, then rank them one per lane
    auto list_select = [&](int c, int j1, bool two_ranks, double &v1, double &v2) __attribute__((always_inline)) -> bool {
        for (int it = 0; it < 10; it++) {
            if (c <= 64) {
                const double m = (lane < c) ? list[lane] : INF;
                int rk = 0;
                for (int k = 0; k < c; k++) {
                    const double mk = list[k];
                    rk += (mk < m || (mk == m && k < lane)) ? 1 : 0;
                }
                const unsigned long long h1 = __ballot(lane < c && rk == j1);
                if (h1 == 0ull) return false;
                v1 = readlane64(m, (int)__builtin_ctzll(h1));
                v2 = v1;
                if (two_ranks) {
                    const unsigned long long h2 = __ballot(lane < c && rk == j1 + 1);
                    if (h2 == 0ull) return false;
                    v2 = readlane64(m, (int)__builtin_ctzll(h2));
                }
                return true;
            }
            double e[CAPL / 64];                                  // my entries: list[lane + 64 k]
            double lmn = INF, lmx = -INF;
#pragma unroll
            for (int k = 0; k < CAPL / 64; k++) {
                e[k] = (lane + 64 * k < c) ? list[lane + 64 * k] : INF;
                if (lane + 64 * k < c) { lmn = vmin64(lmn, e[k]); lmx = vmax64(lmx, e[k]); }
            }
            lmn = readlane64(wave_min64(lmn), 0);
            lmx = readlane64(wave_max64(lmx), 0);
            if (lmn == lmx) { v1 = v2 = lmn; return true; }
            const double sc = ((double)NB - 0.5) / (lmx - lmn);
            if (!(sc > 0.0 && sc < 1e300)) return false;
#pragma unroll
            for (int k = 0; k < CAPL / 64; k++)
                if (lane + 64 * k < c) atomicAdd(&hist[(unsigned)((e[k] - lmn) * sc)], 1u);
            RankSel r1, r2;
            rank_select2(hist, lane, j1, two_ranks ? j1 + 1 : j1, r1, r2);
            clear_hist();
            if (lane == 0) cnt[0] = 0u;
            if (r1.b < 0 || r2.b < 0) return false;
            const bool two = r2.b != r1.b;
            double m2 = INF;
#pragma unroll
            for (int k = 0; k < CAPL / 64; k++) {
                if (lane + 64 * k < c) {
                    const unsigned b = (unsigned)((e[k] - lmn) * sc);
                    if (b == (unsigned)r1.b) list[atomicAdd(&cnt[0], 1u)] = e[k];      // (every read of the old list is done)
                    if (two && b == (unsigned)r2.b) m2 = vmin64(m2, e[k]);
                }
            }
            if (two) {                                            // rank j1 + 1 is the first value of the next occupied bin
                m2 = readlane64(wave_min64(m2), 0);
                double dummy;
                const bool ok1 = r1.c <= c;
                c = r1.c; j1 -= r1.pre;
                if (!ok1) return false;
                // resolve rank j1 alone, then attach m2
                two_ranks = false;
                double t1 = 0.0;
                // (tail call by iteration: fall through with the narrowed list)
                bool done = false;
                for (int it2 = it + 1; it2 < 10 && !done; it2++) {
                    if (c <= 64) {
                        const double m = (lane < c) ? list[lane] : INF;
                        int rk = 0;
                        for (int k = 0; k < c; k++) {
                            const double mk = list[k];
                            rk += (mk < m || (mk == m && k < lane)) ? 1 : 0;
                        }
                        const unsigned long long h1 = __ballot(lane < c && rk == j1);
                        if (h1 == 0ull) return false;
                        t1 = readlane64(m, (int)__builtin_ctzll(h1));
                        done = true;
                    } else {
                        // still more than 64 equal-bin members: they are the largest of the bin... take the exact path again
                        double e2[CAPL / 64];
                        double qmn = INF, qmx = -INF;
#pragma unroll
                        for (int k = 0; k < CAPL / 64; k++) {
                            e2[k] = (lane + 64 * k < c) ? list[lane + 64 * k] : INF;
                            if (lane + 64 * k < c) { qmn = vmin64(qmn, e2[k]); qmx = vmax64(qmx, e2[k]); }
                        }
                        qmn = readlane64(wave_min64(qmn), 0);
                        qmx = readlane64(wave_max64(qmx), 0);
                        if (qmn == qmx) { t1 = qmn; done = true; break; }
                        const double sc2 = ((double)NB - 0.5) / (qmx - qmn);
                        if (!(sc2 > 0.0 && sc2 < 1e300)) return false;
#pragma unroll
                        for (int k = 0; k < CAPL / 64; k++)
                            if (lane + 64 * k < c) atomicAdd(&hist[(unsigned)((e2[k] - qmn) * sc2)], 1u);
                        RankSel q1, q2;
                        rank_select2(hist, lane, j1, j1, q1, q2);
                        clear_hist();
                        if (lane == 0) cnt[0] = 0u;
                        if (q1.b < 0) return false;
#pragma unroll
                        for (int k = 0; k < CAPL / 64; k++)
                            if (lane + 64 * k < c && (unsigned)((e2[k] - qmn) * sc2) == (unsigned)q1.b)
                                list[atomicAdd(&cnt[0], 1u)] = e2[k];
                        c = q1.c; j1 -= q1.pre;
                    }
                }
                (void)dummy;
                if (!done) return false;
                v1 = t1; v2 = m2;
                return true;
            }
            c = r1.c; j1 -= r1.pre;
        }
        return false;
    };

    const int nwaves = gridDim.x * WPB;
    for (int r = blockIdx.x * WPB + w; r < a.nreads; r += nwaves) {
        const int64_t o0 = a.off[r];
        int64_t Mfull = max(a.off[r + 1] - o0, (int64_t)0);
        if (a.len) Mfull = min(Mfull, (int64_t)max(a.len[r], 0));
        const int M = __builtin_amdgcn_readfirstlane((int)min(Mfull, (int64_t)0x3fffffff));
        const double *row = a.sig + o0;
        const int nwin = (M + LWIN - 1) / LWIN;

        double x[LNJ];
        unsigned kplo = 0u, kphi = 0u;                  // lane j: the kept word of slot j of the CURRENT window
        int Mw = 0, nkw = 0;                            // samples / kept samples of the current window
        // one window into registers, filtered: dropped samples and slots past the end become +inf
        auto load_window = [&](int wi) __attribute__((always_inline)) {
            Mw = __builtin_amdgcn_readfirstlane(min(M - wi * LWIN, LWIN));
            const double *prow = row + (int64_t)wi * LWIN + lane;
#pragma unroll
            for (int j = 0; j < LNJ; j++) {
                const int rem = __builtin_amdgcn_readfirstlane(Mw - 64 * j);   // (scalar and opaque: see k_f64_stats)
                x[j] = (lane < rem) ? prow[64 * j] : INF;
            }
            nkw = 0;
#pragma unroll
            for (int j = 0; j < LNJ; j++) {
                if (64 * j >= Mw) {                     // (wave-uniform)
                    kplo = (unsigned)sk_writelane_i32(0, j, (int)kplo);
                    kphi = (unsigned)sk_writelane_i32(0, j, (int)kphi);
                    continue;
                }
                const bool k = x[j] > a.lo && x[j] < a.hi;
                const unsigned long long km = __ballot(k);
                kplo = (unsigned)sk_writelane_i32((int)(unsigned)km, j, (int)kplo);
                kphi = (unsigned)sk_writelane_i32((int)(unsigned)(km >> 32), j, (int)kphi);
                nkw += __popcll(km);
                if (km != ~0ull) x[j] = k ? x[j] : INF;
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        // ---- look A ---------------------------------------------------------------------------------------------
        int n = 0;
        double mn = INF, mx = -INF, S1 = 0.0, S2 = 0.0;
        double K = (M > 0) ? row[0] : 0.0;              // shift of the sums (any finite value near the data)
        if (!(K > a.lo && K < a.hi)) K = 0.5 * (a.lo + a.hi);
        if (!(fabs(K) < 1e300)) K = 0.0;
        // The median's first histogram rides in this look, over a PROVISIONAL range -- the first window's extremes, widened
        // by a quarter -- with everything outside clamped into the edge bins: any monotone binning serves the rank select,
        // a poor range only costs more members in the selected bin (resolved by the later rounds).  Saves one look.
        bool pre = false;
        double plo = 0.0, psc = 0.0;
        for (int wi = 0; wi < nwin; wi++) {
            load_window(wi);
            n += nkw;
            if (wi == 0) {
                double m0 = INF, m1 = -INF;
#pragma unroll
                for (int j = 0; j < LNJ; j++)
                    if (64 * j < Mw && x[j] != INF) { m0 = vmin64(m0, x[j]); m1 = vmax64(m1, x[j]); }
                m0 = readlane64(wave_min64(m0), 0);
                m1 = readlane64(wave_max64(m1), 0);
                const double wd = 0.25 * (m1 - m0);
                plo = m0 - wd;
                psc = ((double)NB - 0.5) / ((m1 + wd) - plo);
                pre = (m1 > m0) && (psc > 0.0) && (psc < 1e300) && (fabs(plo) < 1e300);
            }
#pragma unroll
            for (int j = 0; j < LNJ; j++) {
                if (64 * j >= Mw) continue;
                if (x[j] != INF) {
                    mn = vmin64(mn, x[j]);
                    mx = vmax64(mx, x[j]);
                    if (MODE == MODE_SEG) {
                        const double d = x[j] - K;
                        S1 += d;
                        S2 = fma(d, d, S2);
                    }
                }
                if (pre) {                              // (wave-uniform)
                    const double t = vmin64(vmax64((x[j] - plo) * psc, 0.0), (double)(NB - 1));
                    atomicAdd(&hist[(x[j] != INF) ? (unsigned)t : (unsigned)NB], 1u);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        mn = readlane64(wave_min64(mn), 0);
        mx = readlane64(wave_max64(mx), 0);
        if (MODE == MODE_SEG) {
            S1 = readlane64(wave_sum64(S1), 0);
            S2 = readlane64(wave_sum64(S2), 0);
        }

        sk_prep pr;
        pr.n = n; pr.flags = 0; pr.center = 0.0; pr.scale = 1.0; pr.top = 0.0; pr.bot = 0.0;
        bool ok = true;

        // exact order statistics k1 <= k2 <= k1 + 1 of val(x) over the kept samples, val in [vlo, vhi], vlo < vhi
        // have_hist: the histogram of round 0 is already in LDS (look A built it: bins clamped, range plo / psc)
        auto select2 = [&](auto val, double vlo, double vhi, int k1, int k2, double &v1, double &v2, bool have_hist)
                       __attribute__((always_inline)) -> bool {
            bool restricted = false;                    // later rounds: only values inside [vlo, vhi] take part
            bool have2 = false;                         // v2 already known (k2 fell into the next occupied bin)
            for (int round = 0; round < 6; round++) {
                const bool clamped = have_hist && round == 0;
                double sc = ((double)NB - 0.5) / (vhi - vlo);
                if (clamped) { vlo = plo; sc = psc; }
                if (!(sc > 0.0 && sc < 1e300)) return false;
                auto bin_of = [&](double xv, double v) -> unsigned {
                    bool in = xv != INF;
                    if (restricted) in = in && v >= vlo && v <= vhi;
                    double t = (v - vlo) * sc;
                    if (clamped) t = vmin64(vmax64(t, 0.0), (double)(NB - 1));
                    return in ? (unsigned)t : (unsigned)NB;
                };
                if (!clamped)
                for (int wi = 0; wi < nwin; wi++) {     // ---- look B
                    load_window(wi);
#pragma unroll
                    for (int j = 0; j < LNJ; j++) {
                        if (64 * j >= Mw) continue;
                        atomicAdd(&hist[bin_of(x[j], val(x[j]))], 1u);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                RankSel r1, r2;
                rank_select2(hist, lane, k1, have2 ? k1 : k2, r1, r2);
                clear_hist();
                if (lane == 0) cnt[0] = 0u;
                if (r1.b < 0 || r2.b < 0) return false;
                const bool two = !have2 && r2.b != r1.b;
                double mnm = INF, mxm = -INF, mn2 = INF;
                for (int wi = 0; wi < nwin; wi++) {     // ---- look C
                    load_window(wi);
#pragma unroll
                    for (int j = 0; j < LNJ; j++) {
                        if (64 * j >= Mw) continue;
                        const double v = val(x[j]);
                        const unsigned b = bin_of(x[j], v);
                        if (b == (unsigned)r1.b) {
                            const unsigned slot = atomicAdd(&cnt[0], 1u);
                            if (slot < (unsigned)CAPL) list[slot] = v;
                            mnm = vmin64(mnm, v);
                            mxm = vmax64(mxm, v);
                        }
                        if (two && b == (unsigned)r2.b) mn2 = vmin64(mn2, v);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                mnm = readlane64(wave_min64(mnm), 0);
                mxm = readlane64(wave_max64(mxm), 0);
                if (two) { v2 = readlane64(wave_min64(mn2), 0); have2 = true; }
                const int j1 = k1 - r1.pre;
                const bool both_here = !have2 && k2 != k1;      // k2 = k1 + 1 lies in the same bin
                if (mnm == mxm) { v1 = mnm; if (!have2) v2 = mnm; return true; }
                if (r1.c <= CAPL) {
                    double t1 = 0.0, t2 = 0.0;
                    if (!list_select(r1.c, j1, both_here, t1, t2)) return false;
                    v1 = t1;
                    if (!have2) v2 = t2;
                    return true;
                }
                // too many distinct members: again, on the members' own range
                vlo = mnm; vhi = mxm; restricted = true;
                k1 = j1; k2 = both_here ? j1 + 1 : j1;
            }
            return false;
        };

        double median = 0.0, s_lo = 0.0, s_hi = 0.0;
        if (n == 0) {
            pr.flags = SK_FLAG_EMPTY;
            const double qnan = __builtin_nan("");
            pr.center = qnan; pr.scale = qnan; pr.top = qnan; pr.bot = qnan;
            s_lo = -1.0; s_hi = -1.0;                   // nothing is in band
        } else {
            const int k1 = (n - 1) / 2, k2 = n / 2;
            double v1 = mn, v2 = mn;
            if (mn != mx) ok = select2([&](double v) { return v; }, mn, mx, k1, k2, v1, v2, pre);
            else if (pre) clear_hist();                  // (unused provisional histogram)
            median = (k1 == k2) ? v1 : (v1 + v2) / 2.0;
            if (MODE == MODE_MEDMAD) {
                double w1 = 0.0, w2 = 0.0;
                const double umax = vmax64(fabs(mn - median), fabs(mx - median));
                if (ok && umax > 0.0)
                    ok = select2([&](double v) { return fabs(v - median); }, 0.0, umax, k1, k2, w1, w2, false);
                const double mad = (k1 == k2) ? w1 : (w1 + w2) / 2.0;
                pr.center = median;
                pr.scale = mad * 1.4826;
                if (mad == 0.0) pr.flags |= SK_FLAG_DEGENERATE;
            } else {
                const double dn = (double)n;
                const double md = S1 / dn, Q = S2 / dn;
                const double var = Q - md * md;
                const double Ev = 8.0 * (dn + 8.0) * U53 * Q;
                const double sd = sqrt(var);
                const double A = vmax64(fabs(mn), fabs(mx));
                const double dstd = Ev / sd + dn * U53 * A + sd * (dn + 8.0) * U53;
                const double spread = sd * a.std_scale;
                const double dlt = a.delta_scale * 4.0 *
                                   (fabs(a.std_scale) * dstd + U53 * (4.0 * fabs(spread) + fabs(median) + 2.0 * A));
                if (!(var > 4.0 * Ev)) ok = false;
                s_lo = spread - dlt; s_hi = spread + dlt;
                pr.center = median; pr.scale = sd; pr.top = median + spread; pr.bot = median - spread;
            }
        }

        if (MODE == MODE_MEDMAD) {
            if (n == M) {
                pr.flags |= SK_IFLAG_INPLACE;           // nothing dropped: the DTW feed reads the input itself
            } else {
                double *crow = a.comp + o0;             // the filtered samples, in order
                int base = 0;
                for (int wi = 0; wi < nwin; wi++) {
                    load_window(wi);
#pragma unroll
                    for (int j = 0; j < LNJ; j++) {
                        if (64 * j >= Mw) continue;
                        const unsigned klo = (unsigned)__builtin_amdgcn_readlane((int)kplo, j);
                        const unsigned khi = (unsigned)__builtin_amdgcn_readlane((int)kphi, j);
                        const unsigned long long km = ((unsigned long long)khi << 32) | klo;
                        const int pos = base + (int)__builtin_amdgcn_mbcnt_hi(khi, __builtin_amdgcn_mbcnt_lo(klo, 0u));
                        if ((km >> lane) & 1ull) crow[pos] = x[j];
                        base += __popcll(km);
                    }
                }
            }
        } else {
            // ---- look D: in band / out of band / undecided; the window's entries go out as they are made -------------
            unsigned long long unc = 0ull;
            for (int wi = 0; wi < nwin; wi++) {
                load_window(wi);
                unsigned inlo = 0u, inhi = 0u;
#pragma unroll
                for (int j = 0; j < LNJ; j++) {
                    if (64 * j >= Mw) continue;
                    const double u = fabs(x[j] - median);
                    const unsigned long long im = __ballot(u < s_lo);
                    unc |= ~(im | __ballot(u > s_hi));
                    inlo = (unsigned)sk_writelane_i32((int)(unsigned)im, j, (int)inlo);
                    inhi = (unsigned)sk_writelane_i32((int)(unsigned)(im >> 32), j, (int)inhi);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int nent = (Mw + 63) >> 6;
                if (lane < nent) a.mask2[(int64_t)r * a.row16 + wi * LNJ + lane] = make_uint4(inlo, inhi, kplo, kphi);
            }
            if (n > 0 && unc != 0ull) ok = false;
        }
        if (lane == 0) {
            a.prep[r] = pr;
            if (MODE == MODE_SEG) a.len_out[r] = M;
            if (!ok) a.retry[1 + atomicAdd(&a.retry[0], 1)] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Reads of 4 097 .. 41 472 samples, ONE LOOK (round 5): a WORKGROUP owns the read and keeps all of it in registers --
// W wavefronts x GNJ slots x 64 lanes; wave w, slot j, lane l holds sample (w GNJ + j) 64 + l -- so the algorithm of
// k_f64_stats runs unchanged with "wavefront" replaced by "workgroup": the histogram (one, in LDS, every wave adds to it),
// the member list and the reductions (through LDS: each wave's partial, a barrier, every wave folds all W of them in
// the same order and so holds the same value) are the workgroup's; the rank select is done by every wave redundantly
// (it only reads); the exact ranking of up to 512 members by wave 0.  k_f64_long looked at a read three (segmenter) to
// five (medmad) times, each look a pass over HBM: 0.19 of the HBM roof at 20 000 samples.  Reads beyond 41 472 samples
// keep k_f64_long.  W wavefronts of GNJ slots: 4 / 8 / 16 x 40 (80 VGPRs of samples, 128 in all: four waves per SIMD) for
// reads of up to 10 240 / 20 480 / 40 960 samples, 12 x 54 (168 VGPRs, three per SIMD) up to 41 472.
// ------------------------------------------------------------------------------------------------------------------

// A wave-uniform value, in scalar registers, that the compiler has to treat as new: the slot loops of the kernel below
// are unrolled 54 times and run several times over the same registers; without this the compiler shares what they
// compute (|x - median| of every slot, the 54 "slot inside the read" tests) across the looks and spills hundreds of
// registers to keep it.
__device__ __forceinline__ int opaque_s(int v) { asm volatile("" : "+s"(v)); return v; }
__device__ __forceinline__ double opaque_s(double v)
{
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return __hiloint2double(hi, lo);
}

// exact ranks j1 (and j1 + 1 when two_ranks) among list[0 .. c), c <= CAPL, not all equal, by ONE wavefront: re-histogram
// in LDS until at most 64 are left, then rank them one per lane (k_f64_long's list_select as a function)
__device__ __forceinline__ bool list_select_wave(unsigned *hist, double *list, unsigned *cnt, int lane, int c, int j1, bool two_ranks,
                                 double &v1, double &v2)
{
    const double INF = __builtin_huge_val();
    auto clear_hist = [&]() {
#pragma unroll
        for (int q = 0; q < PER / 4; q++) *(uint4 *)(hist + lane * PER + 4 * q) = make_uint4(0u, 0u, 0u, 0u);
        hist[NB + lane] = 0u;
    };
    bool have2 = false;
    double m2 = INF;
    for (int it = 0; it < 12; it++) {
        if (c <= 64) {
            const double m = (lane < c) ? list[lane] : INF;
            int rk = 0;
            for (int k = 0; k < c; k++) {
                const double mk = list[k];
                rk += (mk < m || (mk == m && k < lane)) ? 1 : 0;
            }
            const unsigned long long h1 = __ballot(lane < c && rk == j1);
            if (h1 == 0ull) return false;
            v1 = readlane64(m, (int)__builtin_ctzll(h1));
            v2 = v1;
            if (have2) v2 = m2;
            else if (two_ranks) {
                const unsigned long long h2 = __ballot(lane < c && rk == j1 + 1);
                if (h2 == 0ull) return false;
                v2 = readlane64(m, (int)__builtin_ctzll(h2));
            }
            return true;
        }
        double e[CAPL / 64];
        double lmn = INF, lmx = -INF;
#pragma unroll
        for (int k = 0; k < CAPL / 64; k++) {
            e[k] = (lane + 64 * k < c) ? list[lane + 64 * k] : INF;
            if (lane + 64 * k < c) { lmn = vmin64(lmn, e[k]); lmx = vmax64(lmx, e[k]); }
        }
        lmn = readlane64(wave_min64(lmn), 0);
        lmx = readlane64(wave_max64(lmx), 0);
        if (lmn == lmx) { v1 = lmn; v2 = have2 ? m2 : lmn; return true; }
        const double sc = ((double)NB - 0.5) / (lmx - lmn);
        if (!(sc > 0.0 && sc < 1e300)) return false;
#pragma unroll
        for (int k = 0; k < CAPL / 64; k++)
            if (lane + 64 * k < c) atomicAdd(&hist[(unsigned)((e[k] - lmn) * sc)], 1u);
        const bool want2 = two_ranks && !have2;
        RankSel r1, r2;
        rank_select2(hist, lane, j1, want2 ? j1 + 1 : j1, r1, r2);
        clear_hist();
        if (lane == 0) cnt[0] = 0u;
        if (r1.b < 0 || r2.b < 0 || r1.c > c) return false;
        const bool two = want2 && r2.b != r1.b;
        double t2 = INF;
#pragma unroll
        for (int k = 0; k < CAPL / 64; k++) {
            if (lane + 64 * k < c) {
                const unsigned b = (unsigned)((e[k] - lmn) * sc);
                if (b == (unsigned)r1.b) list[atomicAdd(&cnt[0], 1u)] = e[k];      // (every read of the old list is done)
                if (two && b == (unsigned)r2.b) t2 = vmin64(t2, e[k]);
            }
        }
        if (two) { m2 = readlane64(wave_min64(t2), 0); have2 = true; }   // rank j1 + 1 = the first value of the next occupied bin
        c = r1.c; j1 -= r1.pre;
    }
    return false;
}

template <int W, int GNJ, int OCC, int MODE>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
void k_f64_wg(const F64StatArgs a)
{
    __shared__ __align__(16) unsigned hist[NB + 64];               // [NB]: dump bin (dropped samples, slots past the end)
    __shared__ __align__(16) double list[CAPL];
    __shared__ __align__(16) double red[W][8];                      // each wave's partials of the reduction in progress
    __shared__ double res[4];                                       // wave 0's answers (list_select_wave)
    __shared__ unsigned cnt[2];
    __shared__ int nk_all[W];                                       // kept samples per wave
    __shared__ int flag;                                            // != 0: some wave found an undecided sample
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double INF = __builtin_huge_val();
    constexpr int CAPW = 64 * GNJ;                                  // samples per wave

    auto clear_hist_wg = [&]() {                                    // (between two barriers)
        for (int i = threadIdx.x; i < (NB + 64) / 4; i += 64 * W) ((uint4 *)hist)[i] = make_uint4(0u, 0u, 0u, 0u);
        if (threadIdx.x == 0) cnt[0] = 0u;
    };
    // fold every wave's partials: op(k, a, b) combines slot k.  The values are wave-uniform on entry and workgroup-uniform
    // on return (every wave folds the W partials in the same order).
    auto fold = [&](double (&v)[4], int nv, auto op) {
        if (lane == 0)
            for (int k = 0; k < nv; k++) red[w][k] = v[k];
        __syncthreads();
        for (int k = 0; k < nv; k++) {
            double t = red[0][k];
            for (int ww = 1; ww < W; ww++) t = op(k, t, red[ww][k]);
            v[k] = opaque_s(t);
        }
        __syncthreads();
    };
    clear_hist_wg();
    if (threadIdx.x == 0) flag = 0;
    __syncthreads();

    for (int r = blockIdx.x; r < a.nreads; r += gridDim.x) {
        const int64_t o0 = a.off[r];
        int64_t Mfull = max(a.off[r + 1] - o0, (int64_t)0);
        if (a.len) Mfull = min(Mfull, (int64_t)max(a.len[r], 0));
        const int M = __builtin_amdgcn_readfirstlane((int)min(Mfull, (int64_t)(W * CAPW)));
        const double *row = a.sig + o0;
        const int Mw = __builtin_amdgcn_readfirstlane(min(max(M - w * CAPW, 0), CAPW));    // this wave's samples

        // ---- the read into registers, all of it at once ---------------------------------------------------------
        double x[GNJ];
        const double *prow = row + (int64_t)w * CAPW + lane;
#pragma unroll
        for (int j = 0; j < GNJ; j++) {
            int rem = Mw - 64 * j;
            asm("" : "+s"(rem));                        // (opaque: see k_f64_stats)
            x[j] = (lane < rem) ? prow[64 * j] : INF;
        }
        double K = (M > 0) ? row[0] : 0.0;              // shift of the sums (any finite value near the data)
        if (!(K > a.lo && K < a.hi)) K = 0.5 * (a.lo + a.hi);
        if (!(fabs(K) < 1e300)) K = 0.0;

        // ---- first look: filter, extremes, shifted sums -----------------------------------------------------------
        unsigned kplo = 0u, kphi = 0u;                  // lane j: the kept word of slot j (of this wave)
        int nkw = 0;
        double mn = INF, mx = -INF, S1 = 0.0, S2 = 0.0;
        const int Mw1 = opaque_s(Mw);
#pragma unroll
        for (int j = 0; j < GNJ; j++) {
            if (64 * j >= Mw1) continue;                // (wave-uniform)
            const bool k = x[j] > a.lo && x[j] < a.hi;
            const unsigned long long km = __ballot(k);
            kplo = (unsigned)sk_writelane_i32((int)(unsigned)km, j, (int)kplo);
            kphi = (unsigned)sk_writelane_i32((int)(unsigned)(km >> 32), j, (int)kphi);
            nkw += __popcll(km);
            if (km != ~0ull) x[j] = k ? x[j] : INF;
            if (k) {
                mn = vmin64(mn, x[j]);
                mx = vmax64(mx, x[j]);
                if (MODE == MODE_SEG) {
                    const double d = x[j] - K;
                    S1 += d;
                    S2 = fma(d, d, S2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (lane == 0) nk_all[w] = nkw;
        double f4[4];
        f4[0] = readlane64(wave_min64(mn), 0);
        f4[1] = readlane64(wave_max64(mx), 0);
        f4[2] = (MODE == MODE_SEG) ? readlane64(wave_sum64(S1), 0) : 0.0;
        f4[3] = (MODE == MODE_SEG) ? readlane64(wave_sum64(S2), 0) : 0.0;
        fold(f4, MODE == MODE_SEG ? 4 : 2, [](int k, double p, double q) { return k == 0 ? (q < p ? q : p) : k == 1 ? (q > p ? q : p) : p + q; });
        mn = f4[0]; mx = f4[1]; S1 = f4[2]; S2 = f4[3];
        int n = 0, nbelow = 0;                          // kept samples of the read / of the waves before this one
        for (int ww = 0; ww < W; ww++) { const int t = nk_all[ww]; n += t; nbelow += (ww < w) ? t : 0; }

        sk_prep pr;
        pr.n = n; pr.flags = 0; pr.center = 0.0; pr.scale = 1.0; pr.top = 0.0; pr.bot = 0.0;
        bool ok = true;

        // exact order statistics k1 <= k2 <= k1 + 1 of val(x) over the kept samples, val in [vlo, vhi], vlo < vhi.
        // Every argument and every result is workgroup-uniform.
        auto select2 = [&](auto val, double center, double vlo, double vhi, int k1, int k2, double &v1, double &v2) -> bool {
            bool restricted = false;                    // later rounds: only values inside [vlo, vhi] take part
            bool have2 = false;                         // v2 already known (k2 fell into the next occupied bin)
#pragma unroll 1
            for (int round = 0; round < 6; round++) {
                const double sc = ((double)NB - 0.5) / (vhi - vlo);
                if (!(sc > 0.0 && sc < 1e300)) return false;
                {
                    const int Mwo = opaque_s(Mw);
                    const double c_o = opaque_s(center), lo_o = opaque_s(vlo), hi_o = opaque_s(vhi), sc_o = opaque_s(sc);
#pragma unroll
                    for (int j = 0; j < GNJ; j++) {
                        if (64 * j >= Mwo) continue;
                        const double v = val(x[j], c_o);
                        bool in = x[j] != INF;
                        if (restricted) in = in && v >= lo_o && v <= hi_o;
                        atomicAdd(&hist[in ? (unsigned)((v - lo_o) * sc_o) : (unsigned)NB], 1u);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __syncthreads();
                RankSel r1, r2;
                rank_select2(hist, lane, k1, have2 ? k1 : k2, r1, r2);
                __syncthreads();
                clear_hist_wg();
                __syncthreads();
                if (r1.b < 0 || r2.b < 0) return false;
                const bool two = !have2 && r2.b != r1.b;
                double mnm = INF, mxm = -INF, mn2 = INF;
                {
                    const int Mwo = opaque_s(Mw);
                    const double c_o = opaque_s(center), lo_o = opaque_s(vlo), hi_o = opaque_s(vhi), sc_o = opaque_s(sc);
                    const unsigned b1 = (unsigned)r1.b, b2 = two ? (unsigned)r2.b : 0xffffffffu;
#pragma unroll
                    for (int j = 0; j < GNJ; j++) {
                        if (64 * j >= Mwo) continue;
                        const double v = val(x[j], c_o);
                        bool in = x[j] != INF;
                        if (restricted) in = in && v >= lo_o && v <= hi_o;
                        const unsigned b = in ? (unsigned)((v - lo_o) * sc_o) : (unsigned)NB;
                        if (b == b1) {
                            const unsigned slot = atomicAdd(&cnt[0], 1u);
                            if (slot < (unsigned)CAPL) list[slot] = v;
                            mnm = vmin64(mnm, v);
                            mxm = vmax64(mxm, v);
                        }
                        if (b == b2) mn2 = vmin64(mn2, v);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                double g4[4];
                g4[0] = readlane64(wave_min64(mnm), 0);
                g4[1] = readlane64(wave_max64(mxm), 0);
                g4[2] = readlane64(wave_min64(mn2), 0);
                g4[3] = 0.0;
                fold(g4, 3, [](int k, double p, double q) { return k == 1 ? (q > p ? q : p) : (q < p ? q : p); });
                mnm = g4[0]; mxm = g4[1];
                if (two) { v2 = g4[2]; have2 = true; }
                const int j1 = k1 - r1.pre;
                const bool both_here = !have2 && k2 != k1;          // k2 = k1 + 1 lies in the same bin
                if (mnm == mxm) {
                    v1 = mnm;
                    if (!have2) v2 = mnm;
                    if (threadIdx.x == 0) cnt[0] = 0u;
                    __syncthreads();
                    return true;
                }
                if (r1.c <= CAPL) {
                    if (w == 0) {
                        double t1 = 0.0, t2 = 0.0;
                        const bool good = list_select_wave(hist, list, cnt, lane, r1.c, j1, both_here, t1, t2);
                        if (lane == 0) { res[0] = good ? 1.0 : 0.0; res[1] = t1; res[2] = t2; cnt[0] = 0u; }
                    }
                    __syncthreads();
                    const bool good = res[0] != 0.0;
                    v1 = res[1];
                    if (!have2) v2 = res[2];
                    __syncthreads();
                    return good;
                }
                // too many distinct members: again, on the members' own range
                if (threadIdx.x == 0) cnt[0] = 0u;
                __syncthreads();
                vlo = mnm; vhi = mxm; restricted = true;
                k1 = j1; k2 = both_here ? j1 + 1 : j1;
            }
            return false;
        };

        double median = 0.0, s_lo = -1.0, s_hi = -1.0;
        if (n == 0) {
            pr.flags = SK_FLAG_EMPTY;
            const double qnan = __builtin_nan("");
            pr.center = qnan; pr.scale = qnan; pr.top = qnan; pr.bot = qnan;
        } else {
            const int k1 = (n - 1) / 2, k2 = n / 2;
            double v1 = mn, v2 = mn;
            if (mn != mx) ok = select2([](double v, double) { return v; }, 0.0, mn, mx, k1, k2, v1, v2);
            median = (k1 == k2) ? v1 : (v1 + v2) / 2.0;                      // np.median: mean of the two middle elements
            if (MODE == MODE_MEDMAD) {
                double w1 = 0.0, w2 = 0.0;
                const double umax = vmax64(fabs(mn - median), fabs(mx - median));
                if (ok && umax > 0.0)
                    ok = select2([](double v, double c) { return fabs(v - c); }, median, 0.0, umax, k1, k2, w1, w2);
                const double mad = (k1 == k2) ? w1 : (w1 + w2) / 2.0;
                pr.center = median;
                pr.scale = mad * 1.4826;                                     // MotifSeq.py:196
                if (mad == 0.0) pr.flags |= SK_FLAG_DEGENERATE;
            } else {
                const double dn = (double)n;
                const double md = S1 / dn, Q = S2 / dn;
                const double var = Q - md * md;
                const double Ev = 8.0 * (dn + 8.0) * U53 * Q;                // |var - var_true|
                const double sd = sqrt(var);
                const double A = vmax64(fabs(mn), fabs(mx));
                const double dstd = Ev / sd + dn * U53 * A + sd * (dn + 8.0) * U53;     // |sd - numpy's std|
                const double spread = sd * a.std_scale;                      // segmenter.py:413-414
                const double dlt = a.delta_scale * 4.0 *
                                   (fabs(a.std_scale) * dstd + U53 * (4.0 * fabs(spread) + fabs(median) + 2.0 * A));
                if (!(var > 4.0 * Ev)) ok = false;                           // (also NaN; all-equal reads)
                s_lo = spread - dlt; s_hi = spread + dlt;
                pr.center = median; pr.scale = sd; pr.top = median + spread; pr.bot = median - spread;
            }
        }

        if (MODE == MODE_MEDMAD) {
            if (n == M) {
                pr.flags |= SK_IFLAG_INPLACE;           // nothing dropped: the DTW feed reads the input itself
            } else {
                double *crow = a.comp + o0;             // the filtered samples, in order
                int base = nbelow;
                const int Mwo = opaque_s(Mw);
#pragma unroll
                for (int j = 0; j < GNJ; j++) {
                    if (64 * j >= Mwo) continue;
                    const unsigned klo = (unsigned)__builtin_amdgcn_readlane((int)kplo, j);
                    const unsigned khi = (unsigned)__builtin_amdgcn_readlane((int)kphi, j);
                    const unsigned long long km = ((unsigned long long)khi << 32) | klo;
                    const int pos = base + (int)__builtin_amdgcn_mbcnt_hi(khi, __builtin_amdgcn_mbcnt_lo(klo, 0u));
                    if ((km >> lane) & 1ull) crow[pos] = x[j];
                    base += __popcll(km);
                }
            }
        } else {
            // ---- second look at the registers: in band / out of band / undecided --------------------------------
            unsigned inlo = 0u, inhi = 0u;
            unsigned long long unc = 0ull;
            const int Mwo = opaque_s(Mw);
#pragma unroll
            for (int j = 0; j < GNJ; j++) {
                if (64 * j >= Mwo) continue;
                const double u = fabs(x[j] - median);                        // dropped samples: +inf, out of band
                const unsigned long long im = __ballot(u < s_lo);
                unc |= ~(im | __ballot(u > s_hi));
                inlo = (unsigned)sk_writelane_i32((int)(unsigned)im, j, (int)inlo);
                inhi = (unsigned)sk_writelane_i32((int)(unsigned)(im >> 32), j, (int)inhi);
                __builtin_amdgcn_sched_barrier(0);
            }
            const int nent = (Mw + 63) >> 6;
            if (lane < nent) a.mask2[(int64_t)r * a.row16 + w * GNJ + lane] = make_uint4(inlo, inhi, kplo, kphi);
            if (n > 0 && unc != 0ull && lane == 0) atomicOr(&flag, 1);
            __syncthreads();
            if (flag != 0) ok = false;
        }
        if (threadIdx.x == 0) {
            a.prep[r] = pr;
            if (MODE == MODE_SEG) a.len_out[r] = M;
            if (!ok) a.retry[1 + atomicAdd(&a.retry[0], 1)] = r;
        }
        __syncthreads();                                // (every wave has read `flag` / nk_all)
        if (threadIdx.x == 0) flag = 0;
    }
}

typedef void (*f64stat_fn)(const F64StatArgs);
constexpr int64_t F64_WG_MAX = 12 * 64 * 54;               // longest read the one-look kernel holds (41 472 samples)

// Wavefronts per workgroup of the one-look kernel for this call, 0: another kernel.  The wavefronts of a workgroup have
// to spread evenly over the four SIMDs (6 waves at 168 VGPRs: two SIMDs take two of them and have no room for a second
// workgroup's -- measured: half the chip idle), so 4 / 8 waves of 40 slots at 128 VGPRs (4 / 2 workgroups per CU) up to
// 10 240 / 20 480 samples and 12 waves of 54 slots at 168 VGPRs beyond (16 x 40 at 128: 3.79 against 3.56 ms at 36 977
// samples).  Measured (MI355X, round 5): segmenter statistics 4.32 -> 3.77 ms at 50 000 x 19 999, 4.30 -> 3.56 ms at
// 25 000 x 36 977; medmad (two selections, 52-86 spilled registers at these budgets) 6.5 -> 7.0 ms at 19 999 samples --
// it keeps the window-by-window kernel there -- and 6.1 -> 5.95 ms at 36 977.
static int wg_waves(int mode, int64_t maxlen)
{
    if (maxlen <= 4096 || maxlen > F64_WG_MAX || sk_tune("SK_F64_LONG_LOOKS")) return 0;
    if (mode == MODE_SEG) return maxlen <= 4 * 64 * 40 ? 4 : maxlen <= 8 * 64 * 40 ? 8 : 12;
    return maxlen > 8 * 64 * 40 ? 12 : 0;
}

f64stat_fn pick(int mode, int64_t maxlen)
{
    const bool seg = mode == MODE_SEG;
    if (wg_waves(mode, maxlen)) {
        if (maxlen <= 4 * 64 * 40)  return k_f64_wg<4, 40, 4, MODE_SEG>;
        if (maxlen <= 8 * 64 * 40)  return k_f64_wg<8, 40, 4, MODE_SEG>;
        return seg ? k_f64_wg<12, 54, 3, MODE_SEG> : k_f64_wg<12, 54, 3, MODE_MEDMAD>;
    }
    if (maxlen > 4096) return seg ? k_f64_long<MODE_SEG> : k_f64_long<MODE_MEDMAD>;
    if (maxlen <= 1024) return seg ? k_f64_stats<16, MODE_SEG, 6> : k_f64_stats<16, MODE_MEDMAD, 6>;
    if (maxlen <= 2048) return seg ? k_f64_stats<32, MODE_SEG, 4> : k_f64_stats<32, MODE_MEDMAD, 4>;
    return seg ? k_f64_stats<64, MODE_SEG, 3> : k_f64_stats<64, MODE_MEDMAD, 3>;
}

} // namespace

// Is (longest read, std_scale) inside the streaming float64 path's range?  (else: k_prep_f64 for every read)
bool sk_f64_fast_applies(int64_t maxlen, double std_scale)
{
    if (sk_tune("SK_F64_OLD")) return false;                  // A/B switch: the numpy-order kernel for everything
    if (maxlen > (1 << 20)) return false;                    // (window-by-window kernel from 4 097 samples to 2^20)
    if (!(std_scale == std_scale) || fabs(std_scale) > 1e6) return false;
    return true;
}

int sk_f64_row16(int64_t maxlen)
{
    const int64_t e = (maxlen + 63) / 64;
    return (int)((e + 7) & ~(int64_t)7);                     // whole 128-byte lines per read
}

// mode: SK_PREP_SEGMENT (masks + len_out) or SK_PREP_MEDMAD (comp).  d_retry: nreads + 16 ints, zeroed here; the
// caller runs the numpy-order kernel over that list next.
int sk_launch_f64_stats(sk_ctx *c, const double *d_sig, const int64_t *d_off, const int32_t *d_rlen, int32_t nreads, int64_t maxlen,
                        double lo, double hi, int mode, double std_scale, sk_prep *d_prep, void *d_mask2, int row16,
                        int32_t *d_len, int32_t *d_retry, double *d_comp)
{
    if (nreads <= 0) return SK_OK;
    F64StatArgs a;
    a.sig = d_sig; a.off = d_off; a.len = d_rlen; a.nreads = nreads; a.lo = lo; a.hi = hi; a.std_scale = std_scale;
    a.delta_scale = 1.0;
    if (const char *e = sk_tune("SK_SEG_DELTA_SCALE")) { const double v = atof(e); if (v > 0) a.delta_scale = v; }
    a.prep = d_prep; a.mask2 = (uint4 *)d_mask2; a.row16 = row16; a.len_out = d_len; a.retry = d_retry; a.comp = d_comp;
    f64stat_fn fn = pick(mode == SK_PREP_SEGMENT ? MODE_SEG : MODE_MEDMAD, maxlen);
    SK_HIP(hipMemsetAsync(d_retry, 0, 16 * sizeof(int32_t), c->stream));
    // a workgroup per read (one-look kernel: 2 .. 16 wavefronts by the longest read) or WPB reads per workgroup
    const int wgw = wg_waves(mode == SK_PREP_SEGMENT ? MODE_SEG : MODE_MEDMAD, maxlen);
    const bool wg = wgw > 0;
    const int waves = wg ? wgw : WPB;
    const int per_block = wg ? 1 : WPB;
    int resident = 2;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, (const void *)fn, 64 * waves, 0) != hipSuccess || resident < 1)
        resident = wg ? 1 : 2;
    int rounds = 8;
    if (const char *e = sk_tune("SK_PREP_ROUNDS")) { int v = atoi(e); if (v > 0) rounds = v; }
    const long long g = (long long)c->num_cu * resident * rounds;
    const long long need = ((long long)nreads + per_block - 1) / per_block;
    const int grid = (int)(g > need ? need : g);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * waves), 0, c->stream, a);
    SK_HIP(hipGetLastError());
    return SK_OK;
}
