// sk_segstat.hip -- the segmenter's filter + statistics + classification as ONE streaming pass (gfx950).
//
// Covers, per read (segmenter.py):
//   scale_outliers      :311-318     strict lo < x < hi
//   np.median, np.std   :410, :412   -> top / bot (:413-414)
//   a < top and a > bot :431         one bit per sample
// and hands the walk kernel (k_seg_walk2 below, the state machine of :420-464) two bit masks per read in RAW
// sample coordinates: "in band" and "kept by the filter".  The walk deletes the dropped samples' bits on the
// fly, so nothing in here needs an order-preserving compaction.
//
// Why this is allowed to be simpler than numpy's arithmetic (k_prep_i16 in sk_prep.hip reproduces np.std's
// summation order bit for bit and pays for it with a dozen dependent phases per read): the samples are
// integers, so the only thing the state machine ever sees of top / bot is ceil(top) and floor(bot).
//   * n, sum(x) and sum(x^2) are exact integers (v_dot2c_i32_i16 over the packed samples), so
//     V = n sum(x^2) - sum(x)^2 is exact in int64 and std_true = sqrt(V) / n to a few ulp.
//   * numpy's std differs from std_true by at most ~(n + 16) eps relative (mean: one rounding; each
//     (x - mean)^2: three; a sum of n non-negative terms in ANY order: <= (n - 1) eps), so
//     |top_numpy - top_here| <= delta := 8 eps (|spread| (n + 16) + |median| + |spread|)   (8x headroom).
//   * if no integer lies within delta of top (resp. bot), ceil(top) (floor(bot)) is numpy's.  CERTIFIED.
//     Otherwise (probability ~1e-10 per read) the read goes to a retry list and is redone by the
//     numpy-order kernel (k_prep_i16 + k_segment_walk over the listed reads only).
// One wavefront owns one read from its first load to its last store: the whole read (<= 4096 samples) sits in
// 32 VGPRs as packed int16 pairs, all loads are issued up front, no workgroup barrier anywhere, 32 reads in
// flight per CU.  LDS: one value histogram per wave (median by rank select from registers, as k_prepw_medmad).
// Algorithmic HBM traffic per read: 2 M in, M / 4 out (two bit masks), 48 B of statistics.
#include "sk_common.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int WPB = 4;                 // wavefronts (= reads in flight) per workgroup
constexpr int MAXBINS = 2047;          // t' = min(x - lo - 1, nbins) must keep (t' << 2) inside 16 bits, sum(t'^2) inside 32
constexpr int MAXLONG = 1 << 20;       // longest read: n (n sum t^2) - (sum t)^2 stays inside 63 bits

typedef short          s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_clamp_i16(unsigned q, unsigned lo2, unsigned hi2)
{
    const s16x2 x = __builtin_bit_cast(s16x2, q);
    const s16x2 c = __builtin_elementwise_min(__builtin_elementwise_max(x, __builtin_bit_cast(s16x2, lo2)),
                                              __builtin_bit_cast(s16x2, hi2));
    return __builtin_bit_cast(unsigned, c);
}
// (inline asm: given the generic vector operations the compiler turns these short packed sequences into
// SDWA compares + selects + a permute, two to three times the instructions.  The second operand of each is a
// wave-uniform constant: it rides in an SGPR, so no v_mov and no vector register is spent on it.)
__device__ __forceinline__ unsigned pk_sub_u16(unsigned a, unsigned b_uniform)
{
    unsigned r;
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b_uniform));
    return r;
}
__device__ __forceinline__ unsigned pk_subsat_u16(unsigned a, unsigned b_uniform)      // max(a - b, 0) per half
{
    unsigned r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "s"(b_uniform));
    return r;
}
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b_uniform)
{
    unsigned r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b_uniform));
    return r;
}
__device__ __forceinline__ unsigned pk_max_i16(unsigned a, unsigned b)          // signed maximum per half
{
    unsigned r;
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned pk_min1_u16(unsigned a)                     // min(a, 1) per half
{
    unsigned r;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ unsigned pk_shl2_u16(unsigned a)                     // both halves << 2
{
    unsigned r;
    asm("v_pk_lshlrev_b16 %0, 2, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ unsigned udot2(unsigned a, unsigned b_uniform, unsigned c)   // a.lo b.lo + a.hi b.hi + c
{
    unsigned r;
    asm("v_dot2_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
    return r;
}

// inclusive scan across the wavefront on the vector ALU (see sk_prep.hip)
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
__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_incl_scan(v), 63); }
// inclusive running maximum across the wavefront (unsigned; lanes without a source contribute 0)
__device__ __forceinline__ unsigned wave_incl_max(unsigned v)
{
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
    return v;
}
// lane i <- lane i - 1 across the whole wavefront (lane 0 <- 0)
__device__ __forceinline__ unsigned wave_shr1(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);
}

// Hints for k_seg_walk4, SEG_HINTS dwords per read: [0] = number of stretches of quiet entries (SEG_HINT_NONE: walk
// this read without jumps), [1 + i] = stretch i: anchor (12 bits) | last entry << 12 (6 bits) | samples dropped before
// the anchor's entry << 18 (13 bits).
constexpr int SEG_HINTS = 8;
constexpr unsigned SEG_HINT_NONE = 255u;

// lane i <- lane i + N inside a row of 16 (lanes without a source get 0)
template <int N>
__device__ __forceinline__ unsigned dpp_shl(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xF, 0xF, true);
}

struct SegStatArgs {
    const int16_t *sig;
    int64_t        stride;
    const int32_t *len;
    int            nreads;
    int            lo, hi;           // outlier limits (strict)
    double         std_scale;
    double         delta_scale;      // 1.0; tests raise it to push reads onto the retry list
    sk_prep       *prep;
    uint4         *mask2;            // [nreads][row16] of {in band lo, hi, kept lo, hi}: 64 raw samples per entry
    int            row16;            // entries per read (8 per 512-sample tile)
    int32_t       *retry;            // [0] = count, [1 ..] = reads that could not be certified
    unsigned      *hints;            // [nreads][SEG_HINTS] for k_seg_walk4 (reads of up to 4 096 samples), or nullptr
    int            e1;               // the walk's error + 1 (hints)
    const double  *cal;              // PA: [nreads][2] = {offset, range / digitisation} of each read (else unused)
};

// ---- the pA route in the raw domain (PA = true; round 6) -----------------------------------------------------------
// segmenter.py:345-349 / :366-370 turn a fast5 / slow5 read into v(x) = np.round((x + offset) * unit, 2) before
// scale_outliers / get_segs see it.  np.round(y, 2) is rint(y * 100) / 100, so v(x) = c(x) / 100 with
//     c(x) = rint(fl(fl(x + offset) * unit) * 100)        an INTEGER ("centi-pA"), non-decreasing in x for unit > 0
// (every step is a monotone function followed by a monotone rounding).  So everything the segmenter does with v can be
// done on the int16 samples themselves, 2 bytes per sample instead of 8:
//   * kept   <=> lim_low < v(x) < lim_hi <=> 100 lim_low < c(x) < 100 lim_hi   (exact: RN(c / 100) > L <=> c > 100 L for
//     an integer L, as c / 100 >= L + 0.01 cannot round down to L)            <=> xa <= x <= xb;
//   * np.median of the kept v = (v(x1) + v(x2)) / 2 with x1, x2 the two middle kept SAMPLES (v is monotone);
//   * np.std of the kept v: the exact integer sums C1 = sum c', C2 = sum c'^2 (c' = c - c(xa)) come out of the read's
//     value histogram, V = n C2 - C1^2 is exact in 128 bits, std_true = sqrt(V) / (100 n) up to the roundings of the
//     c / 100 (<= vmax eps / 2 each) and numpy's own summation error, both inside delta below;
//   * in band <=> bot < v(x) < top <=> x inside [first x with c(x) certainly above 100 bot, last x with c(x) certainly
//     below 100 top]: CERTIFIED when no c(x) of the window lies within Delta of 100 top / 100 bot, Delta from delta.
// delta: |std_numpy - std_true| <= (n + 16) eps std_true (the n roundings of a sum of non-negative terms in ANY order,
// three per term, the division, the root) + |mean_numpy - mean_true| (a shift of the mean by e changes the root of the
// mean square deviation by at most |e|; numpy adds pairwise inside chunks of 8 192 and serially across them:
// |e| <= (n / 8192 + 32) eps vmax); |std_true - sqrt(V) / (100 n)| <= vmax eps (the c / 100) + 4 eps std (the 128-bit
// value to double, root, two divisions); the median's two roundings and top / bot's one: 3 eps vmax.  Hence
//     delta = 8 eps (|spread| (n + 17) + |median| + vmax (1 + |std_scale|) (n / 8192 + 36))         (8x headroom)
// Reads that cannot be certified (about 1e-8), whose kept samples do not fit the histogram window (a spike more than
// HBINS - 1 raw units above xa), or whose calibration is outside the plain range go to the retry list and are redone from
// their float64 values in numpy's order (k_prep_pa_listed, sk_prep.hip).
constexpr double PA_UNIT_MIN = 1.0 / 64, PA_UNIT_MAX = 1.25, PA_OFFSET_MAX = 16777216.0;   // c' < 2^18 over 2 047 bins

__device__ __forceinline__ double pa_centi(int x, double off, double unit)
{
    return rint((((double)x + off) * unit) * 100.0);      // (-ffp-contract=off: three roundings and an exact rint)
}
// first x in [x0, x0 + count) with c(x) >= T (STRICT: > T), or x0 + count; count <= 65 536; c non-decreasing
template <bool STRICT>
__device__ __forceinline__ int pa_first(double T, int x0, int count, double off, double unit, int lane)
{
    int base = x0, span = count;                          // the answer lies in [base, base + span] (span: none)
#pragma unroll 1
    for (int blk = count > 2048 ? 1024 : (count > 64 ? 32 : 1); ; blk = blk > 32 ? 16 : 1) {
        const int x = base + blk * (lane + 1) - 1;        // the last x of my block
        const bool inside = blk * (lane + 1) - 1 < span;
        const double cx = pa_centi(x, off, unit);
        const unsigned long long m = __ballot(inside && (STRICT ? cx > T : cx >= T));
        if (m == 0ull) {
            // nothing among the block ends: the answer lies behind the last complete block (or nowhere)
            const int done = (span / blk) * blk;
            if (blk == 1 || done == span) return x0 + count;
            base += done; span -= done;
        } else {
            const int l1 = (int)__builtin_ctzll(m);
            if (blk == 1) return base + l1;
            base += blk * l1; span = blk;                 // inside block l1, whose end satisfies: always found below
        }
    }
}
__device__ __forceinline__ void umul64wide(unsigned long long a, unsigned long long b, unsigned long long &hi, unsigned long long &lo)
{
    lo = a * b;
    hi = __umul64hi(a, b);
}

// NT: 512-sample tiles held in registers (one "window" of 512 NT samples)
// NQ: 16-byte histogram chunks per lane -- 256 NQ bins per wave
// OCC: wavefronts per SIMD the register allocation is sized for
// LONG: reads longer than one window.  The statistics need the whole read before any sample can be classified, so a
//       long read is looked at twice, window by window: the second look re-reads it (L2 / Infinity Cache / HBM).
//
// Sample images.  t = (x - (lo + 1)) mod 2^16 is the sample's histogram bin: kept <=> t < nbins (hi <= 32768 makes
// every dropped x land at t >= nbins, no aliasing), and t' = min(t, nbins) sends every dropped sample -- and the
// slots past the read's end -- to ONE dump bin, `nbins`.  Two packed instructions per pair of samples; the exact
// sums run over t' and are corrected by the dump bin's count afterwards.
template <int NT, int NQ, int OCC, bool LONG, bool PA = false>
__global__ __launch_bounds__(64 * WPB, OCC)
void k_seg_stats(const SegStatArgs a)
{
    constexpr int HBINS = 64 * 4 * NQ;
    constexpr int WIN = 512 * NT;
    __shared__ __align__(16) unsigned hist_all[WPB][HBINS];
    __shared__ __align__(16) unsigned char plane_all[WPB][2][64 * NT];     // one byte per 8 samples: in band / dropped
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // (scalar: r, M and every row address too)
    unsigned *hist = hist_all[w];
    unsigned char *p_in = plane_all[w][0], *p_dr = plane_all[w][1];
    const int hb0 = lane * 4 * NQ;                         // first bin this lane owns

    // (PA: the limits, and with them the histogram window, are per read -- wave-uniform values set at the top of the loop)
    int nbins = a.hi - a.lo - 1;                           // 1 .. min(HBINS - 1, MAXBINS) (host)
    int lo_r = a.lo;                                       // bin t holds the sample value lo_r + 1 + t
    unsigned lo1p = (unsigned)((a.lo + 1) & 0xffff) * 0x10001u;
    unsigned nbp = (unsigned)nbins * 0x10001u, nbm1p = (unsigned)(nbins - 1) * 0x10001u;
    const int maxM = (int)min(a.stride, (int64_t)(LONG ? MAXLONG : WIN));

#pragma unroll
    for (int j = 0; j < NQ; j++) *(uint4 *)(hist + hb0 + 4 * j) = make_uint4(0u, 0u, 0u, 0u);

    unsigned y[NT][4];                                     // one window of the read: packed samples, then their t'
    // LONG (round 6): a second window.  The second look walks the read backwards and finds the t' of its last two windows
    // still in registers -- 8 192 of a 20 000-sample read's samples are not fetched a second time (the kernel sits at the
    // HBM ceiling because of that second look: profiles/r06_pa_long_counters.txt); 32 more VGPRs, same four waves per SIMD
    // (the LDS histograms hold it there anyway).
    unsigned y2[NT][4];                                    // (unused, and dropped by the compiler, when !LONG or KEEP == 1)
    // KEEP windows stay in registers between the looks: 2 on the int16 route (statistics kernel of 50 000 x 20 000 samples
    // 0.69 -> 0.54 ms, of 100 000 x 8 192 -- both windows kept, one look -- 0.67 -> 0.40); none on the pA route, whose
    // statistics need ~80 registers of their own: one kept window is 144 VGPRs (three waves per SIMD: 0.70 -> 0.77 ms at
    // 20 000 samples, 0.74 -> 0.67 at 37 000), two spill.
    constexpr int KEEP = PA ? 0 : 2;
    // the window's samples into registers: NT x 16-byte loads per lane, all in flight at once
    auto load_window = [&](unsigned (&yy)[NT][4], const int16_t *wrow, int Mw) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NT; t++) {
            uint4 q = make_uint4(0u, 0u, 0u, 0u);
            if (t * 512 + lane * 8 < Mw) q = *(const uint4 *)(wrow + t * 512 + lane * 8);   // (rows are 16-byte aligned)
            yy[t][0] = q.x; yy[t][1] = q.y; yy[t][2] = q.z; yy[t][3] = q.w;
        }
    };
    // t' of one packed pair; the read's last tile also sends the slots past its end to the dump bin
    auto image = [&](unsigned q, int t, int k, int ntiles, int nvalid) -> unsigned {
        unsigned tt = pk_min_u16(pk_sub_u16(q, lo1p), nbp);
        if (t == ntiles - 1) {                             // (wave-uniform)
            const unsigned tail = nvalid >= 2 * k + 2 ? 0xffffffffu : (nvalid == 2 * k + 1 ? 0xffffu : 0u);
            tt = (tt & tail) | (nbp & ~tail);
        }
        return tt;
    };

    const int nwaves = gridDim.x * WPB;
    for (int r = blockIdx.x * WPB + w; r < a.nreads; r += nwaves) {
        const int M = __builtin_amdgcn_readfirstlane(min(max(a.len[r], 0), maxM));
        const int16_t *row = a.sig + (int64_t)r * a.stride;
        // (measured and dropped: touching the wave's NEXT read with one 4-byte load per 128-byte line, to have it on
        // its way into the L2 -- 2.34 ms against 2.00 ms per 1 M reads: the extra requests cost more than they hide)
        const int nwin = LONG ? (M + WIN - 1) / WIN : 1;
        const int tiles_total = (M + 511) >> 9;

        // ---- PA: this read's limits in the raw domain, its histogram window -------------------------------------
        bool pa_ok = true;
        double pa_off = 0.0, pa_unit = 1.0, pa_c0 = 0.0;
        int pa_span = 0;                                   // sample values the filter keeps: xa .. xa + pa_span - 1
        if constexpr (PA) {
            pa_off = a.cal[2 * r]; pa_unit = a.cal[2 * r + 1];
            pa_ok = pa_unit >= PA_UNIT_MIN && pa_unit <= PA_UNIT_MAX && fabs(pa_off) <= PA_OFFSET_MAX;   // (NaN: false)
            int xa = 0, xb = -1;
            if (pa_ok) {
                xa = pa_first<true>(100.0 * (double)a.lo, -32768, 65536, pa_off, pa_unit, lane);       // first kept value
                xb = pa_first<false>(100.0 * (double)a.hi, -32768, 65536, pa_off, pa_unit, lane) - 1;  // last kept value
            }
            xa = __builtin_amdgcn_readfirstlane(xa); xb = __builtin_amdgcn_readfirstlane(xb);
            pa_span = max(xb - xa + 1, 0);
            nbins = min(pa_span, HBINS - 1);
            lo_r = xa - 1;
            lo1p = (unsigned)(xa & 0xffff) * 0x10001u;
            nbp = (unsigned)nbins * 0x10001u; nbm1p = (unsigned)((nbins - 1) & 0xffff) * 0x10001u;
            pa_c0 = pa_centi(xa, pa_off, pa_unit);
            if (nbins > 0 && !(pa_centi(xa + nbins - 1, pa_off, pa_unit) - pa_c0 < 262144.0)) { pa_ok = false; nbins = 0; }
            if (nbins == 0) {
                // nothing can be kept (or a calibration outside the plain range: the redo rewrites all of this)
                for (int wi = 0; wi < nwin; wi++) {
                    const int ntiles = (min(M - wi * WIN, WIN) + 511) >> 9;
                    if (lane < 8 * ntiles) a.mask2[(int64_t)r * a.row16 + wi * (8 * NT) + lane] = make_uint4(0u, 0u, 0u, 0u);
                }
                if (lane == 0) {
                    sk_prep pe;
                    const double qnan = __builtin_nan("");
                    pe.n = 0; pe.flags = SK_FLAG_EMPTY; pe.center = qnan; pe.scale = qnan; pe.top = qnan; pe.bot = qnan;
                    a.prep[r] = pe;
                    if (!pa_ok) a.retry[1 + atomicAdd(&a.retry[0], 1)] = r;
                    if (!LONG && a.hints) a.hints[(int64_t)r * SEG_HINTS] = SEG_HINT_NONE;
                }
                continue;
            }
        }

        // ---- first look: t' per sample, exact sums, histogram -------------------------------------------------
        long long S = 0, Q = 0;
        unsigned mxp = 0x80008000u;                        // PA: running maximum of the raw samples (packed halves)
        auto first_window = [&](unsigned (&yy)[NT][4], int wi) __attribute__((always_inline)) {
            const int Mw = min(M - wi * WIN, WIN);
            const int ntiles = (Mw + 511) >> 9;
            load_window(yy, row + (int64_t)wi * WIN, Mw);
            int st = 0, stt = 0;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (t >= ntiles) continue;                 // (wave-uniform) nothing of the read in this tile
                const int nvalid = min(max(Mw - (t * 512 + lane * 8), 0), 8);   // samples of this lane's eight that exist
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if constexpr (PA) {                    // a kept value above the histogram window? (the maximum tells)
                        unsigned qm = yy[t][k];
                        if (t == ntiles - 1) {
                            const unsigned tail = nvalid >= 2 * k + 2 ? 0xffffffffu : (nvalid == 2 * k + 1 ? 0xffffu : 0u);
                            qm = (qm & tail) | (0x80008000u & ~tail);
                        }
                        mxp = pk_max_i16(mxp, qm);
                    }
                    const unsigned tt = image(yy[t][k], t, k, ntiles, nvalid);
                    yy[t][k] = tt;
                    const s16x2 ts = __builtin_bit_cast(s16x2, tt);
                    if constexpr (!PA) {
                    st = __builtin_amdgcn_sdot2(ts, __builtin_bit_cast(s16x2, 0x10001u), st, false);
                    stt = __builtin_amdgcn_sdot2(ts, ts, stt, false);
                    }
                    const unsigned t4 = pk_shl2_u16(tt);                     // byte offsets of the two bins
                    atomicAdd((unsigned *)((char *)hist + (t4 & 0xffffu)), 1u);
                    atomicAdd((unsigned *)((char *)hist + (t4 >> 16)), 1u);
                }
                // keep the tiles apart: left alone the scheduler precomputes the LDS addresses of all 8 tiles
                // before it issues the first atomic
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!PA) {
            S += (long long)wave_sum(st);
            Q += (long long)wave_sum(stt & 0xffff) + ((long long)wave_sum((int)((unsigned)stt >> 16)) << 16);
            }
        
        };
        if constexpr (LONG) {
            // (three call sites instead of a choice inside the loop: with `wi == nwin - 2 ? y2 : y` in the loop body the
            // compiler kept both buffers and both sets of loads live across it -- 257 VGPRs)
            // KEEP: windows whose t' stay in registers for the second look.  The pA variant's statistics need ~80 registers
            // of their own: with two windows kept it spills them (107 VGPRs), with one it fits 128
            for (int wi = 0; wi < nwin - KEEP; wi++) first_window(y, wi);
            if constexpr (KEEP == 2) { if (nwin >= 2) first_window(y2, nwin - 2); }
            if constexpr (KEEP >= 1) { if (nwin >= 1) first_window(y, nwin - 1); }
        } else first_window(y, 0);

        // ---- exact integer totals (dump-bin entries taken out) ------------------------------------------------
        const long long D = (long long)__builtin_amdgcn_readfirstlane((int)hist[nbins]);   // dropped samples + slots past the end
        if (lane == 0) hist[nbins] = 0u;                                     // (LDS ops of a wave are in order)
        S -= D * nbins;
        Q -= D * nbins * nbins;
        const int n = tiles_total * 512 - (int)D;                            // samples that survived the filter

        // ---- median: rank select on the histogram (lane l owns bins [hb0, hb0 + 4 NQ)) ---------------------------
        // Two sweeps over the lane's own bins, four at a time (16-byte LDS reads), instead of holding all of them:
        // the samples already occupy 4 NT registers per lane.
        int local = 0;
#pragma unroll
        for (int j = 0; j < NQ; j++) {
            const uint4 q = *(const uint4 *)(hist + hb0 + 4 * j);
            local += (int)(q.x + q.y + q.z + q.w);
        }
        asm volatile("" ::: "memory");                     // (the second sweep re-reads: do not keep the bins live)

        sk_prep pr;
        pr.n = n; pr.flags = 0; pr.center = 0.0; pr.scale = 1.0; pr.top = 0.0; pr.bot = 0.0;
        int tlo = 0, width = 0;                                              // in band: (unsigned)(t - tlo) < width
        bool certified = true;
        // PA: a kept sample above the histogram window was counted as dropped -- the statistics are not the read's
        bool over = false;
        if constexpr (PA) {
            int mx = max((int)(short)(mxp & 0xffffu), (int)(short)(mxp >> 16));
            mx = (int)__builtin_amdgcn_readlane((int)wave_incl_max((unsigned)(mx + 32768)), 63) - 32768;
            over = pa_span > nbins && mx >= lo_r + 1 + nbins;
        }
        if (n == 0) {
            certified = !over;
            pr.flags = SK_FLAG_EMPTY;
            const double qnan = __builtin_nan("");
            pr.center = qnan; pr.scale = qnan; pr.top = qnan; pr.bot = qnan;
        } else {
            const int inc = wave_incl_scan(local);
            const int k1 = (n - 1) / 2, k2 = n / 2;
            const int pre = inc - local;
            int i1 = hb0, i2 = hb0, acc = pre;
#pragma unroll
            for (int j = 0; j < NQ; j++) {
                const uint4 q = *(const uint4 *)(hist + hb0 + 4 * j);
                const unsigned c4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    acc += (int)c4[i];
                    i1 += (acc <= k1) ? 1 : 0;
                    i2 += (acc <= k2) ? 1 : 0;
                }
            }
            const unsigned long long own1 = __ballot(local > 0 && k1 >= pre && k1 < pre + local);
            const unsigned long long own2 = __ballot(local > 0 && k2 >= pre && k2 < pre + local);
            const int b1 = __builtin_amdgcn_readlane(i1, own1 ? (int)__builtin_ctzll(own1) : 0);
            const int b2 = __builtin_amdgcn_readlane(i2, own2 ? (int)__builtin_ctzll(own2) : 0);
            const double eps8 = 8.0 * 1.1102230246251565e-16;
            if constexpr (PA) {
                // ---- the pA values' median, their std from exact centi-pA sums, thresholds certified in the c domain ----
                const int xa = lo_r + 1;
                const double median = (pa_centi(xa + b1, pa_off, pa_unit) / 100.0 + pa_centi(xa + b2, pa_off, pa_unit) / 100.0) / 2.0;
                unsigned long long V1 = 0, V2 = 0;                               // C1, C2
                {
                    unsigned long long c1 = 0, c2 = 0;
                    for (int j = 0; j < 4 * NQ; j++) {                           // bin 64 j + lane (conflict-free)
                        const unsigned k = hist[64 * j + lane];
                        if (__ballot(k != 0u) == 0ull) continue;
                        const unsigned cp = (unsigned)(int)(pa_centi(xa + 64 * j + lane, pa_off, pa_unit) - pa_c0);
                        const unsigned long long pk = (unsigned long long)k * (k ? cp : 0u);
                        c1 += pk; c2 += pk * (k ? cp : 0u);
                    }
#pragma unroll
                    for (int sft = 32; sft >= 1; sft >>= 1) { c1 += __shfl_xor(c1, sft); c2 += __shfl_xor(c2, sft); }
                    V1 = c1; V2 = c2;
                }
                unsigned long long ah, al, bh, bl;
                umul64wide((unsigned long long)n, V2, ah, al);                   // n C2
                umul64wide(V1, V1, bh, bl);                                      // C1^2
                const unsigned long long vl = al - bl, vh = ah - bh - (al < bl ? 1ull : 0ull);   // V = n C2 - C1^2 >= 0
                const double Vd = (double)vh * 18446744073709551616.0 + (double)vl;
                const double sd = sqrt(Vd) / (double)n / 100.0;
                const double spread = sd * a.std_scale;                          // segmenter.py:413-414
                const double top = median + spread, bot = median - spread;
                const double vmax = fmax(fabs(pa_c0), fabs(pa_centi(xa + nbins - 1, pa_off, pa_unit))) / 100.0;
                const double delta = a.delta_scale * eps8 * (fabs(spread) * (double)(n + 17) + fabs(median) +
                                                             vmax * (1.0 + fabs(a.std_scale)) * ((double)n / 8192.0 + 36.0));
                const double Dl = 100.0 * delta + 100.0 * eps8 * (fabs(top) + fabs(bot) + vmax);
                const double Tt = top * 100.0, Tb = bot * 100.0;
                // xt: the first window value not certainly below top; xl: the first certainly above bot
                const int xt = __builtin_amdgcn_readfirstlane(pa_first<true>(Tt - Dl, xa, nbins, pa_off, pa_unit, lane));
                const int xl = __builtin_amdgcn_readfirstlane(pa_first<false>(Tb + Dl, xa, nbins, pa_off, pa_unit, lane));
                const bool cert_t = xt == xa + nbins || pa_centi(xt, pa_off, pa_unit) >= Tt + Dl;
                const bool cert_b = xl == xa || pa_centi(xl - 1, pa_off, pa_unit) <= Tb - Dl;
                certified = cert_t && cert_b && !over;
                pr.center = median; pr.scale = sd; pr.top = top; pr.bot = bot;
                tlo = xl - xa;                                                   // first in-band bin
                width = max((xt - 1 - xa) - tlo + 1, 0);
            } else {
            const double median = (double)(b1 + b2 + 2 * (a.lo + 1)) * 0.5;  // exact (half-integer)

            // ---- thresholds from exact integers; certify ceil(top) / floor(bot) against numpy's rounding -------
            const long long V = (long long)n * Q - S * S;                    // n^2 var, exact (n <= 2^20: below 2^63)
            const double sd = sqrt((double)V) / (double)n;
            const double spread = sd * a.std_scale;                          // segmenter.py:413-414
            const double top = median + spread, bot = median - spread;
            const double delta = a.delta_scale * eps8 * (fabs(spread) * (double)(n + 17) + fabs(median));
            const double ct = ceil(top), fb = floor(bot);
            certified = (V == 0) || (ceil(top - delta) == ct && ceil(top + delta) == ct &&
                                     floor(bot - delta) == fb && floor(bot + delta) == fb);
            pr.center = median; pr.scale = sd; pr.top = top; pr.bot = bot;
            // integer band in bin coordinates (clamped around the histogram range first: everything stays small)
            const double ctc = fmin(fmax(ct, (double)a.lo - 4.0), (double)a.hi + 4.0);
            const double fbc = fmin(fmax(fb, (double)a.lo - 4.0), (double)a.hi + 4.0);
            tlo = max((int)fbc - a.lo, 0);                                   // first in-band bin: value floor(bot) + 1
            const int thi = min((int)ctc - a.lo - 2, nbins - 1);             // last in-band bin: value ceil(top) - 1
            width = max(thi - tlo + 1, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < NQ; j++) *(uint4 *)(hist + hb0 + 4 * j) = make_uint4(0u, 0u, 0u, 0u);   // (my reads are done)
        if (lane == 0) {
            a.prep[r] = pr;
            if (!certified) a.retry[1 + atomicAdd(&a.retry[0], 1)] = r;
        }

        // ---- second look: one "in band" and one "dropped" bit per raw sample ------------------------------------
        // Each lane has 8 consecutive samples of a tile -> one byte of each mask; v_dot2_u32_u16 with the bit
        // weights {1 << 2k, 1 << (2k + 1)} builds the bytes from 0/1 flags.  The bytes go through LDS so that
        // lane e can pick up entry e's 8 + 8 bytes and the wave stores a window's masks with ONE 16-byte store
        // per lane (1 KB contiguous).
        const unsigned tlop = (unsigned)__builtin_amdgcn_readfirstlane(tlo) * 0x10001u;
        const unsigned wm1p = (unsigned)((__builtin_amdgcn_readfirstlane(width) - 1) & 0xffff) * 0x10001u;
        // LONG: backwards -- the last window (in y) and the one before it (in y2) are classified out of the registers, the
        // others are fetched again
        uint2 vi = make_uint2(0u, 0u), vd = make_uint2(0u, 0u);
        auto classify_window = [&](unsigned (&yy)[NT][4], int wi, bool reload) __attribute__((always_inline)) {
            const int Mw = min(M - wi * WIN, WIN);
            const int ntiles = (Mw + 511) >> 9;
            if (reload) {
                load_window(yy, row + (int64_t)wi * WIN, Mw);
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    if (t >= ntiles) continue;
                    const int nvalid = min(max(Mw - (t * 512 + lane * 8), 0), 8);
#pragma unroll
                    for (int k = 0; k < 4; k++) yy[t][k] = image(yy[t][k], t, k, ntiles, nvalid);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (t >= ntiles) continue;
                unsigned drop8 = 0, out8 = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const unsigned wts = (1u << (2 * k)) | (2u << (2 * k + 16));
                    drop8 = udot2(pk_subsat_u16(yy[t][k], nbm1p), wts, drop8);                   // t' - (nbins - 1) is 0 or 1
                    const unsigned over = pk_subsat_u16(pk_sub_u16(yy[t][k], tlop), wm1p);       // > 0: outside the band
                    out8 = udot2(pk_min1_u16(over), wts, out8);
                }
                p_in[t * 64 + lane] = (unsigned char)((width > 0) ? ~out8 : 0u);
                p_dr[t * 64 + lane] = (unsigned char)drop8;
                __builtin_amdgcn_sched_barrier(0);
            }
            vi = *(const uint2 *)(p_in + 8 * lane); vd = *(const uint2 *)(p_dr + 8 * lane);
            if (lane < 8 * ntiles)
                a.mask2[(int64_t)r * a.row16 + wi * (8 * NT) + lane] = make_uint4(vi.x, vi.y, ~vd.x, ~vd.y);   // {in band, kept}
        };
        {
            if constexpr (LONG) {
                if constexpr (KEEP >= 1) { if (nwin >= 1) classify_window(y, nwin - 1, false); }
                if constexpr (KEEP == 2) { if (nwin >= 2) classify_window(y2, nwin - 2, false); }
                for (int wi = nwin - 1 - KEEP; wi >= 0; wi--) classify_window(y, wi, true);
            } else classify_window(y, 0, false);
            if (!LONG && a.hints) {
                // What k_seg_walk4 would otherwise read the whole mask row for (see there): lane e holds entry e, so
                // the stretches of quiet entries, their anchors and the drop counts are a few wave-wide operations.
                const int nent = (M + 63) >> 6;
                const bool has = lane < nent;
                const unsigned klo = has ? ~vd.x : 0u, khi = has ? ~vd.y : 0u;
                const unsigned olo = vi.x & klo, ohi = vi.y & khi;           // kept and in band
                const unsigned zlo = ~vi.x & klo, zhi = ~vi.y & khi;         // kept and out of band
                const bool quiet = has && __builtin_popcount(zlo) + __builtin_popcount(zhi) < a.e1;
                const int dcnt = has ? 64 - __builtin_popcount(klo) - __builtin_popcount(khi) : 0;
                const int dcum = wave_incl_scan(dcnt) - dcnt;                // samples dropped before my entry
                // bit i of (alo, ahi): the e1 samples before sample i of my entry are all kept and out of band
                const unsigned zprev = wave_shr1(zhi);
                unsigned alo = ~0u, ahi = ~0u;
                if (a.e1 == 6) {                                             // (the default) by doubling: 1, 2, 4, then 6 before
                    const unsigned t1lo = __builtin_amdgcn_alignbit(zlo, zprev, 31), t1hi = __builtin_amdgcn_alignbit(zhi, zlo, 31);
                    const unsigned t2lo = t1lo & __builtin_amdgcn_alignbit(t1lo, wave_shr1(t1hi), 31);
                    const unsigned t2hi = t1hi & __builtin_amdgcn_alignbit(t1hi, t1lo, 31);
                    const unsigned t2prev = wave_shr1(t2hi);
                    alo = t2lo & __builtin_amdgcn_alignbit(t2lo, t2prev, 30) & __builtin_amdgcn_alignbit(t2lo, t2prev, 28);
                    ahi = t2hi & __builtin_amdgcn_alignbit(t2hi, t2lo, 30) & __builtin_amdgcn_alignbit(t2hi, t2lo, 28);
                } else {
                    for (int j = 1; j <= a.e1; j++) {
                        alo &= __builtin_amdgcn_alignbit(zlo, zprev, 32 - j);
                        ahi &= __builtin_amdgcn_alignbit(zhi, zlo, 32 - j);
                    }
                }
                alo &= olo; ahi &= ohi;
                unsigned key = 0u;                                           // newest anchor of my entry << 16 | dcum
                if (alo | ahi)
                    key = ((unsigned)(64 * lane + (ahi ? 63 - __builtin_clz(ahi) : 31 - __builtin_clz(alo))) << 16) | (unsigned)dcum;
                const unsigned akey = wave_shr1(wave_shr1(wave_incl_max(key)));   // newest anchor in entries <= mine - 2 (none: sample 0)
                const unsigned long long Qm = __ballot(quiet);
                const bool rise = quiet && wave_shr1(quiet ? 1u : 0u) == 0u;
                const unsigned long long Rm = __ballot(rise);
                const int cnt = __builtin_popcountll(Rm);
                if (rise) {
                    const unsigned long long nq = ~Qm >> lane;               // (bit 0 clear: I am quiet)
                    const int kb = nq ? lane + (int)__builtin_ctzll(nq) - 1 : 63;
                    const int idx = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(Rm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)Rm, 0u));
                    if (idx < SEG_HINTS - 1)
                        a.hints[(int64_t)r * SEG_HINTS + 1 + idx] = (akey >> 16) | ((unsigned)kb << 12) | ((akey & 0xffffu) << 18);
                }
                if (lane == 0) a.hints[(int64_t)r * SEG_HINTS] = (certified && cnt <= SEG_HINTS - 1) ? (unsigned)cnt : SEG_HINT_NONE;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// reads of 4 097 .. 65 536 samples: a WORKGROUP per read, ONE look (round 6)
// ------------------------------------------------------------------------------------------------------
// k_seg_stats<.., LONG> gives a long read to one wavefront, which can hold 4 096 samples in registers and therefore
// looks at the read twice.  Counters on 50 000 x 20 000 samples (profiles/r06_pa_long_counters.txt): 4.0 GB fetched for
// 2.0 GB of samples at 6.1 TB/s of L2 misses -- the second look IS the limiter (vector ALU 58 % busy).  Here the four
// wavefronts of a workgroup share one read: wave w keeps windows w, w + 4, ... (up to KW of them, 32 VGPRs each) in its
// registers, the value histogram is one per workgroup (LDS atomics from all four waves), and after ONE barrier every
// wave derives the same statistics from it -- redundantly: a few hundred instructions against thousands per window --
// and classifies its own windows out of its registers.  The read is fetched once.  Same arithmetic, same certificate,
// same masks as k_seg_stats (the per-read part is the same code, parameterised by where the histogram lives).
template <int KT, bool PA>
__global__ __launch_bounds__(256, 2)
void k_seg_stats_wg(const SegStatArgs a)
{
    constexpr int NQ = 8, HBINS = 64 * 4 * NQ;
    __shared__ __align__(16) unsigned hist[HBINS];
    __shared__ __align__(16) unsigned char plane_all[4][2][64];
    __shared__ unsigned long long red[2];                  // not PA: sum t', sum t'^2;  PA: sum c', sum c'^2
    __shared__ int red_mx;                                 // PA: largest raw sample + 32 768
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char *p_in = plane_all[w][0], *p_dr = plane_all[w][1];
    const int hb0 = lane * 4 * NQ;

    int nbins = a.hi - a.lo - 1;
    int lo_r = a.lo;
    unsigned lo1p = (unsigned)((a.lo + 1) & 0xffff) * 0x10001u;
    unsigned nbp = (unsigned)nbins * 0x10001u, nbm1p = (unsigned)(nbins - 1) * 0x10001u;
    const int maxM = (int)min(a.stride, (int64_t)(4 * KT * 512));

    for (int i = threadIdx.x; i < HBINS; i += 256) hist[i] = 0u;
    if (threadIdx.x == 0) { red[0] = 0ull; red[1] = 0ull; red_mx = 0; }
    __syncthreads();

    // The read's 512-sample tiles are dealt round robin: wave w holds tiles w, w + 4, ... (1 KB each, 4 VGPRs a tile) --
    // whatever the read's length the four waves carry the same load to within one tile.
    unsigned y[KT][4];
    for (int r = blockIdx.x; r < a.nreads; r += gridDim.x) {
        const int M = __builtin_amdgcn_readfirstlane(min(max(a.len[r], 0), maxM));
        const int16_t *row = a.sig + (int64_t)r * a.stride;
        const int tiles_total = (M + 511) >> 9;

        // ---- all loads of my tiles first ----
#pragma unroll
        for (int k = 0; k < KT; k++) {
            const int g = w + 4 * k;
            uint4 q = make_uint4(0u, 0u, 0u, 0u);
            if (g * 512 + lane * 8 < M) q = *(const uint4 *)(row + (int64_t)g * 512 + lane * 8);
            y[k][0] = q.x; y[k][1] = q.y; y[k][2] = q.z; y[k][3] = q.w;
        }

        // ---- PA: this read's limits in the raw domain (every wave the same) ----
        bool pa_ok = true;
        double pa_off = 0.0, pa_unit = 1.0, pa_c0 = 0.0;
        int pa_span = 0;
        if constexpr (PA) {
            pa_off = a.cal[2 * r]; pa_unit = a.cal[2 * r + 1];
            pa_ok = pa_unit >= PA_UNIT_MIN && pa_unit <= PA_UNIT_MAX && fabs(pa_off) <= PA_OFFSET_MAX;
            int xa = 0, xb = -1;
            if (pa_ok) {
                xa = pa_first<true>(100.0 * (double)a.lo, -32768, 65536, pa_off, pa_unit, lane);
                xb = pa_first<false>(100.0 * (double)a.hi, -32768, 65536, pa_off, pa_unit, lane) - 1;
            }
            xa = __builtin_amdgcn_readfirstlane(xa); xb = __builtin_amdgcn_readfirstlane(xb);
            pa_span = max(xb - xa + 1, 0);
            nbins = min(pa_span, HBINS - 1);
            lo_r = xa - 1;
            lo1p = (unsigned)(xa & 0xffff) * 0x10001u;
            nbp = (unsigned)nbins * 0x10001u; nbm1p = (unsigned)((nbins - 1) & 0xffff) * 0x10001u;
            pa_c0 = pa_centi(xa, pa_off, pa_unit);
            if (nbins > 0 && !(pa_centi(xa + nbins - 1, pa_off, pa_unit) - pa_c0 < 262144.0)) { pa_ok = false; nbins = 0; }
        }
        const bool skip = PA && nbins == 0;                // nothing can be kept / calibration outside the plain range

        // ---- first look: t' per sample, histogram, sums / maximum ----
        if (!skip) {
            int st = 0;
            unsigned stt = 0u;                             // (8 x 2047^2 x 32 tiles < 2^31)
            unsigned mxp = 0x80008000u;
#pragma unroll
            for (int k = 0; k < KT; k++) {
                const int g = w + 4 * k;
                if (g >= tiles_total) continue;            // (wave-uniform)
                const bool last = g == tiles_total - 1;
                const int nvalid = min(max(M - (g * 512 + lane * 8), 0), 8);
#pragma unroll
                for (int k2 = 0; k2 < 4; k2++) {
                    const unsigned tail = !last ? 0xffffffffu : nvalid >= 2 * k2 + 2 ? 0xffffffffu : (nvalid == 2 * k2 + 1 ? 0xffffu : 0u);
                    if constexpr (PA) mxp = pk_max_i16(mxp, (y[k][k2] & tail) | (0x80008000u & ~tail));
                    unsigned tt = pk_min_u16(pk_sub_u16(y[k][k2], lo1p), nbp);
                    if (last) tt = (tt & tail) | (nbp & ~tail);
                    y[k][k2] = tt;
                    if constexpr (!PA) {
                        const s16x2 ts = __builtin_bit_cast(s16x2, tt);
                        st = __builtin_amdgcn_sdot2(ts, __builtin_bit_cast(s16x2, 0x10001u), st, false);
                        stt = (unsigned)__builtin_amdgcn_sdot2(ts, ts, (int)stt, false);
                    }
                    const unsigned t4 = pk_shl2_u16(tt);
                    atomicAdd((unsigned *)((char *)hist + (t4 & 0xffffu)), 1u);
                    atomicAdd((unsigned *)((char *)hist + (t4 >> 16)), 1u);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (PA) {
                int mx = max((int)(short)(mxp & 0xffffu), (int)(short)(mxp >> 16));
                mx = (int)__builtin_amdgcn_readlane((int)wave_incl_max((unsigned)(mx + 32768)), 63);
                if (lane == 0) atomicMax(&red_mx, mx);
            } else {
                unsigned long long S = (unsigned long long)(unsigned)st, Q = (unsigned long long)stt;
#pragma unroll
                for (int sft = 32; sft >= 1; sft >>= 1) { S += __shfl_xor(S, sft); Q += __shfl_xor(Q, sft); }
                if (lane == 0) { atomicAdd(&red[0], S); atomicAdd(&red[1], Q); }
            }
        }
        __syncthreads();                                   // the read's histogram (and sums / maximum) are complete

        // ---- D, n, the two middle values: every wave for itself (a few hundred instructions) ----
        long long D = 0;
        int n = 0, b1 = 0, b2 = 0;
        bool over = false;
        if (!skip) {
            D = (long long)__builtin_amdgcn_readfirstlane((int)hist[nbins]);   // dropped + slots past the read's end
            n = tiles_total * 512 - (int)D;
            if constexpr (PA) over = pa_span > nbins && (red_mx - 32768) >= lo_r + 1 + nbins;
            // (the dump bin sits ABOVE every kept value: a rank below n never reaches it, the sweeps need not skip it)
            int local = 0;
#pragma unroll
            for (int j = 0; j < NQ; j++) {
                const uint4 q = *(const uint4 *)(hist + hb0 + 4 * j);
                local += (int)(q.x + q.y + q.z + q.w);
            }
            asm volatile("" ::: "memory");
            if (n > 0) {
                const int inc = wave_incl_scan(local);
                const int k1 = (n - 1) / 2, k2 = n / 2;
                const int pre = inc - local;
                int i1 = hb0, i2 = hb0, acc = pre;
#pragma unroll
                for (int j = 0; j < NQ; j++) {
                    const uint4 q = *(const uint4 *)(hist + hb0 + 4 * j);
                    const unsigned c4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        acc += (int)c4[i];
                        i1 += (acc <= k1) ? 1 : 0;
                        i2 += (acc <= k2) ? 1 : 0;
                    }
                }
                const unsigned long long own1 = __ballot(local > 0 && k1 >= pre && k1 < pre + local);
                const unsigned long long own2 = __ballot(local > 0 && k2 >= pre && k2 < pre + local);
                b1 = __builtin_amdgcn_readlane(i1, own1 ? (int)__builtin_ctzll(own1) : 0);
                b2 = __builtin_amdgcn_readlane(i2, own2 ? (int)__builtin_ctzll(own2) : 0);
            }
            if constexpr (PA) {
                // the centi-pA sums off the histogram: wave w takes bins [512 w, 512 w + 512)
                if (n > 0) {
                    const int xa = lo_r + 1;
                    unsigned long long c1 = 0, c2 = 0;
                    for (int j = 8 * w; j < 8 * w + 8; j++) {
                        const int b = 64 * j + lane;
                        const unsigned k = b < nbins ? hist[b] : 0u;
                        if (__ballot(k != 0u) == 0ull) continue;
                        const unsigned cp = (unsigned)(int)(pa_centi(xa + b, pa_off, pa_unit) - pa_c0);
                        const unsigned long long pk = (unsigned long long)k * (k ? cp : 0u);
                        c1 += pk; c2 += pk * (k ? cp : 0u);
                    }
#pragma unroll
                    for (int sft = 32; sft >= 1; sft >>= 1) { c1 += __shfl_xor(c1, sft); c2 += __shfl_xor(c2, sft); }
                    if (lane == 0 && (c1 | c2)) { atomicAdd(&red[0], c1); atomicAdd(&red[1], c2); }
                }
            }
        }
        if constexpr (PA) __syncthreads();                 // the four partial sums are in

        sk_prep pr;
        pr.n = n; pr.flags = 0; pr.center = 0.0; pr.scale = 1.0; pr.top = 0.0; pr.bot = 0.0;
        int tlo = 0, width = 0;
        bool certified = true;
        if (skip || n == 0) {
            certified = skip ? pa_ok : !over;
            pr.n = 0;
            pr.flags = SK_FLAG_EMPTY;
            const double qnan = __builtin_nan("");
            pr.center = qnan; pr.scale = qnan; pr.top = qnan; pr.bot = qnan;
        } else {
            const double eps8 = 8.0 * 1.1102230246251565e-16;
            if constexpr (PA) {
                const int xa = lo_r + 1;
                const double median = (pa_centi(xa + b1, pa_off, pa_unit) / 100.0 + pa_centi(xa + b2, pa_off, pa_unit) / 100.0) / 2.0;
                const unsigned long long c1 = red[0], c2 = red[1];
                unsigned long long ah, al, bh, bl;
                umul64wide((unsigned long long)n, c2, ah, al);
                umul64wide(c1, c1, bh, bl);
                const unsigned long long vl = al - bl, vh = ah - bh - (al < bl ? 1ull : 0ull);
                const double Vd = (double)vh * 18446744073709551616.0 + (double)vl;
                const double sd = sqrt(Vd) / (double)n / 100.0;
                const double spread = sd * a.std_scale;
                const double top = median + spread, bot = median - spread;
                const double vmax = fmax(fabs(pa_c0), fabs(pa_centi(xa + nbins - 1, pa_off, pa_unit))) / 100.0;
                const double delta = a.delta_scale * eps8 * (fabs(spread) * (double)(n + 17) + fabs(median) +
                                                             vmax * (1.0 + fabs(a.std_scale)) * ((double)n / 8192.0 + 36.0));
                const double Dl = 100.0 * delta + 100.0 * eps8 * (fabs(top) + fabs(bot) + vmax);
                const double Tt = top * 100.0, Tb = bot * 100.0;
                const int xt = __builtin_amdgcn_readfirstlane(pa_first<true>(Tt - Dl, xa, nbins, pa_off, pa_unit, lane));
                const int xl = __builtin_amdgcn_readfirstlane(pa_first<false>(Tb + Dl, xa, nbins, pa_off, pa_unit, lane));
                const bool cert_t = xt == xa + nbins || pa_centi(xt, pa_off, pa_unit) >= Tt + Dl;
                const bool cert_b = xl == xa || pa_centi(xl - 1, pa_off, pa_unit) <= Tb - Dl;
                certified = cert_t && cert_b && !over;
                pr.center = median; pr.scale = sd; pr.top = top; pr.bot = bot;
                tlo = xl - xa;
                width = max((xt - 1 - xa) - tlo + 1, 0);
            } else {
                long long S = (long long)red[0], Q = (long long)red[1];
                S -= D * nbins;
                Q -= D * nbins * nbins;
                const double median = (double)(b1 + b2 + 2 * (a.lo + 1)) * 0.5;
                const long long V = (long long)n * Q - S * S;
                const double sd = sqrt((double)V) / (double)n;
                const double spread = sd * a.std_scale;
                const double top = median + spread, bot = median - spread;
                const double delta = a.delta_scale * eps8 * (fabs(spread) * (double)(n + 17) + fabs(median));
                const double ct = ceil(top), fb = floor(bot);
                certified = (V == 0) || (ceil(top - delta) == ct && ceil(top + delta) == ct &&
                                         floor(bot - delta) == fb && floor(bot + delta) == fb);
                pr.center = median; pr.scale = sd; pr.top = top; pr.bot = bot;
                const double ctc = fmin(fmax(ct, (double)a.lo - 4.0), (double)a.hi + 4.0);
                const double fbc = fmin(fmax(fb, (double)a.lo - 4.0), (double)a.hi + 4.0);
                tlo = max((int)fbc - a.lo, 0);
                const int thi = min((int)ctc - a.lo - 2, nbins - 1);
                width = max(thi - tlo + 1, 0);
            }
        }
        if (threadIdx.x == 0) {
            a.prep[r] = pr;
            if (!certified) a.retry[1 + atomicAdd(&a.retry[0], 1)] = r;
        }

        // ---- second look, out of the registers: the masks of my tiles (8 entries = 128 bytes a tile) ----
        const unsigned tlop = (unsigned)__builtin_amdgcn_readfirstlane(tlo) * 0x10001u;
        const unsigned wm1p = (unsigned)((__builtin_amdgcn_readfirstlane(width) - 1) & 0xffff) * 0x10001u;
        const bool nothing = skip || n == 0;
#pragma unroll
        for (int k = 0; k < KT; k++) {
            const int g = w + 4 * k;
            if (g >= tiles_total) continue;
            unsigned drop8 = 0, out8 = 0;
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++) {
                const unsigned wts = (1u << (2 * k2)) | (2u << (2 * k2 + 16));
                drop8 = udot2(pk_subsat_u16(y[k][k2], nbm1p), wts, drop8);
                const unsigned ov = pk_subsat_u16(pk_sub_u16(y[k][k2], tlop), wm1p);
                out8 = udot2(pk_min1_u16(ov), wts, out8);
            }
            p_in[lane] = (unsigned char)((width > 0 && !nothing) ? ~out8 : 0u);
            p_dr[lane] = (unsigned char)(nothing ? 0xffu : drop8);
            if (lane < 8) {
                const uint2 vi = *(const uint2 *)(p_in + 8 * lane), vd = *(const uint2 *)(p_dr + 8 * lane);
                a.mask2[(int64_t)r * a.row16 + g * 8 + lane] = make_uint4(vi.x, vi.y, ~vd.x, ~vd.y);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                   // everybody is done with this read's histogram
        for (int i = threadIdx.x; i < HBINS; i += 256) hist[i] = 0u;
        if (threadIdx.x == 0) { red[0] = 0ull; red[1] = 0ull; red_mx = 0; }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------
// the walk over {in band, kept} pairs: get_segs' state machine (segmenter.py:420-464), one lane per read
// ------------------------------------------------------------------------------------------------------
struct WalkParams {
    int error, corrector, window, seg_dist, first_len;   // first_len = ceil(window * stall_len)
};
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

// 32 samples, all valid, corrector test dead (error < corrector): the straight-line step of sk_segment.hip
__device__ __forceinline__ void walk_fast32(WalkState &st, unsigned bits, int i0, const WalkParams &p,
                                            int thr_first, int32_t *my, int max_segs)
{
    unsigned prev = (unsigned)st.prev, err = (unsigned)st.err, perr = (unsigned)st.prev_err, c = (unsigned)st.c;
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

// general step: any parameters, samples at index >= n ignored; keeps `start` and `w`
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

// FAST: error < corrector (the corrector test can never fire, see sk_segment.hip) and positive thresholds.
// Each lane streams its read's entries; the kept bits of an entry are squeezed together (the filter drops a
// handful of samples per read, so the squeeze loop runs a few times per READ) and appended to a bit queue;
// whenever the queue holds 64 bits they go through the state machine.
template <bool FAST>
__global__ __launch_bounds__(64)
void k_seg_walk2(const uint4 *__restrict__ mask2, int row16, const int32_t *__restrict__ len, int64_t stride,
                 int nreads, WalkParams p, int32_t *__restrict__ segs, int32_t *__restrict__ nsegs, int max_segs)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < nreads;
    const int M = live ? min(max(len[r], 0), (int)min(stride, (int64_t)row16 * 64)) : 0;
    const uint4 *mrow = mask2 + (int64_t)(live ? r : 0) * row16;
    int32_t *my = segs + (int64_t)(live ? r : 0) * 2 * max_segs;

    WalkState st;
    st.prev = 0; st.err = 0; st.prev_err = 0; st.c = 0;
    st.w = FAST ? 0x7fffffff : p.corrector;       // segmenter.py:424 -- never reset inside a read
    st.start = 0; st.nseg = 0; st.last_end = 0;
    const int thr_first = min(p.window, p.first_len);

    const int nent = (M + 63) >> 6;               // my entries
    int nmax = nent;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, __shfl_xor(nmax, d));

    unsigned long long qlo = 0ull, qhi = 0ull;    // bit queue: `fill` bits, oldest at bit 0 of qlo
    int fill = 0, done = 0;                       // done: filtered samples already through the state machine
    uint4 next = (nent > 0) ? mrow[0] : make_uint4(0u, 0u, 0u, 0u);
    for (int e = 0; e < nmax; e++) {
        const uint4 cur = next;
        if (e + 1 < nent) next = mrow[e + 1];     // prefetch
        if (e < nent) {
            unsigned long long inb = ((unsigned long long)cur.y << 32) | cur.x;
            unsigned long long kp = ((unsigned long long)cur.w << 32) | cur.z;
            int cnt = 64;
            if (kp == 0ull) { cnt = 0; kp = ~0ull; }
            const int hz = __builtin_clzll(kp);   // dropped samples at the top of the entry (the read's tail) go at once
            if (hz > 0) { cnt -= hz; kp |= ~0ull << (64 - hz); }
            while (kp != ~0ull) {                 // delete the lowest dropped sample's bit, close the gap
                const int pos = __builtin_ctzll(~kp);
                const unsigned long long below = (1ull << pos) - 1ull;
                inb = (inb & below) | ((inb >> 1) & ~below);
                kp = (kp & below) | ((kp >> 1) & ~below) | (1ull << 63);
                cnt--;
            }
            // (the `cnt` surviving bits are now bits 0 .. cnt-1 of inb; bits above are garbage -> masked)
            if (cnt < 64) inb &= (1ull << cnt) - 1ull;
            qlo |= (fill < 64) ? inb << fill : 0ull;
            qhi |= (fill > 0) ? inb >> (64 - fill) : 0ull;
            fill += cnt;
        }
        if (fill >= 64) {
            if (FAST) {
                walk_fast32(st, (unsigned)qlo, done, p, thr_first, my, max_segs);
                walk_fast32(st, (unsigned)(qlo >> 32), done + 32, p, thr_first, my, max_segs);
            } else {
                walk_general32(st, (unsigned)qlo, done, 0x7fffffff, p, my, max_segs);
                walk_general32(st, (unsigned)(qlo >> 32), done + 32, 0x7fffffff, p, my, max_segs);
            }
            done += 64; fill -= 64;
            qlo = qhi; qhi = 0ull;
        }
    }
    // the last fill (< 64) bits
    if (FAST) st.start = done - st.c;             // hand over to the general step (which tracks `start`)
    if (fill > 0) {
        const int n = done + fill;
        walk_general32(st, (unsigned)qlo, done, n, p, my, max_segs);
        walk_general32(st, (unsigned)(qlo >> 32), done + 32, n, p, my, max_segs);
    }
    if (live) nsegs[r] = st.nseg;                 // a segment still open at EOF is dropped (:466)
}

// ------------------------------------------------------------------------------------------------------
// the same walk by RUNS instead of by samples (error < corrector, positive thresholds: what k_seg_walk2<true> covers)
// ------------------------------------------------------------------------------------------------------
// With the corrector test dead, get_segs' per-sample state has a closed form.  A candidate segment ("run") opens at
// an in-band sample s; every in-band sample and the first E = max(error, 0) out-of-band ones extend it (:431-447);
// the (E + 1)-th out-of-band sample z closes it (:448 / :458) with
//     c = z - s                      (every sample of [s, z) was counted)
//     end = z - prev_err = l + 1     (l = last in-band sample before z: prev_err counts the out-of-band tail)
// and the scan resumes behind z.  So a lane hops from run to run with bit scans (find-first-set, popcount,
// clear-lowest-set) on 32-bit pieces of its read's in-band mask instead of stepping through every sample:
// ~200 runs instead of 4 000 samples per read, 3-4x fewer vector instructions (the kernel is issue-bound).
// An entry's in-band bits with the dropped samples' bits deleted (the filter drops a handful of samples per read, so
// the loop runs a few times per READ); returns the samples kept.
__device__ __forceinline__ int squeeze_entry(unsigned long long &inb, unsigned long long kp)
{
    int cnt = 64;
    if (kp != ~0ull) {
        if (kp == 0ull) { cnt = 0; kp = ~0ull; }
        const int hz = __builtin_clzll(kp);                    // dropped samples at the top of the entry go at once
        if (hz > 0) { cnt -= hz; kp |= ~0ull << (64 - hz); }
        while (kp != ~0ull) {                                  // delete the lowest dropped sample's bit, close the gap
            const int pos = __builtin_ctzll(~kp);
            const unsigned long long below = (1ull << pos) - 1ull;
            inb = (inb & below) | ((inb >> 1) & ~below);
            kp = (kp & below) | ((kp >> 1) & ~below) | (1ull << 63);
            cnt--;
        }
        if (cnt < 64) inb &= (1ull << cnt) - 1ull;
    }
    return cnt;
}

struct RunState {
    int in_run, zl, start, last1;     // zl: out-of-band samples the open run still needs to close
    int nseg, last_end;
    unsigned thr;                     // report threshold: min(window, first_len) until the first segment (:448)
};

__device__ __forceinline__ void run_report(RunState &st, int start, int end, const WalkParams &p, int32_t *my, int max_segs)
{
    if (st.nseg > 0 && start - st.last_end < p.seg_dist) {                         // :451 merge
        if (st.nseg <= max_segs) my[2 * (st.nseg - 1) + 1] = end;
    } else {
        if (st.nseg < max_segs) { my[2 * st.nseg] = start; my[2 * st.nseg + 1] = end; }
        st.nseg++;
    }
    st.last_end = end;
    st.thr = (unsigned)p.window;
}

// 32 samples W (bit b = filtered sample base + b in band), all valid
__device__ __forceinline__ void run_word32(RunState &st, unsigned W, int base, int E1, const WalkParams &p,
                                           int32_t *my, int max_segs)
{
    int pos = 0;
    while (pos < 32) {
        if (!st.in_run) {
            const unsigned m = W >> pos;
            if (m == 0u) break;                                   // nothing opens in the rest of the word
            pos += __builtin_ctz(m);
            st.in_run = 1; st.start = base + pos; st.zl = E1; st.last1 = st.start;
        }
        const unsigned ones = W >> pos;                           // (pos < 32)
        unsigned Z = ~W >> pos;                                   // out-of-band samples at >= pos
        const int nz = __builtin_popcount(Z);
        if (nz < st.zl) {                                         // the run outlives this word
            st.zl -= nz;
            if (ones) st.last1 = base + 31 - __builtin_clz(W);
            break;
        }
        for (int i = 1; i < st.zl; i++) Z &= Z - 1u;              // the zl-th out-of-band sample closes the run
        const int q = __builtin_ctz(Z);
        const unsigned before = ones & ((1u << q) - 1u);          // in-band samples of [pos, z)
        if (before) st.last1 = base + pos + 31 - __builtin_clz(before);
        const int z = base + pos + q;
        if ((unsigned)(z - st.start) >= st.thr) run_report(st, st.start, st.last1 + 1, p, my, max_segs);   // :448-454
        st.in_run = 0;
        pos += q + 1;
    }
}

// one lane's read, the wavefront's lanes on the same entry (nmax: the longest read's entries); returns the segment count
__device__ __forceinline__ int walk_sync_read(const uint4 *__restrict__ mrow, int nent, int nmax, const WalkParams &p,
                                              int32_t *my, int max_segs)
{
    RunState st;
    st.in_run = 0; st.zl = 0; st.start = 0; st.last1 = 0; st.nseg = 0; st.last_end = 0;
    st.thr = (unsigned)min(p.window, p.first_len);
    const int E1 = max(p.error, 0) + 1;

    unsigned long long qlo = 0ull, qhi = 0ull;    // bit queue: `fill` bits, oldest at bit 0 of qlo
    int fill = 0, done = 0;                       // done: filtered samples already walked
    for (int e0 = 0; e0 < nmax; e0 += 8) {
        // eight entries = one 128-byte line of my row per visit (the rows of a wave's 64 lanes are 1 KB apart:
        // fetching an entry at a time brings every line in from HBM three times over)
        uint4 buf[8];
#pragma unroll
        for (int k = 0; k < 8; k++) buf[k] = (e0 + k < nent) ? mrow[e0 + k] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (e0 + k >= nmax) break;            // (wave-uniform)
            if (e0 + k < nent) {
                unsigned long long inb = ((unsigned long long)buf[k].y << 32) | buf[k].x;
                int cnt = squeeze_entry(inb, ((unsigned long long)buf[k].w << 32) | buf[k].z);
                if (fill == 0) { qlo = inb; qhi = 0ull; }
                else { qlo |= inb << fill; qhi = inb >> (64 - fill); }
                fill += cnt;
            }
            if (fill >= 64) {
                run_word32(st, (unsigned)qlo, done, E1, p, my, max_segs);
                run_word32(st, (unsigned)(qlo >> 32), done + 32, E1, p, my, max_segs);
                done += 64; fill -= 64;
                qlo = qhi; qhi = 0ull;
            }
        }
    }
    // the last fill (< 64) samples: slots past the read's end count as in band -- they can extend an open run
    // (which is dropped at EOF either way, :466) but never close one, and a run opening there never closes
    if (fill > 0) {
        const unsigned long long w = qlo | (~0ull << fill);
        run_word32(st, (unsigned)w, done, E1, p, my, max_segs);
        run_word32(st, (unsigned)(w >> 32), done + 32, E1, p, my, max_segs);
    }
    return st.nseg;
}

__global__ __launch_bounds__(64)
void k_seg_walk3(const uint4 *__restrict__ mask2, int row16, const int32_t *__restrict__ len, int64_t stride,
                 int nreads, WalkParams p, int32_t *__restrict__ segs, int32_t *__restrict__ nsegs, int max_segs)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < nreads;
    const int M = live ? min(max(len[r], 0), (int)min(stride, (int64_t)row16 * 64)) : 0;
    const uint4 *mrow = mask2 + (int64_t)(live ? r : 0) * row16;
    int32_t *my = segs + (int64_t)(live ? r : 0) * 2 * max_segs;
    const int nent = (M + 63) >> 6;               // my entries
    int nmax = nent;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, __shfl_xor(nmax, d));
    const int ns = walk_sync_read(mrow, nent, nmax, p, my, max_segs);
    if (live) nsegs[r] = ns;
}

// ------------------------------------------------------------------------------------------------------
// the walk by runs with every lane at ITS OWN position, jumping between the places that can matter
// (same preconditions as k_seg_walk3, plus window >= 127 and error <= 31)
// ------------------------------------------------------------------------------------------------------
// k_seg_walk3 keeps its 64 lanes on the same 32-sample word (a lane leaves run_word32's loop after (runs closing in
// the word) + 1 trips, the wavefront after the slowest lane's) and squeezes the dropped samples' bits out of every
// entry to get there.  This kernel stays in RAW sample coordinates -- a dropped sample is neither in band nor out of
// band, it is skipped and counted -- and lets every lane hop from run to run on a 64-sample window taken at its own
// position out of the two mask entries it holds in registers: one trip per run, plus one per 64 samples of a long run.
//
// Jumps.  Once a read has its first segment only runs of c >= window samples count (:448).  Such a run [s, z) spans
// >= 127 raw samples, so it covers the aligned entry k1 = ceil(s / 64) completely -- an entry with <= E out-of-band
// samples, "quiet".  Somebody who sees every entry of the read's mask row once -- the statistics kernel, which holds
// entry e in lane e when it stores the row (HINTS: k_seg_stats above, 8 dwords per read), or else a first pass of this
// kernel over the row (uniform across the wavefront, ~55 instructions per entry; float64 reads, reads beyond 4 096
// samples, SK_WALK_OWNPASS) -- notes every stretch [ka, kb] of quiet entries together with an ANCHOR: the newest position <= 64 (ka - 1) at which the state
// of the chain is known without walking it -- an in-band sample whose E + 1 raw predecessors are all kept and out of
// band opens a run whatever came before (a run open there has closed inside them, the scan was idle behind it), or
// sample 0 -- and the number of samples dropped before the anchor's entry.  Every run of interest that opens in a stretch
// (s in (64 (ka - 1), 64 kb]) opens at or behind the stretch's anchor, and no anchor lies inside a run with E
// out-of-band samples.  So an idle lane whose read has a segment drops the stretches that end before its position and
// continues at max(position, next stretch's anchor); with no stretch ahead it is done.  A read with more stretches than
// the list holds (7 with hints, 8 without), or whose masks the numpy-order redo rewrote, is walked without jumps.
// What the kernel's time is made of is the latency of a lane's dependent 16-byte loads (rows 1 KB apart): with hints
// the row goes through LDS a 128-byte line at a time and the first line leaves with the hints in one round trip
// (tests/test_walk_model.py is this scheme sample by sample against the oracle).
// Positions: start = z_f - c and end = z_f - prev_err (:449), z_f = z - (samples dropped before z), c = (z - s) -
// (samples dropped inside the run), prev_err = the out-of-band samples since the run's last in-band one.
constexpr int W4_ITEMS = 8;            // stretches noted per read

template <int E1C, bool HINTS>         // E + 1 at compile time (0: at run time); HINTS: the statistics kernel's list
__global__ __launch_bounds__(64)
void k_seg_walk4(const uint4 *__restrict__ mask2, int row16, const int32_t *__restrict__ len, int64_t stride,
                 int nreads, WalkParams p, int32_t *__restrict__ segs, int32_t *__restrict__ nsegs, int max_segs,
                 int use_jumps, const unsigned *__restrict__ hints)
{
    // LDS: with hints, one 128-byte line (8 entries) of every lane's mask row -- a lane asks for its row 16 bytes at a
    // time, the rows are 1 KB apart, and with 32 wavefronts per CU in flight a line did not survive in the L2 until
    // its next entry was wanted: 1.75 GB fetched per 1 M reads for 1 GB of rows.  Staged, a line is fetched once; and
    // the 8 KB leave 20 wavefronts per CU, which is where this kernel runs best anyway (0.338 ms at 32 per CU, 0.318
    // at 20, 0.456 at 10 with nothing else changed).  Without hints: the list of stretches of the kernel's own pass.
    __shared__ uint4 w4_lds[HINTS ? 8 * 64 : (2 * W4_ITEMS * 64) / 4];
    const int lane = threadIdx.x;
    unsigned *items = (unsigned *)w4_lds + lane;               // stretch i: items[128 i] = anchor | kb << 16, items[128 i + 64] = drops
    uint4 *line = w4_lds + lane;                               // entry k of my line at line[64 k]
    int have = -1;                                             // which line of my row that is
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < nreads;
    const int M = live ? min(max(len[r], 0), (int)min(stride, (int64_t)row16 * 64)) : 0;
    const uint4 *mrow = mask2 + (int64_t)(live ? r : 0) * row16;
    int32_t *my = segs + (int64_t)(live ? r : 0) * 2 * max_segs;
    const int E1 = E1C ? E1C : max(p.error, 0) + 1;

    const int nent = (M + 63) >> 6;
    int nmax = nent;
    if (!HINTS) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, __shfl_xor(nmax, d));
    }

    // ---- first pass: quiet stretches, anchors ----
    int nitems = 0;
    const unsigned *myhints = hints + (int64_t)(live ? r : 0) * SEG_HINTS;   // (the statistics kernel's, when it wrote any)
    uint4 ha = make_uint4(0u, 0u, 0u, 0u), hb = ha;            // (HINTS) my read's hints, all eight dwords
    if (HINTS) {
        // everything the walk will ask for first, in one round trip: the hints and the first line of the row (a lane's
        // loads depend on each other from here on -- the latency of those chains is what this kernel's time is made of)
        if (live) {                                            // (not "if the read has samples": that would wait for len[r])
            ha = ((const uint4 *)myhints)[0]; hb = ((const uint4 *)myhints)[1];
#pragma unroll
            for (int j = 0; j < 8; j++) line[64 * j] = mrow[j];
            have = 0;
        }
        nitems = (int)ha.x;                                    // SEG_HINT_NONE: more than the list holds, no jumps
    } else if (use_jumps) {
        int dcum = 0, anchor = 0, anchor_d = 0, anchor_prev = 0, anchor_prev_d = 0, quiet_prev = 0;
        unsigned zprev = 0u, cur_anchor = 0u;  // zprev: the previous entry's upper 32 out-of-band bits (before the read: none)
        unsigned t1prev = 0u, t2prev = 0u;
        for (int e0 = 0; e0 < nmax; e0 += 8) {
            // eight entries = one 128-byte line of my row per visit (the rows of a wave's 64 lanes are 1 KB apart)
            uint4 buf[8];
#pragma unroll
            for (int k = 0; k < 8; k++) buf[k] = (e0 + k < nent) ? mrow[e0 + k] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (e0 + k >= nmax) break;                     // (wave-uniform)
                const unsigned olo = buf[k].x & buf[k].z, ohi = buf[k].y & buf[k].w;       // kept and in band
                const unsigned zlo = ~buf[k].x & buf[k].z, zhi = ~buf[k].y & buf[k].w;     // kept and out of band
                const int quiet = (e0 + k < nent) && __builtin_popcount(zlo) + __builtin_popcount(zhi) < E1;
                if (quiet) {
                    if (!quiet_prev) {
                        cur_anchor = (unsigned)anchor_prev;
                        if (nitems < W4_ITEMS) items[128 * nitems + 64] = (unsigned)anchor_prev_d;
                        nitems++;
                    }
                    if (nitems <= W4_ITEMS) items[128 * (nitems - 1)] = cur_anchor | ((unsigned)(e0 + k) << 16);
                }
                quiet_prev = quiet;
                anchor_prev = anchor; anchor_prev_d = anchor_d;
                // bit i of (alo, ahi): the E1 samples before sample i of this entry are all kept and out of band
                unsigned alo, ahi;
                if (E1C == 6) {
                    // by doubling: t1 = the sample before, t2 = the two before, then four, then six (t1 / t2 of the
                    // previous entry's upper half ride along)
                    const unsigned t1lo = __builtin_amdgcn_alignbit(zlo, zprev, 31), t1hi = __builtin_amdgcn_alignbit(zhi, zlo, 31);
                    const unsigned t2lo = t1lo & __builtin_amdgcn_alignbit(t1lo, t1prev, 31), t2hi = t1hi & __builtin_amdgcn_alignbit(t1hi, t1lo, 31);
                    const unsigned t4lo = t2lo & __builtin_amdgcn_alignbit(t2lo, t2prev, 30), t4hi = t2hi & __builtin_amdgcn_alignbit(t2hi, t2lo, 30);
                    alo = t4lo & __builtin_amdgcn_alignbit(t2lo, t2prev, 28);
                    ahi = t4hi & __builtin_amdgcn_alignbit(t2hi, t2lo, 28);
                    t1prev = t1hi; t2prev = t2hi;
                } else {
                    alo = ~0u; ahi = ~0u;
                    for (int j = 1; j <= E1; j++) {
                        alo &= __builtin_amdgcn_alignbit(zlo, zprev, 32 - j);
                        ahi &= __builtin_amdgcn_alignbit(zhi, zlo, 32 - j);
                    }
                }
                alo &= olo; ahi &= ohi;
                if (alo | ahi) {
                    anchor = 64 * (e0 + k) + (ahi ? 63 - __builtin_clz(ahi) : 31 - __builtin_clz(alo));
                    anchor_d = dcum;
                }
                zprev = zhi;
                dcum += 64 - __builtin_popcount(buf[k].z) - __builtin_popcount(buf[k].w);
            }
        }
    }
    const bool jumps = use_jumps && nitems <= (HINTS ? SEG_HINTS - 1 : W4_ITEMS);
    // stretch i: its anchor, last entry, samples dropped before the anchor's entry
    auto stretch = [&](int i, int &a, int &kb, int &d) __attribute__((always_inline)) {
        if (HINTS) {
            const unsigned h = i == 0 ? ha.y : i == 1 ? ha.z : i == 2 ? ha.w : i == 3 ? hb.x : i == 4 ? hb.y : i == 5 ? hb.z : hb.w;
            a = (int)(h & 0xfffu); kb = (int)((h >> 12) & 63u); d = (int)(h >> 18);
        }
        else { const unsigned h = items[128 * i]; a = (int)(h & 0xffffu); kb = (int)(h >> 16); d = (int)items[128 * i + 64]; }
    };
    int item = 0;

    // ---- the walk ----
    RunState st;                                               // (last1 holds prev_err here)
    st.in_run = 0; st.zl = 0; st.start = 0; st.last1 = 0; st.nseg = 0; st.last_end = 0;
    st.thr = (unsigned)min(p.window, p.first_len);
    int pos = 0, ce = -2, dcur = 0, jump_d = 0, rundrops = 0;
    unsigned long long Oc = 0ull, Zc = 0ull, On = 0ull, Zn = 0ull;
    while (true) {
        bool act = pos < M;
        if (act && jumps && !st.in_run && st.nseg > 0) {
            int a = 0, kb = 0, d = 0;
            bool found = false;
            while (item < nitems) {
                stretch(item, a, kb, d);
                if (kb * 64 + 63 >= pos) { found = true; break; }
                item++;
            }
            if (found) {
                if (a > pos) { pos = a; jump_d = d; ce = -2; }
            } else {
                pos = M; act = false;
            }
        }
        if (__builtin_amdgcn_ballot_w64(act) == 0ull) break;
        if (!act) continue;
        const int e = pos >> 6;
        if (e != ce) {                                         // the two entries my window lies in
            auto entry = [&](int k) __attribute__((always_inline)) -> uint4 {
                if (k >= nent) return make_uint4(0u, 0u, 0u, 0u);
                if (!HINTS) return mrow[k];
                if ((k >> 3) != have) {                        // my row's line with entry k, all of it, once
                    have = k >> 3;
#pragma unroll
                    for (int j = 0; j < 8; j++) line[64 * j] = mrow[8 * have + j];
                }
                return line[64 * (k & 7)];
            };
            if (e == ce + 1) { dcur += 64 - __builtin_popcountll(Oc | Zc); Oc = On; Zc = Zn; }
            else {
                const uint4 v = entry(e);                      // (e < nent: pos < M)
                Oc = ((unsigned long long)(v.y & v.w) << 32) | (v.x & v.z);
                Zc = ((unsigned long long)(~v.y & v.w) << 32) | (~v.x & v.z);
                dcur = jump_d;
            }
            const uint4 v = entry(e + 1);
            On = ((unsigned long long)(v.y & v.w) << 32) | (v.x & v.z);
            Zn = ((unsigned long long)(~v.y & v.w) << 32) | (~v.x & v.z);
            ce = e;
        }
        const int sh = pos & 63;
        unsigned long long Ow = (Oc >> sh) | ((On << 1) << (63 - sh));
        unsigned long long Zw = (Zc >> sh) | ((Zn << 1) << (63 - sh));
        int V = 64;
        if (!st.in_run) {
            if (Ow == 0ull) { pos += 64; continue; }
            const int t = __builtin_ctzll(Ow);
            pos += t; Ow >>= t; Zw >>= t; V -= t;
            st.in_run = 1; st.start = pos; st.zl = E1; st.last1 = 0; rundrops = 0;
        }
        // One piece of the run: up to its closing sample (the zl-th out-of-band one of the window) or, when the
        // run outlives the window, all V samples of it.  Both cases share the bookkeeping below.
        const unsigned long long Z = Zw & ((2ull << (V - 1)) - 1ull);
        const int nz = __builtin_popcountll(Z);
        const bool closes = nz >= st.zl;
        unsigned long long Zk = Z;
        const int kth = closes ? st.zl : 1;
        for (int i = 1; i < kth; i++) Zk &= Zk - 1ull;
        const int L = closes ? (int)__builtin_ctzll(Zk) : V;                  // samples of the piece (without the closing one)
        const int zeros = closes ? st.zl - 1 : nz;                           // out-of-band samples in it
        const unsigned long long below = (L == 64) ? ~0ull : ((1ull << L) - 1ull);
        const unsigned long long ones = Ow & below;
        rundrops += L - zeros - __builtin_popcountll(ones);                  // neither in band nor out of band: dropped
        if (ones) st.last1 = zeros - __builtin_popcountll(Z & below & ((2ull << (63 - __builtin_clzll(ones))) - 1ull));
        else st.last1 += zeros;                                              // prev_err (:445, :449)
        st.zl -= zeros;
        if (closes) {
            const int z = pos + L;
            const int c = (z - st.start) - rundrops;
            if ((unsigned)c >= st.thr) {                       // :448-454, in filtered coordinates
                const int off = z - 64 * ce;                   // z lies in one of my two entries
                const unsigned long long dc = ~(Oc | Zc), dn = ~(On | Zn);
                const int dz = dcur + (off >= 64 ? __builtin_popcountll(dc) + __builtin_popcountll(dn & ((1ull << (off - 64)) - 1ull))
                                                 : __builtin_popcountll(dc & ((1ull << off) - 1ull)));
                const int zf = z - dz;
                run_report(st, zf - c, zf - st.last1, p, my, max_segs);
            }
            st.in_run = 0;
            pos = z + 1;
        } else pos += V;
    }
    if (live) nsegs[r] = st.nseg;
}


// ------------------------------------------------------------------------------------------------------
// the walk by runs with a WAVEFRONT per read: 64 lanes on 64 pieces of the same read (round 6)
// ------------------------------------------------------------------------------------------------------
// k_seg_walk4 gives every read one lane.  That is the right shape for a million short reads; for 25 000 reads of 37 000
// samples it leaves 390 wavefronts on 1 024 SIMDs, each lane walking 580 mask entries through a chain of dependent
// loads and ~60 dependent instructions per run: 1.4 ms, twice the statistics kernel in front of it (and the same for
// 10 000 reads of 4 000 samples, BASELINE config 2: 157 wavefronts).  get_segs' scan (segmenter.py:420-464) is
// sequential, but its state is KNOWN at every anchor -- an in-band sample whose E + 1 raw predecessors are all kept and
// out of band: whatever run was open has been closed inside them, the scan is idle, the sample opens a run (the same
// fact k_seg_walk4's jumps rest on).  So the read is cut into 64 pieces of ceil(entries / 64) mask entries; lane j finds
// the first anchor at or behind the start of piece j (lane 0: sample 0), walks by runs from there and stops when the
// next run would open at or behind the next lane's anchor -- where, by the anchor's definition, it is idle.  What the
// pieces cannot know is whether the read already HAS a segment (:448: the first one needs c >= window * stall_len only)
// and where the last one ended (:451, the merge): each lane keeps its first run with c >= min(window, first_len) and
// every run with c >= window (at most 8: a piece of a staged row is <= 1 024 samples), and the wave combines them in
// order -- the first lane that has anything contributes its first run, everything later contributes its long runs,
// merged by seg_dist.  A read with more long runs in one piece than the list holds (rows beyond 65 536 samples) is
// walked by lane 0 alone, sequentially.  Rows of up to 1 024 entries are staged in LDS with coalesced 16-byte loads.
constexpr int WL_NB = 8;               // long runs noted per lane

template <int E1C, bool STAGED>        // STAGED: the row fits the LDS staging area (rows of up to lds_entries mask entries)
__global__ __launch_bounds__(64)
void k_seg_walkL(const uint4 *__restrict__ mask2, int row16, const int32_t *__restrict__ len, int64_t stride,
                 int nreads, WalkParams p, int32_t *__restrict__ segs, int32_t *__restrict__ nsegs, int max_segs,
                 int lds_entries)
{
    extern __shared__ __align__(16) uint4 wl_lds[];            // [lds_entries] the read's mask row, then [WL_NB][64] int2
    const int lane = threadIdx.x;
    const int r = blockIdx.x;
    const int M = min(max(len[r], 0), (int)min(stride, (int64_t)row16 * 64));
    const uint4 *mrow = mask2 + (int64_t)r * row16;
    int32_t *my = segs + (int64_t)r * 2 * max_segs;
    int2 *cand = (int2 *)(wl_lds + lds_entries) + lane;        // my i-th long run at cand[64 i]
    const int E1 = E1C ? E1C : max(p.error, 0) + 1;
    const int nent = (M + 63) >> 6;
    if (nent == 0) { if (lane == 0) nsegs[r] = 0; return; }
    if constexpr (STAGED) {
        for (int k = lane; k < nent; k += 64) wl_lds[k] = mrow[k];
        __syncthreads();                                       // (one wavefront: orders the LDS writes before the reads)
    }
    auto entry = [&](int k) __attribute__((always_inline)) -> uint4 {
        if (k < 0 || k >= nent) return make_uint4(0u, 0u, 0u, 0u);
        if constexpr (STAGED) return wl_lds[k];
        else return mrow[k];
    };

    // ---- my piece: samples dropped in it, its first anchor ----
    const int chunk = (nent + 63) >> 6;
    const int e_lo = lane * chunk, e_hi = min(e_lo + chunk, nent);
    int dcnt = 0, start = 0x7fffffff, dloc = 0;
    if (lane == 0) start = 0;
    {
        unsigned zprev = 0u;
        if (e_lo > 0 && e_lo < nent) { const uint4 v = entry(e_lo - 1); zprev = ~v.y & v.w; }
        for (int e = e_lo; e < e_hi; e++) {
            const uint4 v = entry(e);
            const unsigned zlo = ~v.x & v.z, zhi = ~v.y & v.w;               // kept and out of band
            if (start == 0x7fffffff) {
                unsigned alo = v.x & v.z, ahi = v.y & v.w;                   // kept and in band ...
                for (int j = 1; j <= E1; j++) {                              // ... behind E1 kept out-of-band samples
                    alo &= __builtin_amdgcn_alignbit(zlo, zprev, 32 - j);
                    ahi &= __builtin_amdgcn_alignbit(zhi, zlo, 32 - j);
                }
                if (alo | ahi) { start = 64 * e + (alo ? __builtin_ctz(alo) : 32 + __builtin_ctz(ahi)); dloc = dcnt; }
            }
            zprev = zhi;
            dcnt += 64 - __builtin_popcount(v.z) - __builtin_popcount(v.w);
        }
    }
    const int dbase = wave_incl_scan(dcnt) - dcnt;             // samples dropped before my piece
    int stop;                                                  // the next lane's anchor (none: the read's end)
    {
        int sfx = start;                                       // inclusive minimum over lanes >= mine
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_down(sfx, d);
            if (lane + d < 64) sfx = min(sfx, t);
        }
        const int nxt = __shfl_down(sfx, 1);
        stop = lane < 63 ? min(nxt, M) : M;
    }

    // ---- the walk of my piece (k_seg_walk4's step; no jumps) ----
    const unsigned thr0 = (unsigned)min(p.window, p.first_len);
    bool haveF = false, Fbig = false;
    int Fs = 0, Fe = 0, nB = 0;
    int in_run = 0, zl = 0, rstart = 0, last1 = 0;
    int pos = start, ce = -2, dcur = 0, rundrops = 0;
    const int jump_d = dbase + dloc;
    bool fin = !(start < M);
    unsigned long long Oc = 0ull, Zc = 0ull, On = 0ull, Zn = 0ull;
    while (true) {
        const bool act = !fin && pos < M;
        if (__builtin_amdgcn_ballot_w64(act) == 0ull) break;
        if (!act) continue;
        const int e = pos >> 6;
        if (e != ce) {                                         // the two entries my window lies in
            // (a position never advances by more than 64 samples: after the first visit the new entry is the next one)
            if (ce != -2) { dcur += 64 - __builtin_popcountll(Oc | Zc); Oc = On; Zc = Zn; }
            else {
                const uint4 v = entry(e);
                Oc = ((unsigned long long)(v.y & v.w) << 32) | (v.x & v.z);
                Zc = ((unsigned long long)(~v.y & v.w) << 32) | (~v.x & v.z);
                dcur = jump_d;                                 // samples dropped before the anchor's entry
            }
            const uint4 v = entry(e + 1);
            On = ((unsigned long long)(v.y & v.w) << 32) | (v.x & v.z);
            Zn = ((unsigned long long)(~v.y & v.w) << 32) | (~v.x & v.z);
            ce = e;
        }
        const int sh = pos & 63;
        unsigned long long Ow = (Oc >> sh) | ((On << 1) << (63 - sh));
        unsigned long long Zw = (Zc >> sh) | ((Zn << 1) << (63 - sh));
        int V = 64;
        if (!in_run) {
            if (Ow == 0ull) { pos += 64; if (pos >= stop) fin = true; continue; }
            const int t = __builtin_ctzll(Ow);
            pos += t; Ow >>= t; Zw >>= t; V -= t;
            if (pos >= stop) { fin = true; continue; }         // that run is the next lane's
            in_run = 1; rstart = pos; zl = E1; last1 = 0; rundrops = 0;
        }
        const unsigned long long Z = Zw & ((2ull << (V - 1)) - 1ull);
        const int nz = __builtin_popcountll(Z);
        const bool closes = nz >= zl;
        unsigned long long Zk = Z;
        const int kth = closes ? zl : 1;
        for (int i = 1; i < kth; i++) Zk &= Zk - 1ull;
        const int L = closes ? (int)__builtin_ctzll(Zk) : V;                  // samples of the piece (without the closing one)
        const int zeros = closes ? zl - 1 : nz;                              // out-of-band samples in it
        const unsigned long long below = (L == 64) ? ~0ull : ((1ull << L) - 1ull);
        const unsigned long long ones = Ow & below;
        rundrops += L - zeros - __builtin_popcountll(ones);                  // neither in band nor out of band: dropped
        if (ones) last1 = zeros - __builtin_popcountll(Z & below & ((2ull << (63 - __builtin_clzll(ones))) - 1ull));
        else last1 += zeros;                                                 // prev_err (:445, :449)
        zl -= zeros;
        if (closes) {
            const int z = pos + L;
            const int c = (z - rstart) - rundrops;
            if ((unsigned)c >= thr0) {                         // :448-454, in filtered coordinates
                const int off = z - 64 * ce;                   // z lies in one of my two entries
                const unsigned long long dc = ~(Oc | Zc), dn = ~(On | Zn);
                const int dz = dcur + (off >= 64 ? __builtin_popcountll(dc) + __builtin_popcountll(dn & ((1ull << (off - 64)) - 1ull))
                                                 : __builtin_popcountll(dc & ((1ull << off) - 1ull)));
                const int zf = z - dz;
                const bool big = c >= p.window;
                if (!haveF) { haveF = true; Fbig = big; Fs = zf - c; Fe = zf - last1; }
                if (big) { if (nB < WL_NB) cand[64 * nB] = make_int2(zf - c, zf - last1); nB++; }
            }
            in_run = 0;
            pos = z + 1;
        } else pos += V;
    }

    // ---- combine, in order (uniform across the wave; lane 0 writes) ----
    if (__builtin_amdgcn_ballot_w64(nB > WL_NB) != 0ull) {     // a piece with more long runs than the list holds
        if (lane == 0) nsegs[r] = walk_sync_read(mrow, nent, nent, p, my, max_segs);
        return;
    }
    const unsigned long long mF = __builtin_amdgcn_ballot_w64(haveF);
    int nseg = 0, last_end = 0;
    auto report = [&](int s0, int e0) __attribute__((always_inline)) {
        if (nseg > 0 && s0 - last_end < p.seg_dist) {          // :451 merge
            if (lane == 0 && nseg <= max_segs) my[2 * (nseg - 1) + 1] = e0;
        } else {
            if (lane == 0 && nseg < max_segs) { my[2 * nseg] = s0; my[2 * nseg + 1] = e0; }
            nseg++;
        }
        last_end = e0;
    };
    if (mF != 0ull) {
        const int j0 = (int)__builtin_ctzll(mF);
        report(__builtin_amdgcn_readlane(Fs, j0), __builtin_amdgcn_readlane(Fe, j0));
        const int skip0 = __builtin_amdgcn_readlane((int)Fbig, j0);          // its first long run IS that first run
        unsigned long long mB = __builtin_amdgcn_ballot_w64(nB > 0) & (~0ull << j0);
        while (mB != 0ull) {
            const int j = (int)__builtin_ctzll(mB);
            mB &= mB - 1ull;
            const int n = __builtin_amdgcn_readlane(nB, j);
            const int2 *cj = (const int2 *)(wl_lds + lds_entries) + j;
            for (int i = (j == j0 ? skip0 : 0); i < n; i++) { const int2 q = cj[64 * i]; report(q.x, q.y); }
        }
    }
    if (lane == 0) nsegs[r] = nseg;
}

// k_seg_walk4's preconditions (else: k_seg_walk3 / k_seg_walk2)
bool walk_jumps_apply(const WalkParams &wp, bool fast, bool by_runs, int row16)
{
    return fast && by_runs && wp.error < 32 && wp.window >= 127 && (int64_t)row16 * 64 <= 65536 &&
           sk_tune("SK_WALK_SYNC") == nullptr;
}

void launch_walk(hipStream_t ws, const uint4 *mask2, int row16, const int32_t *len, int64_t stride, int nr,
                 const WalkParams &wp, bool fast, bool by_runs, int32_t *d_segs, int32_t *d_nsegs, int max_segs,
                 const unsigned *d_hints = nullptr)
{
    const int wgrid = (nr + 63) / 64;
    // long rows, or too few reads to fill the chip with a lane each: a wavefront per read (k_seg_walkL)
    // (measured crossovers against the lane-per-read walk, same box: rows of 4 000 samples between 10 000 and 25 000
    // reads -- 0.039 / 0.055 ms at 10 000, 0.076 / 0.057 at 25 000 --, rows of 20 000 samples near 200 000 reads -- 0.35 / 0.76
    // at 50 000, 1.29 / 1.23 at 200 000, 2.54 / 1.96 at 400 000: the wave-per-read walk is bound by vector issue and grows
    // with the batch, the lane-per-read one by one lane's latency chain and barely does)
    int long_max = row16 > 64 ? 131072 : 16384;
    if (const char *e = sk_tune("SK_WALK_WAVE_MAXREADS")) long_max = atoi(e);
    if (fast && by_runs && wp.error < 32 && nr <= long_max && nr > 0 &&
        sk_tune("SK_WALK_NOWAVE") == nullptr && sk_tune("SK_WALK_SYNC") == nullptr) {
        // rows of up to 1 024 entries are staged in LDS; longer rows (only some of whose reads may fit) read global memory
        const bool staged = row16 <= 1024;
        const int lds_entries = staged ? row16 : 0;
        const size_t lds = (size_t)lds_entries * 16 + (size_t)WL_NB * 64 * sizeof(int2);
        auto fn = wp.error == 5 ? (staged ? k_seg_walkL<6, true> : k_seg_walkL<6, false>)
                                : (staged ? k_seg_walkL<0, true> : k_seg_walkL<0, false>);
        hipLaunchKernelGGL(fn, dim3(nr), dim3(64), lds, ws, mask2, row16, len, stride, nr, wp, d_segs, d_nsegs, max_segs,
                           lds_entries);
        return;
    }
    if (walk_jumps_apply(wp, fast, by_runs, row16)) {
        const int use_jumps = sk_tune("SK_WALK_NOJUMP") == nullptr;
        const unsigned *h = use_jumps ? d_hints : nullptr;
        auto fn = wp.error == 5 ? (h ? k_seg_walk4<6, true> : k_seg_walk4<6, false>)
                                : (h ? k_seg_walk4<0, true> : k_seg_walk4<0, false>);
        hipLaunchKernelGGL(fn, dim3(wgrid), dim3(64), 0, ws, mask2, row16, len, stride, nr, wp, d_segs, d_nsegs, max_segs,
                           use_jumps, h);
    } else if (fast && by_runs)
        hipLaunchKernelGGL(k_seg_walk3, dim3(wgrid), dim3(64), 0, ws, mask2, row16, len, stride, nr, wp, d_segs, d_nsegs,
                           max_segs);
    else if (fast)
        hipLaunchKernelGGL(k_seg_walk2<true>, dim3(wgrid), dim3(64), 0, ws, mask2, row16, len, stride, nr, wp, d_segs,
                           d_nsegs, max_segs);
    else
        hipLaunchKernelGGL(k_seg_walk2<false>, dim3(wgrid), dim3(64), 0, ws, mask2, row16, len, stride, nr, wp, d_segs,
                           d_nsegs, max_segs);
}

WalkParams walk_params(const sk_seg_params *p, bool *fast)
{
    WalkParams wp;
    wp.error = p->error; wp.corrector = p->corrector; wp.window = p->window; wp.seg_dist = p->seg_dist;
    const double fl = (double)p->window * p->stall_len;            // segmenter.py:448
    if (!(fl == fl))           wp.first_len = 0x7fffffff;          // NaN: never true
    else if (fl > 2147483000.) wp.first_len = 0x7fffffff;
    else if (fl < -2147483000.) wp.first_len = -0x7fffffff;
    else                       wp.first_len = (int)ceil(fl);
    *fast = wp.error < wp.corrector && wp.window >= 1 && wp.first_len >= 1 && sk_tune("SK_WALK_GENERAL") == nullptr;
    return wp;
}

typedef void (*segstat_fn)(const SegStatArgs);

// reads of 4 097 .. 65 536 samples: the workgroup-per-read kernel (one look); *wg = its workgroup size, else 0
segstat_fn pick_stats_wg(int64_t stride, int nbins, bool pa)
{
    if (stride <= 4096 || stride > 65536 || sk_tune("SK_SEG_NO_WG")) return nullptr;
    if (!pa && nbins > MAXBINS) return nullptr;
    // Where it pays (same box, statistics kernel alone, ms; wavefront per read / workgroup per read): int16 rows
    // 100 000 x 8 192: 0.67 / 0.89; 50 000 x 20 000: 0.69 / 0.78; 25 000 x 36 977: 0.78 / 0.64; 20 000 x 65 535: 1.17 / 0.94
    // (with the wavefront kernel keeping its last two windows in registers, later in round 6: 0.40, 0.54, 0.65, 0.99);
    // pA rows 0.70 / 0.95, 0.75 / 0.85, 1.17 / 0.94 at the last three -- one look halves the fetched bytes, but a read's
    // loads, its barrier and its statistics are a latency chain that three or four resident workgroups per CU hide
    // worse than sixteen independent wavefronts do.  SK_SEG_WG_ALL=1 takes it for every long row (tests).
    if (!sk_tune("SK_SEG_WG_ALL") && stride <= (pa ? 49152 : 32768)) return nullptr;
    const int tiles = (int)((stride + 511) / 512);         // 512-sample tiles, dealt to the four waves
    const int KT = tiles <= 32 ? 8 : tiles <= 48 ? 12 : tiles <= 64 ? 16 : tiles <= 80 ? 20 : tiles <= 96 ? 24 : 32;
    switch (KT) {
    case 8:  return pa ? k_seg_stats_wg<8, true> : k_seg_stats_wg<8, false>;
    case 12: return pa ? k_seg_stats_wg<12, true> : k_seg_stats_wg<12, false>;
    case 16: return pa ? k_seg_stats_wg<16, true> : k_seg_stats_wg<16, false>;
    case 20: return pa ? k_seg_stats_wg<20, true> : k_seg_stats_wg<20, false>;
    case 24: return pa ? k_seg_stats_wg<24, true> : k_seg_stats_wg<24, false>;
    default: return pa ? k_seg_stats_wg<32, true> : k_seg_stats_wg<32, false>;
    }
}

segstat_fn pick_stats(int64_t stride, int nbins, bool pa)
{
    const bool small = nbins <= 1023;          // 4 KB of histogram per wave
    const int NT = (int)((stride + 511) / 512);
    if (pa) {                                  // per-read windows of up to 2 047 raw units
        if (NT <= 2) return k_seg_stats<2, 8, 8, false, true>;
        if (NT <= 4) return k_seg_stats<4, 8, 8, false, true>;
        if (NT > 8) return k_seg_stats<8, 8, 8, true, true>;
        return k_seg_stats<8, 8, 8, false, true>;
    }
    if (NT <= 2) return small ? k_seg_stats<2, 4, 8, false> : k_seg_stats<2, 8, 8, false>;
    if (NT <= 4) return small ? k_seg_stats<4, 4, 8, false> : k_seg_stats<4, 8, 8, false>;
    if (NT > 8) return small ? k_seg_stats<8, 4, 4, true> : k_seg_stats<8, 8, 8, true>;   // (two kept windows: 126 VGPRs)
    if (small) {
        if (const char *e = sk_tune("SK_SEG_OCC")) {         // tuning: registers per lane vs reads in flight
            const int v = atoi(e);
            if (v == 8) return k_seg_stats<8, 4, 8, false>;
            if (v == 7) return k_seg_stats<8, 4, 7, false>;
        }
        return k_seg_stats<8, 4, 6, false>;
    }
    return k_seg_stats<8, 8, 8, false>;
}

} // namespace

// 16-byte entries per read in the {in band, kept} mask: 8 per 512-sample tile the statistics kernel holds
// (reads longer than 4 096 samples: whole 64-entry windows)
int sk_segment_fast_row16(int64_t stride)
{
    const int NT = (int)((stride + 511) / 512);
    if (NT > 8) return 64 * (int)((stride + 4095) / 4096);
    return 8 * (NT <= 2 ? 2 : NT <= 4 ? 4 : 8);
}

// The raw-domain pA route: any limits (they are per read, in the kernel); rows as the streaming path wants them.
bool sk_segment_pa_applies(const void *d_sig, int64_t stride, double std_scale)
{
    if (sk_tune("SK_SEG_PA_F64")) return false;              // A/B switch: expand to float64, the float64 kernels
    if (stride > MAXLONG || (stride % 8) != 0 || ((uintptr_t)d_sig & 15) != 0) return false;
    if (!(std_scale == std_scale) || fabs(std_scale) > 1e6) return false;
    return true;
}

// Is (stride, limits, std_scale) inside the streaming path's range?  (else: k_prep_i16 + k_segment_walk)
bool sk_segment_fast_applies(const void *d_sig, int64_t stride, int32_t lo, int32_t hi, double std_scale)
{
    if (sk_tune("SK_SEG_OLD")) return false;                 // A/B switch: the numpy-order kernels for everything
    const int64_t nbins = (int64_t)hi - lo - 1;
    if (nbins < 1 || nbins > MAXBINS) return false;
    if (stride > MAXLONG || (stride % 8) != 0 || ((uintptr_t)d_sig & 15) != 0) return false;
    if (!(std_scale == std_scale) || fabs(std_scale) > 1e6) return false;
    return true;
}

// Streaming statistics, the numpy-order redo of the (almost always empty) list of uncertified reads, then the walk.
// Large batches go in SK_SEG_CHUNKS chunks (4), the walk of chunk i on a second stream beside the statistics of
// chunk i + 1 (see below; folding the walk into the statistics kernel itself was built and measured too: DESIGN 4.0).
// d_retry: nreads + 16 ints (per chunk: [0] = count, [1 ..] = list; zeroed here).  Records ev[0..3] like the
// other segment paths: ev[0]..ev[1] statistics of all chunks, ev[2]..ev[3] what is left of the walks after that.
int sk_launch_segment_fast(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads,
                           const sk_seg_params *p, int32_t lo, int32_t hi, sk_prep *d_prep, void *d_mask2,
                           int32_t *d_retry, int32_t *d_segs, int32_t *d_nsegs, int32_t max_segs,
                           const double *d_cal, double *d_scratch)
{
    segstat_fn fn = pick_stats(stride, hi - lo - 1, d_cal != nullptr);
    const segstat_fn fn_wg = pick_stats_wg(stride, hi - lo - 1, d_cal != nullptr);
    if (fn_wg) fn = fn_wg;
    SegStatArgs a;
    a.stride = stride; a.lo = lo; a.hi = hi; a.cal = d_cal;
    a.std_scale = p->std_scale; a.delta_scale = 1.0;
    if (const char *e = sk_tune("SK_SEG_DELTA_SCALE")) { const double v = atof(e); if (v > 0) a.delta_scale = v; }
    a.row16 = sk_segment_fast_row16(stride);

    bool fast;
    const WalkParams wp = walk_params(p, &fast);

    const bool by_runs = sk_tune("SK_WALK_STEP") == nullptr;       // A/B switch: the per-sample straight-line walk
    // Large batches go in four chunks, the walk of one on a second stream beside the statistics of the next -- which
    // only pays when the persistent statistics grid leaves the walk's waves room on the SIMDs: at its full six
    // workgroups per CU (78 VGPRs x 24 waves) nothing else fits and the overlap gained nothing (round 2: 3.35 / 3.33 /
    // 3.38 / 3.78 ms for 1 / 2 / 4 / 8 chunks); with four workgroups per CU the statistics kernel is as fast (it is
    // co-limited by HBM) and the step drops from 2.89 to 2.70 ms per 1 M reads (round 3, same box).
    // (round 4) k_seg_walk4 takes a fifth of the statistics kernel's time and gains nothing beside it (1 / 2 / 3 / 4 / 6
    // chunks: 2.53 / 2.56 / 2.57 / 2.58 / 2.60 ms per 1 M reads): one chunk; the older walks keep the four.
    int nchunks = nreads >= 262144 && !walk_jumps_apply(wp, fast, by_runs, a.row16) ? 4 : 1;
    if (const char *e = sk_tune("SK_SEG_CHUNKS")) { int v = atoi(e); if (v >= 1 && v <= 8) nchunks = v; }
    if (nreads < 65536) nchunks = 1;
    if (nchunks > 1 && !c->stream2) {
        SK_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        for (int i = 0; i < 9; i++) SK_HIP(hipEventCreateWithFlags(&c->ev_chunk[i], hipEventDisableTiming));
    }
    int per_cu = nchunks > 1 ? 4 : 8, rounds = 8;
    if (const char *e = sk_tune("SK_PREP_ROUNDS")) { int v = atoi(e); if (v > 0) rounds = v; }
    if (const char *e = sk_tune("SK_PREP_PERCU")) { int v = atoi(e); if (v > 0 && v < per_cu) per_cu = v; }

    // the walk's hints come out of the statistics kernel (reads of up to 4 096 samples, the jumping walk)
    a.hints = nullptr; a.e1 = (p->error > 0 ? p->error : 0) + 1;
    if (walk_jumps_apply(wp, fast, by_runs, a.row16) && stride <= 4096 && sk_tune("SK_WALK_OWNPASS") == nullptr) {
        if (int rch = sk_reserve(c, &c->seghints, (size_t)nreads * SEG_HINTS * sizeof(unsigned))) return rch;
        a.hints = (unsigned *)c->seghints.p;
    }
    unsigned *const hints0 = a.hints;
    if (d_cal) c->pa_retry_ptrs.clear();
    SK_HIP(hipMemsetAsync(d_retry, 0, ((size_t)nreads + 16) * sizeof(int32_t), c->stream));
    SK_HIP(hipEventRecord(c->ev[0], c->stream));
    if (nchunks > 1) {                                             // the second stream starts behind the memsets
        SK_HIP(hipEventRecord(c->ev_chunk[8], c->stream));
        SK_HIP(hipStreamWaitEvent(c->stream2, c->ev_chunk[8], 0));
    }
    const int32_t per = (int32_t)(((int64_t)nreads + nchunks - 1) / nchunks);
    for (int ci = 0; ci < nchunks; ci++) {
        const int32_t r0 = ci * per;
        const int32_t nr = (nreads - r0 < per) ? nreads - r0 : per;
        if (nr <= 0) break;
        int32_t *retry = d_retry + r0 + ci;
        a.sig = d_sig + (int64_t)r0 * stride; a.len = d_len + r0; a.nreads = nr;
        a.prep = d_prep + r0; a.mask2 = (uint4 *)d_mask2 + (int64_t)r0 * a.row16; a.retry = retry;
        a.hints = hints0 ? hints0 + (int64_t)r0 * SEG_HINTS : nullptr;
        a.cal = d_cal ? d_cal + 2 * (int64_t)r0 : nullptr;
        // persistent grid: a whole number of "rounds" of what the chip actually holds (6 workgroups per CU at 78
        // VGPRs, not the 8 the thread limit allows) -- with 8 assumed the last round ran a third full
        int resident = per_cu;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, (const void *)fn, 64 * WPB, 0) != hipSuccess || resident < 1)
            resident = per_cu;
        if (resident > per_cu) resident = per_cu;
        const long long g = (long long)c->num_cu * resident * rounds;
        const long long need = fn_wg ? (long long)nr : ((long long)nr + WPB - 1) / WPB;   // (a workgroup per read / per WPB reads)
        const int grid = (int)(g > need ? need : g);
        hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * WPB), 0, c->stream, a);
        SK_HIP(hipGetLastError());
        // reads whose ceil(top) / floor(bot) could not be certified: numpy-order statistics, masks rewritten in place
        // (reads too long for the redo's LDS copy: one scratch row per workgroup of its persistent grid, <= one per CU)
        if (d_cal) {
            c->pa_retry_ptrs.push_back(retry);
            // (pA: the listed reads from their float64 values, one scratch row of doubles per workgroup)
            const int g2 = nr < c->num_cu ? nr : c->num_cu;
            int rc = sk_launch_prep_pa_listed(c, a.sig, stride, a.len, a.cal, retry + 1, retry, g2, (double)lo, (double)hi,
                                              p->std_scale, d_scratch, stride, a.prep, a.mask2, a.row16);
            if (rc) return rc;
        }
        int16_t *scratch_rows = nullptr;
        if (!d_cal && stride * (int64_t)sizeof(int16_t) > 24 * 1024) {
            int rc0 = sk_reserve(c, &c->comp, (size_t)c->num_cu * (size_t)stride * sizeof(int16_t));
            if (rc0) return rc0;
            scratch_rows = (int16_t *)c->comp.p;
        }
        if (!d_cal) {
            int rc = sk_launch_prep_i16(c, a.sig, stride, a.len, nr, lo, hi, SK_PREP_SEGMENT, p->std_scale, scratch_rows, a.prep,
                                        nullptr, 0, 0, 0x7fffffff, retry + 1, retry, a.mask2, a.row16);
            if (rc) return rc;
        }
        hipStream_t ws = c->stream;
        if (nchunks > 1) {
            SK_HIP(hipEventRecord(c->ev_chunk[ci], c->stream));
            SK_HIP(hipStreamWaitEvent(c->stream2, c->ev_chunk[ci], 0));
            ws = c->stream2;
        } else {
            SK_HIP(hipEventRecord(c->ev[1], c->stream));
            SK_HIP(hipEventRecord(c->ev[2], c->stream));
        }
        launch_walk(ws, (const uint4 *)a.mask2, a.row16, a.len, stride, nr, wp, fast, by_runs,
                    d_segs + (int64_t)r0 * 2 * max_segs, d_nsegs + r0, max_segs, a.hints);
        SK_HIP(hipGetLastError());
    }
    if (nchunks > 1) {
        SK_HIP(hipEventRecord(c->ev[1], c->stream));
        SK_HIP(hipEventRecord(c->ev[2], c->stream));
        SK_HIP(hipEventRecord(c->ev_chunk[8], c->stream2));        // the caller's stream continues behind the walks
        SK_HIP(hipStreamWaitEvent(c->stream, c->ev_chunk[8], 0));
    }
    SK_HIP(hipEventRecord(c->ev[3], c->stream));
    return SK_OK;
}

// The walk alone, over {in band, kept} entries somebody else wrote (the float64 statistics kernel, sk_f64stat.hip):
// read r has d_len[r] samples (clamped to 64 row16), its entries at d_mask2 + r * row16.  Records ev[2] .. ev[3].
int sk_launch_seg_walk_masks(sk_ctx *c, const void *d_mask2, int row16, const int32_t *d_len, int32_t nreads,
                             const sk_seg_params *p, int32_t *d_segs, int32_t *d_nsegs, int32_t max_segs)
{
    if (nreads <= 0) return SK_OK;
    bool fast;
    const WalkParams wp = walk_params(p, &fast);
    SK_HIP(hipEventRecord(c->ev[2], c->stream));
    launch_walk(c->stream, (const uint4 *)d_mask2, row16, d_len, (int64_t)row16 * 64, nreads, wp, fast,
                sk_tune("SK_WALK_STEP") == nullptr, d_segs, d_nsegs, max_segs);
    SK_HIP(hipGetLastError());
    SK_HIP(hipEventRecord(c->ev[3], c->stream));
    return SK_OK;
}
