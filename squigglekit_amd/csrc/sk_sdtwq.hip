// sk_sdtwq.hip -- subsequence DTW by fixed-point screening + a certified exact window.
//
// The exact FP64 recurrence costs 16 VALU cycles per cell (32 with start tracking), and it has to
// be exact: MotifSeq prints the distance with 17 digits and the path's start/end columns
// (/root/reference/MotifSeq.py:437-449).  But almost all of those cells only serve to show that
// they do NOT hold the minimum.  So:
//
//  pass Q  (k_sdtw_q)   the same wave-systolic sweep in 32-bit fixed point (1 unit = 2^-22):
//          v_min3_u32 + v_sad_u32(clamp) = 2 instructions, 8 cycles per cell.  Its cost matrix Dq
//          differs from the exact one by at most E = N + n + 2 units in any cell (each local cost
//          |q(x)-q(y)| is within one unit of |x-y|, a path has at most N + n cells, FP64 rounding
//          is 10^-10 of a unit).  It stores the last row and, every CK steps, its systolic state.
//  pass P  (k_sdtw_p)   per read: the columns whose screening cost is within 2E of the screening
//          minimum are the only ones that can hold the exact minimum [jlo..jhi]; pick the
//          checkpoint `span` columns before jlo and advance it IN FIXED POINT over the whole blocks
//          up to the restart column (the checkpoints are CK steps apart: those on average CK/2
//          columns would cost four times as much in the exact recurrence).  Leaves the restart
//          state and {tbase, jlo, jhi} of every read for pass W.
//  pass W  (k_sdtw_w)   restart with every restored cell set to a LOWER BOUND of its exact value
//          ((Dq - E) units) and S = -1, run the exact FP64 recurrence with start tracking up to
//          jhi, take the first exact argmin inside [jlo, jhi].
//          Certificate: every cell computed from lower bounds is itself a lower bound (the
//          recurrence is monotone, also after rounding).  If the tie-ordered back-trace of the
//          winning cell never touches a restored cell (S >= 0), then along it lower bound == exact
//          value, cell by cell from row 0 up, and every rejected predecessor is rejected in the
//          exact matrix too -- so distance, end and start are the reference's, bit for bit.
//          Otherwise (S = -1), or if the range is too wide, a sample left the fixed-point range,
//          or the minimum may have saturated: the read goes to the next tier / the retry list.
//  retry   the exact single pass (k_sdtw FULL) on the listed reads only.
//
// Lane layout (round 3): a read is spread over L = 8 lanes for motifs of up to 256 points (8 reads per
// wavefront, R = ceil(N / 8) <= 32 rows per lane), L = 16 up to 512 points, L = 64 beyond.  Per step a lane runs
// 2R cell instructions plus a fixed handful (the neighbour's row by DPP, the short-lane select, LDS traffic), so
// fewer, longer lanes spend a larger share of the issue slots on cells -- and N = 200 fills 8 x 25 slots exactly
// where 16 x 13 leaves 8 of 208 empty.  Groups of 8 lanes share a DPP row of 16: the shift that hands lane l-1's
// bottom row to lane l is a v_and_b32_dpp with a per-lane mask that zeroes what lane 8 would pick up from lane 7.
// Samples reach the lanes through a small LDS ring per read group in all three kernels (one ds_read with an
// immediate offset per step): the vector ALU is the bottleneck, the LDS pipe is idle.
//
// This file is compiled once per sample feed (sk_sdtwq_f64.hip / sk_sdtwq_raw.hip include it with
// SK_SDTWQ_FEED set), so that the three sets of template instantiations build in parallel.
#include "sk_sdtw_dev.h"
#include "sk_prepw_dev.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#ifndef SK_SDTWQ_FEED
#define SK_SDTWQ_FEED 0          /* SK_FEED_I16 */
#define SK_SDTWQ_MAIN 1
#endif

namespace {

// One column of R cells: nw[k] = min(|xq[k] - yq| + min3(old[k-1], old[k], nw[k-1]), 2^32-1), with
// dg standing in for old[-1] and up for nw[-1].  Written as asm blocks of four cells because the
// compiler has no v_sad_u32 pattern, and because it pads every asm statement boundary with a
// wait state (it cannot see inside): 4 statements per column instead of 26.
template <int R>
__device__ __forceinline__ void qcolumn(const unsigned (&old)[R], unsigned (&nw)[R], const unsigned (&xq)[R],
                                        unsigned yq, unsigned dg, unsigned up)
{
    constexpr int R4 = R & ~3;
#pragma unroll
    for (int k = 0; k < R4; k += 4) {
        asm("v_min3_u32 %0, %12, %4, %13\n\t"
            "v_sad_u32 %0, %8, %14, %0 clamp\n\t"
            "v_min3_u32 %1, %4, %5, %0\n\t"
            "v_sad_u32 %1, %9, %14, %1 clamp\n\t"
            "v_min3_u32 %2, %5, %6, %1\n\t"
            "v_sad_u32 %2, %10, %14, %2 clamp\n\t"
            "v_min3_u32 %3, %6, %7, %2\n\t"
            "v_sad_u32 %3, %11, %14, %3 clamp"
            : "=&v"(nw[k]), "=&v"(nw[k + 1]), "=&v"(nw[k + 2]), "=&v"(nw[k + 3])
            : "v"(old[k]), "v"(old[k + 1]), "v"(old[k + 2]), "v"(old[k + 3]),
              "v"(xq[k]), "v"(xq[k + 1]), "v"(xq[k + 2]), "v"(xq[k + 3]),
              "v"(dg), "v"(up), "v"(yq));
        dg = old[k + 3];
        up = nw[k + 3];
    }
#pragma unroll
    for (int k = R4; k < R; k++) {
        asm("v_min3_u32 %0, %3, %1, %4\n\t"
            "v_sad_u32 %0, %2, %5, %0 clamp"
            : "=&v"(nw[k])
            : "v"(old[k]), "v"(xq[k]), "v"(dg), "v"(up), "v"(yq));
        dg = old[k];
        up = nw[k];
    }
}

// biased fixed-point image of a normalised value, |v| < QLIM
__device__ __forceinline__ unsigned qimg(double v)
{
    return (unsigned)((int)rint(v * QSCALE)) + 0x80000000u;
}

// lane l <- lane l-1 of the same read group, 0 into the group's lane 0.  notfirst = 0 in a group's lane 0, ~0
// elsewhere (only looked at for L = 8, where two groups share a DPP row; the s_nop covers the
// VALU-write -> DPP-read hazard the compiler cannot see inside the statement).
template <int L>
__device__ __forceinline__ unsigned shr0(unsigned v, unsigned notfirst)
{
    if constexpr (L >= 16) {
        (void)notfirst;
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, (L == 16) ? DPP_ROW_SHR1 : DPP_WAVE_SHR1, 0xF, 0xF, true);
    } else {
        unsigned r;
        asm("s_nop 1\n\t"
            "v_and_b32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
            : "=v"(r) : "v"(v), "v"(notfirst));
        return r;
    }
}

// the restart point of one read, handed from pass P to pass W
struct wrec {
    int32_t tbase;      // first step of the exact recurrence
    int32_t jlo, jhi;   // candidate columns
    int32_t flags;      // 1: screened (a window exists)  2: the state was restored from a checkpoint
};
// A read whose candidate columns fall into TWO clusters more than wmax apart (two near-minima far from each other: 0.02 % of
// 4 000-sample reads, 0.5 % at 37 000 samples) used to take the exact single pass -- a full sweep, whose latency shows when
// the batch is small and the reads are long (3.4 ms beside 1 ms of window passes at 25 000 x 36 977).  Since round 5 the
// first cluster goes through the window passes as the read's own [jlo, jhi] and the second one as a SIBLING: a second,
// short launch of pass P / pass W over the sibling list, whose results a combine kernel merges (the smaller exact distance
// wins, the lower column on a tie -- np.argmin's first minimum; a sibling that cannot be certified sends the read to the
// exact pass after all).  Three clusters, or a cluster wider than wmax: exact pass, as before.
struct sibrec {
    int32_t r;          // read
    int32_t jlo, jhi;   // the second cluster's candidate columns
    int32_t pad;
};

// Candidate columns of read r: [jlo, jhi] = the first and the last column whose screening cost is within 2E of the
// screening minimum b (jhi < jlo: none).  Pass Q left, per checkpoint interval c and lane l', the minimum over the
// last-row columns c * ck + q * L + l' - (L - 1), q = 0 .. ck / L - 1: only intervals whose summary is within the
// threshold can hold a candidate, so a read costs (nck + 1) * L summary words here instead of its whole last row.
// Every lane of the read's group calls this with its own l and gets the group's result.
// (clo, chi: only columns inside [clo, chi] count -- the two clusters of a read whose candidates lie too far apart for one
// window, round 5)
template <int L>
__device__ __forceinline__ void candidate_columns(const sdtw_kargs &a, int r, int n, unsigned b, int l, int &jlo_out, int &jhi_out,
                                                  int clo = 0, int chi = 0x7fffffff)
{
    const unsigned *lastq = a.lastq + (int64_t)(r - a.read0) * a.lq_stride;
    const unsigned *ls = a.lsum + (int64_t)(r - a.read0) * (a.nck + 1) * L + l;
    const unsigned thr = (b > QINF - 2u * a.qerr) ? QINF : b + 2u * a.qerr;
    int jlo = 0x7fffffff, jhi = -1;
    if (n > 0 && b < QSAFE) {
        for (int cc = 0; cc <= a.nck; cc++) {
            if (ls[(int64_t)cc * L] > thr) continue;
            const int j0 = cc * a.ck + l - (L - 1);
            for (int q = 0; q < a.ck / L; q++) {
                const int j = j0 + q * L;
                if (j >= clo && j <= chi && j >= 0 && j < n && lastq[j + L] <= thr) { jlo = min(jlo, j); jhi = max(jhi, j); }   // (rows start L early)
            }
        }
    }
#pragma unroll
    for (int d = 1; d < L; d <<= 1) { jlo = min(jlo, __shfl_xor(jlo, d)); jhi = max(jhi, __shfl_xor(jhi, d)); }
    jlo_out = jlo; jhi_out = jhi;
}

// ---------------------------------------------------------------------------------------------
// pass Q
// ---------------------------------------------------------------------------------------------
#ifndef SK_Q_WAVES
#define SK_Q_WAVES(R) 4    /* four waves per SIMD (128 VGPRs): R <= 32 rows of 3 words + the feed; the little that does not fit is spilled from the cold ends of the kernel */
#endif
// P0: every lane owns R rows (the motif fills L x R slots exactly): no short-lane select after the column
template <int L, int R, int FEED, bool P0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SK_Q_WAVES(R), 8)))
void k_sdtw_q(const sdtw_kargs a)
{
    static_assert(L == 8 || L == 16 || L == 64, "lanes per read");
    constexpr int G = 64 / L;
    constexpr int CKH = (R + 3) / 2;                // dwords per lane and checkpoint: R + 2 state words, their high halves
    constexpr int U = (L < 16) ? L : 16;            // steps per unrolled run (y values prefetched from LDS)

    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int g = lane / L, l = lane % L;
    int slot = wave * G + g;
    const bool live = slot < a.nreads;
    if (!live) slot = a.nreads - 1;
    const int r = a.read0 + slot;
    // the clock this launch runs at (DVFS: it depends on what the chip is doing): the first wave times its own sweep
    // with the shader-cycle counter and the 100 MHz reference counter (sk_last_dtw_clock)
    const bool timer = a.clk && blockIdx.x == 0 && threadIdx.x < 64;
    unsigned long long tc0 = 0, tr0 = 0;
    if (timer) { tc0 = __builtin_amdgcn_s_memtime(); tr0 = __builtin_amdgcn_s_memrealtime(); }

    // per wave: lds_wave_words words of dynamic LDS -- the prologue's value histogram, then (the prologue is over by then)
    // the last-row values of the checkpoint interval in progress, ck words per read group
    extern __shared__ __align__(16) unsigned char lds_dyn[];
    __shared__ __align__(16) unsigned lds_all[4][5 * 64];   // per wave: the sample ring + dump rows of its read groups (the sweep);
                                                            // before that, 128 doubles of the zscale prologue's pairwise tree
    int n;
    double center = 0.0, scale = 1.0;
    const int16_t *s16 = nullptr;
    const double  *s64 = nullptr;
    if constexpr (FEED == SK_FEED_I16) {
        if (a.fz_raw) {
            // Fused prologue: scale_outliers + medmad of this wave's G reads, one after the other on the whole wave
            // (sk_prepw_dev.h: compacted samples -> a.samples, statistics -> a.prep, both also read by the later
            // passes).  Latency-bound work that the sweeps of the SIMD's other waves hide; as a kernel of its own it
            // cost 4 ms per 1 M reads.
            unsigned *wlds = (unsigned *)lds_dyn + (size_t)(threadIdx.x >> 6) * a.lds_wave_words;
            n = 0;
            if (a.fz_mode == 0) {
                const prepw_env E = prepw_setup<5>(wlds, lane, a.fz_lo, a.fz_hi, a.fz_vec);
                for (int gg = 0; gg < G; gg++) {
                    if (wave * G + gg >= a.nreads) break;       // (wave-uniform)
                    const sk_prep pr = prepw_read<5>(E, a.fz_raw, a.stride, a.fz_len, a.read0 + wave * G + gg, lane,
                                                     (int16_t *)a.samples, (sk_prep *)a.prep);
                    if (g == gg) { n = pr.n; center = pr.center; scale = pr.scale; }
                }
            } else {
                // zscale (round 5): numpy-order mean / std by this wave, the compacted read in its share of the dynamic
                // LDS, the pairwise tree's partial sums in the (not yet used) static sample ring
                const zs_env E = zs_setup((int16_t *)wlds, (double *)lds_all[threadIdx.x >> 6], a.fz_lo, a.fz_hi, a.fz_vec);
#pragma unroll 1
                for (int gg = 0; gg < G; gg++) {
                    if (wave * G + gg >= a.nreads) break;
                    const sk_prep pr = zs_read(E, a.fz_raw, a.stride, a.fz_len, a.read0 + wave * G + gg, lane,
                                               (int16_t *)a.samples, (sk_prep *)a.prep);
                    if (g == gg) { n = pr.n; center = pr.center; scale = pr.scale; }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            // the sweep below reads what other lanes of this wave just stored
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            const sk_prep pr = a.prep[r];
            n = pr.n; center = pr.center; scale = pr.scale;
        }
        s16 = (const int16_t *)a.samples + (int64_t)r * a.stride;
    } else if constexpr (FEED == SK_FEED_F64_NORM) {
        const sk_prep pr = a.prep[r];
        n = pr.n; center = pr.center; scale = pr.scale;
        s64 = (const double *)(((pr.flags & SK_IFLAG_INPLACE) && a.samples_raw) ? a.samples_raw : a.samples) + a.off[r];
    } else {
        n = (int)(a.off[r + 1] - a.off[r]);
        s64 = (const double *)a.samples + a.off[r];
    }
    if (!live) n = 0;
    const double inv_scale = 1.0 / scale;

    int nsteps = (n > 0) ? n - 1 + L : 0;
#pragma unroll
    for (int d = L; d < 64; d <<= 1) nsteps = max(nsteps, __shfl_xor(nsteps, d));
    nsteps = __builtin_amdgcn_readfirstlane(nsteps);
    const int nblk = (nsteps + L - 1) / L;

    unsigned xq[R];
#pragma unroll
    for (int k = 0; k < R; k++) xq[k] = a.xlayq[l * R + k];
    const bool shortlane = l < a.P;
    const unsigned notfirst = (l == 0) ? 0u : 0xffffffffu;

    unsigned Da[R], Db[R];                          // ping-pong: a column reads one, writes the other
#pragma unroll
    for (int k = 0; k < R; k++) { Da[k] = QINF; Db[k] = QINF; }
    unsigned botq = (R == 1 && l == 0 && shortlane) ? 0u : QINF;
    unsigned diagq = (l == 0) ? 0u : QINF;
    int bad = 0;
    if constexpr (FEED == SK_FEED_F64_NORM) {       // re-centred zscale reads: exact feed only (retry pass)
        if (live && (a.prep[r].flags & SK_FLAG_RECENTRE)) bad = 1;
    }
    unsigned qmin = QINF;                           // minimum of the last row so far
    unsigned smin = QINF;                           // ... of this lane's last-row values since the last summary

    // the sample feed in two halves, so that the load for the next block is in flight during this
    // block's L steps and only converted afterwards
    // (unconditional load from a clamped index: a load under a branch is waited for at the join)
    const int nlast = max(n - 1, 0);
    if (n == 0) { s16 = (const int16_t *)a.xlayq; s64 = (const double *)a.xlayq; }   // any valid address
    auto loadraw = [&](int idx) {
        if constexpr (FEED == SK_FEED_I16) return (int)s16[min(idx, nlast)];
        else                               return s64[min(idx, nlast)];
    };
    // screening only: fma(x, 2^22 / s, -c 2^22 / s) instead of the reference's (x - c) / s -- the two differ by
    // < 1e-4 of a fixed-point unit for int16 samples (|c / s| < 32 768 / 0.74: the rounding of the second constant is
    // 2e-5 units), covered by the slack in E (pass W divides, exactly).  float64 samples get (x - c) * (2^22 / s): the
    // constant -c 2^22 / s is only good to 1e-16 of ITS size, which for a near-constant read (c / s ~ 1e14; found by the
    // round-4 fuzz once its float64 batches were large enough to be screened) is 1e5 units -- the difference is taken
    // first, exactly for samples near c.  Rounding to the nearest integer by adding 1.5 * 2^52: the sum's low word is
    // the two's complement integer.
    const double qa = inv_scale * QSCALE, qb = -center * qa;
    // Guard, per read (round 5): the certificate's E = N + n + 2 leaves every sample image 1/2 unit of rounding plus
    // an evaluation error of at most 1 / (N + n) units.  What the evaluation can be off by is known here: the fma form
    // is t (e1 + e3) - c qa e2 with |e| <= 2^-53 (the reciprocal, the fma's rounding, the rounding of the constant
    // -c qa), i.e. <= 3.8e-7 + |c| qa 1.2e-16 units for |t| < 400 * 2^22; the difference form (float64 reads) is three
    // roundings of t, <= 5.7e-7.  A read whose bound times (N + n) exceeds 1 -- a level of 30 000 over a MAD of 1 in a
    // 50 000-sample read, or any near-constant float64 read if the fma form were used for it (the hole the round-4
    // fuzz found; SK_DTW_HOLE=fma64 puts it back for the tests) -- is not screened: exact pass, counted (IMGREJ).
    const bool fma64 = FEED == SK_FEED_F64_NORM && a.hole >= SK_HOLE_FMA64;
    if (live && n > 0) {
        double imgerr = 0.0;
        if (FEED == SK_FEED_I16 || fma64)            imgerr = 3.8e-7 + fabs(center) * qa * 1.2e-16;
        else if constexpr (FEED == SK_FEED_F64_NORM) imgerr = 5.7e-7;
        const bool rej = !(imgerr * (double)a.qerr <= 1.0) && a.hole != SK_HOLE_FMA64_UNGUARDED;   // (NaN: rejected)
        if (rej) {
            bad = 1;
            if (l == 0 && a.guard) atomicAdd(&a.guard[SK_GUARD_IMGREJ], 1);
        }
    }
    auto toq = [&](auto raw_, int idx) -> unsigned {
        const double raw = (double)raw_;
        double t;
        if constexpr (FEED == SK_FEED_I16)           t = __builtin_fma(raw, qa, qb);
        else if constexpr (FEED == SK_FEED_F64_NORM) t = fma64 ? __builtin_fma(raw, qa, qb) : (raw - center) * qa;
        else                                         t = raw * QSCALE;
        const bool ok = fabs(t) < QLIM * QSCALE;     // false for NaN / inf too
        const unsigned q = (unsigned)__double2loint(t + 6755399441055744.0) ^ 0x80000000u;
        const bool in = idx < n;
        bad |= (in && !ok) ? 1 : 0;
        return (in && ok) ? q : QINF;
    };

    const unsigned smask = shortlane ? 0xffffffffu : 0u;
    unsigned *lastq = a.lastq + (int64_t)(r - a.read0) * a.lq_stride;
    // Per read group, in LDS: the sample images of the previous and the current block (lane l
    // needs sample t - l at step t: one ds_read with an immediate offset instead of a DPP shift
    // chain on the vector ALU, which is the bottleneck), and the last-row values lane L-1 produces.
    //   ybuf[0,L) even blocks | ybuf[L,2L) odd blocks | ybuf[2L,3L) copy of [0,L)
    // so that sample t0 + q - l of an even block is ybuf[2L + q - l] and of an odd one ybuf[L + q - l].
    unsigned *ybuf = lds_all[threadIdx.x >> 6] + g * 5 * L;
    unsigned *dump = ybuf + 3 * L;                  // 2 L words: where the lanes that do not hold the last row write
    // The last row is kept per CHECKPOINT INTERVAL in LDS and goes to memory only when the interval can hold a candidate
    // column (round 5).  A column is a candidate when its cost is within 2 E of the read's FINAL minimum b; the minimum so
    // far, m, is >= b, so an interval whose smallest value exceeds m + 2 E holds none -- and every later look at the row
    // (this kernel's epilogue, pass P's second tier, pass W's premise test) only visits intervals whose per-lane
    // summaries are <= b + 2 E, i.e. intervals that WERE stored.  On the C4 batch about one interval in eight goes out:
    // 16 KB of last row per read became 2-3.
    unsigned *ibuf = (unsigned *)lds_dyn + (size_t)(threadIdx.x >> 6) * a.lds_wave_words + g * a.ck;
    int ioff = 0;                                   // steps of the current interval done
    unsigned gmin = QINF;                           // minimum of the read's last row over the intervals already closed
    ybuf[L + l] = QINF;                             // columns before the read
    unsigned F = toq(loadraw(l), l);
    // one step: lane l-1's bottom row comes in by DPP, run the column old -> nw
    auto step = [&](const unsigned (&old)[R], unsigned (&nw)[R], unsigned yq, unsigned *hslot) {
        const unsigned upq = shr0<L>(botq, notfirst);
        qcolumn<R>(old, nw, xq, yq, diagq, upq);
        diagq = upq;
        // (a bit select, v_bitop3_b32: 2 cycles of issue against v_cndmask's 4)
        if constexpr (P0)          botq = nw[R - 1];
        else if constexpr (R >= 2) asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe4" : "=v"(botq) : "v"(nw[R - 2]), "v"(nw[R - 1]), "v"(smask));
        else                       asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe4" : "=v"(botq) : "v"(upq), "v"(nw[0]), "v"(smask));
        *hslot = nw[R - 1];                         // (only lane L-1's lands in the interval buffer)
    };
    // close interval c (nb blocks of it are in ibuf): store it if it can hold a candidate
    auto flush = [&](int c, int nb) {
        unsigned im = smin;
#pragma unroll
        for (int d = 1; d < L; d <<= 1) im = min(im, (unsigned)__shfl_xor((int)im, d));
        const unsigned lim = (gmin > QINF - 2u * a.qerr) ? QINF : gmin + 2u * a.qerr;
        const bool keep = live && im <= lim;
        gmin = min(gmin, im);
        if (keep) {
            unsigned *dst = lastq + (int64_t)c * a.ck + l + 1;        // (rows start L early: see below)
            for (int q = 0; q < nb; q++) dst[q * L] = ibuf[q * L + l];
        }
    };
    for (int blk = 0; blk < nblk; blk++) {
        const auto rawnext = loadraw((blk + 1) * L + l);
        const int t0 = blk * L;
        if (t0 > 0 && (t0 % a.ck) == 0) {                    // (wave-uniform) a checkpoint interval ends here
          if (t0 / a.ck <= a.nck && live) {
            // Checkpoints go out as the HIGH HALVES of the state words, two to a dword (round 5: 27 -> 14 dwords per lane
            // and checkpoint, 27 KB -> 14 KB per read; pass Q 48.5 -> 47.4 ms).  What is dropped is less than 2^16 units =
            // 0.016 signal units, and it is dropped DOWNWARDS: a restored state is cell by cell <= the screening state and
            // >= it minus 2^16, the recurrence is monotone and non-expanding (min and saturating add), so everything the
            // pre-roll computes from it keeps that property -- and the window pass only ever uses restored cells as lower
            // bounds (Dq - E) of exact values (file header), which they still are.
            unsigned *cp = a.ckq + (((int64_t)(r - a.read0) * a.nck + (t0 / a.ck - 1)) * L + l) * CKH;
            auto hi2 = [](unsigned lo_word, unsigned hi_word) -> unsigned {   // {lo_word >> 16, hi_word >> 16} in one v_perm_b32
                return __builtin_amdgcn_perm(hi_word, lo_word, 0x07060302u);
            };
#pragma unroll
            for (int k = 0; k + 1 < R; k += 2) cp[k / 2] = hi2(Da[k], Da[k + 1]);
            if constexpr (R % 2) { cp[R / 2] = hi2(Da[R - 1], botq); cp[R / 2 + 1] = hi2(diagq, 0u); }
            else                 { cp[R / 2] = hi2(botq, diagq); }
            // summary of the last row: the minimum over the columns this lane handed to lastq during the ck steps
            // before this one (columns == l + 1 mod L of one window of ck columns) -- pass P looks at the
            // (nck + 1) * L summaries of a read instead of all its columns
            a.lsum[((int64_t)(r - a.read0) * (a.nck + 1) + (t0 / a.ck - 1)) * L + l] = smin;
          }
            flush(t0 / a.ck - 1, a.ck / L);
            smin = QINF;
            ioff = 0;
        }
        unsigned *hw = (l == L - 1) ? ibuf + ioff : dump;             // every other lane writes to a dump row
        const unsigned *yr;
        if (blk & 1) { ybuf[L + l] = F; yr = ybuf + L - l; }
        else         { ybuf[l] = F; ybuf[2 * L + l] = F; yr = ybuf + 2 * L - l; }
#pragma unroll 1
        for (int qq = 0; qq < L; qq += U) {
            unsigned yv[U];
#pragma unroll
            for (int q = 0; q < U; q++) yv[q] = yr[qq + q];
#pragma unroll
            for (int q = 0; q < U; q += 2) {        // after two steps the ping-pong roles are back
                step(Da, Db, yv[q], hw + qq + q);
                step(Db, Da, yv[q + 1], hw + qq + q + 1);
            }
        }
        F = toq(rawnext, (blk + 1) * L + l);
        // lane L-1's column at step t0 + l is t0 + l - (L - 1).  Rows of lastq carry L columns of padding in front and
        // 2 L behind, so every lane stores without a range test; columns outside the read hold costs of 2^30 units
        // and more (their sample is "infinite"), which neither the summaries nor pass P's column test (j < n) mind.
        { const unsigned hv = ibuf[ioff + l]; smin = min(smin, hv); }
        ioff += L;
    }
    // the columns after the last checkpoint
    {
        const int cl = nblk > 0 ? min((nblk * L - 1) / a.ck, a.nck) : 0;   // checkpoints this wave passed (wave-uniform)
        if (live) a.lsum[((int64_t)(r - a.read0) * (a.nck + 1) + cl) * L + l] = smin;
        if (nblk > 0) flush(cl, ioff / L);
        // (summaries of intervals the read never reached)
        for (int cc = cl + 1; cc <= a.nck; cc++)
            if (live) a.lsum[((int64_t)(r - a.read0) * (a.nck + 1) + cc) * L + l] = QINF;
    }
    // the read's screening minimum = the smallest summary
    if (live) {
        const unsigned *ls = a.lsum + (int64_t)(r - a.read0) * (a.nck + 1) * L + l;
        for (int cc = 0; cc <= a.nck; cc++) qmin = min(qmin, ls[(int64_t)cc * L]);
    }
    // the read's screening minimum for pass P; a sample outside the fixed-point range anywhere in the
    // read disqualifies the screening (reported as an infinite minimum)
#pragma unroll
    for (int d = 1; d < L; d <<= 1) {
        bad |= __shfl_xor(bad, d);
        qmin = min(qmin, (unsigned)__shfl_xor((int)qmin, d));
    }
    const unsigned bq = bad ? QINF : qmin;
    if (live && l == 0) a.qflag[r - a.read0] = (int32_t)bq;
    if (a.wrec_q) {                                 // epilogue: the candidate columns, for the first tier of pass P
        int jlo, jhi;
        candidate_columns<L>(a, r, n, bq, l, jlo, jhi);
        // candidates too far apart for one window: two clusters?  (group-uniform: every lane of the group holds jlo / jhi)
        int sjlo = 0, sjhi = -1;
        if (a.sib && live && bq < QSAFE && jhi >= jlo && jhi - jlo > a.wmax) {
            int alo, ahi, blo, bhi;
            candidate_columns<L>(a, r, n, bq, l, alo, ahi, jlo, jlo + a.wmax);
            candidate_columns<L>(a, r, n, bq, l, blo, bhi, jlo + a.wmax + 1);
            if (bhi >= blo && bhi - blo <= a.wmax) {
                // (lane 0 books the sibling's slot; the list is short -- its launches are sized for it -- and a read
                // that finds it full stays unscreened)
                int slot_ok = 0;
                if (l == 0) {
                    const int idx = atomicAdd(a.sib_cnt, 1);
                    if (idx < a.sib_cap) {
                        sibrec sb;
                        sb.r = r; sb.jlo = blo; sb.jhi = bhi; sb.pad = 0;
                        ((sibrec *)a.sib)[idx] = sb;
                        slot_ok = 1;
                    }
                }
                slot_ok = __shfl(slot_ok, lane - l);
                if (slot_ok) { jhi = ahi; sjlo = blo; sjhi = bhi; }
            }
        }
        if (live && l == 0) {
            wrec w;
            w.tbase = 0; w.jlo = jlo; w.jhi = jhi; w.flags = (sjhi >= sjlo) ? 4 : 0;      // 4: the read has a sibling
            ((wrec *)a.wrec_q)[r - a.read0] = w;
            // a read the window passes cannot take (no usable minimum, candidates too far apart) goes to the exact
            // retry -- which starts now, beside the window passes, instead of behind them
            const bool screened = (bq < QSAFE) && (jhi >= jlo) && (jhi - jlo <= a.wmax);
            if (a.early_cnt && n > 0 && !screened) a.early[atomicAdd(a.early_cnt, 1)] = r;
        }
    }
    if (timer && lane == 0) {
        a.clk[0] = __builtin_amdgcn_s_memtime() - tc0;
        a.clk[1] = __builtin_amdgcn_s_memrealtime() - tr0;
    }
}

// ---------------------------------------------------------------------------------------------
// pass P: candidate columns, restart point, fixed-point pre-roll
// ---------------------------------------------------------------------------------------------
template <int L, int R, int FEED>
__global__ __launch_bounds__(256)
void k_sdtw_p(const sdtw_kargs a)
{
    constexpr int G = 64 / L;
    constexpr int CKW = R + 2;

    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int g = lane / L, l = lane % L;
    int slot = wave * G + g;
    int nreads = a.nreads;
    if (a.wl_count) {                               // second tier: the list length is only known on the device
        nreads = min(*a.wl_count, a.nreads);
        if (a.total_ptr && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.total_ptr, nreads);   // (diagnostic)
        if (nreads <= 0) return;                    // (block-uniform)
    }
    const bool live = slot < nreads;
    if (!live) slot = nreads - 1;
    const int r = a.tier2 == 2 ? ((const sibrec *)a.sib)[slot].r : a.wl_list ? a.wl_list[slot] : a.read0 + slot;

    int n;
    double center = 0.0, scale = 1.0;
    const int16_t *s16 = nullptr;
    const double  *s64 = nullptr;
    if constexpr (FEED == SK_FEED_I16) {
        const sk_prep pr = a.prep[r];
        n = pr.n; center = pr.center; scale = pr.scale;
        s16 = (const int16_t *)a.samples + (int64_t)r * a.stride;
    } else if constexpr (FEED == SK_FEED_F64_NORM) {
        const sk_prep pr = a.prep[r];
        n = pr.n; center = pr.center; scale = pr.scale;
        s64 = (const double *)(((pr.flags & SK_IFLAG_INPLACE) && a.samples_raw) ? a.samples_raw : a.samples) + a.off[r];
    } else {
        n = (int)(a.off[r + 1] - a.off[r]);
        s64 = (const double *)a.samples + a.off[r];
    }
    if (!live) n = 0;

    // ---- candidate columns: screening cost within 2E of the screening minimum ----------------
    // The first tier finds them in pass Q's epilogue (the read's wave scans the interval minima it has just written:
    // latency that the other waves' sweeps hide); a second-tier launch looks again (its reads come from a list).
    const unsigned b = (unsigned)a.qflag[r - a.read0];   // the screening minimum (pass Q), QINF: not usable
    int jlo, jhi, clustered = 0;                    // clustered: this window holds one of two clusters of candidates
    if (!a.tier2) {
        const wrec q = ((const wrec *)a.wrec_q)[r - a.read0];
        jlo = q.jlo; jhi = q.jhi; clustered = q.flags & 4;
    } else if (a.tier2 == 2) {                      // a read's second cluster (sibling list, pass Q's epilogue)
        const sibrec q = ((const sibrec *)a.sib)[slot];
        jlo = q.jlo; jhi = q.jhi; clustered = 4;
    } else {
        candidate_columns<L>(a, r, n, b, l, jlo, jhi);
    }
    const bool screened = (n > 0) && (b < QSAFE) && (jhi >= jlo) && (jhi - jlo <= a.wmax);

    int tbase = 0, c0 = 0, npre = 0;
    if (screened) {
        const int tx = max(0, jlo - a.span);        // where the exact recurrence has to start
        c0 = tx / a.ck;
        if (c0 > a.nck) c0 = a.nck;
        tbase = c0 * a.ck;
        if (c0 > 0) npre = (tx - tbase) / L;        // whole blocks between the checkpoint and tx
    }
    const bool shortlane = l < a.P;
    const unsigned notfirst = (l == 0) ? 0u : 0xffffffffu;

    // ---- pre-roll: from the checkpoint to the block that holds tx, in fixed point --------------
    // Groups of one wave need different numbers of blocks: a group loads its checkpoint in the
    // iteration in which its turn starts (what it computed before on never-initialised state is
    // overwritten), so all groups finish together.
    unsigned Qs[R];                                 // screening state: R cells, lane l-1's row, diagonal
    unsigned botq = QINF, diagq = QINF;
#pragma unroll
    for (int k = 0; k < R; k++) Qs[k] = QINF;
    constexpr int CKH = (R + 3) / 2;                // (pass Q stores the high halves of the R + 2 state words, two to a dword)
    const unsigned *cp = a.ckq + (((int64_t)(r - a.read0) * a.nck + (c0 > 0 ? c0 - 1 : 0)) * L + l) * CKH;
    auto load_ckpt = [&]() {
        auto word = [&](int k) -> unsigned { const unsigned w2 = cp[k / 2]; return (k & 1) ? (w2 & 0xffff0000u) : (w2 << 16); };
#pragma unroll
        for (int k = 0; k < R; k++) Qs[k] = word(k);
        botq = word(R); diagq = word(R + 1);
    };
    int maxpre = npre;
#pragma unroll
    for (int d = L; d < 64; d <<= 1) maxpre = max(maxpre, __shfl_xor(maxpre, d));
    maxpre = __builtin_amdgcn_readfirstlane(maxpre);
    if (maxpre > 0) {
        const double inv_scale = 1.0 / scale;
        const int nlast = max(n - 1, 0);
        if (n == 0) { s16 = (const int16_t *)a.xlayq; s64 = (const double *)a.xlayq; }   // any valid address
        auto toq = [&](int idx) -> unsigned {       // the screening pass's image of sample idx
            const int ci = min(max(idx, 0), nlast);
            double v;
            if constexpr (FEED == SK_FEED_I16)           v = ((double)s16[ci] - center) * inv_scale;
            else if constexpr (FEED == SK_FEED_F64_NORM) v = (s64[ci] - center) * inv_scale;
            else                                         v = s64[ci];
            return (idx >= 0 && idx < n && fabs(v) < QLIM) ? qimg(v) : QINF;
        };
        unsigned xq[R];
#pragma unroll
        for (int k = 0; k < R; k++) xq[k] = a.xlayq[l * R + k];
        unsigned Qn[R];
#pragma unroll
        for (int k = 0; k < R; k++) Qn[k] = QINF;
        const unsigned smask = shortlane ? 0xffffffffu : 0u;
        __shared__ unsigned lds_p[4][3 * 64];       // the sample ring of pass Q (no last-row buffer here)
        unsigned *ybuf = lds_p[threadIdx.x >> 6] + g * 3 * L;
        unsigned Fq = QINF;
        const int startblk = maxpre - npre;         // my group's first block
        auto qstep = [&](const unsigned (&old)[R], unsigned (&nw)[R], unsigned yq) {
            const unsigned upq = shr0<L>(botq, notfirst);
            qcolumn<R>(old, nw, xq, yq, diagq, upq);
            diagq = upq;
            if constexpr (R >= 2) asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe4" : "=v"(botq) : "v"(nw[R - 2]), "v"(nw[R - 1]), "v"(smask));
            else                  asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe4" : "=v"(botq) : "v"(upq), "v"(nw[0]), "v"(smask));
        };
        for (int pb = 0; pb < maxpre; pb++) {
            const int t0 = tbase + (pb - startblk) * L;
            if (pb == startblk && npre > 0) {
                load_ckpt();
                const unsigned prev = toq(t0 - L + l);   // the block before: lane l needs sample t - l
                if (pb & 1) ybuf[l] = prev; else ybuf[L + l] = prev;
                Fq = toq(t0 + l);
            }
            const unsigned Fnext = toq(t0 + L + l);
            const unsigned *yr;
            if (pb & 1) { ybuf[L + l] = Fq; yr = ybuf + L - l; }
            else        { ybuf[l] = Fq; ybuf[2 * L + l] = Fq; yr = ybuf + 2 * L - l; }
#pragma unroll 1
            for (int q = 0; q < L; q += 2) {        // L is even: after two steps the roles are back
                const unsigned y0 = yr[q], y1 = yr[q + 1];
                qstep(Qs, Qn, y0);
                qstep(Qn, Qs, y1);
            }
            Fq = Fnext;
        }
    }
    if (c0 > 0 && npre == 0) load_ckpt();
    tbase += npre * L;

    if (live) {
        if (c0 > 0) {
            unsigned *ws = a.wstate + ((int64_t)slot * L + l) * CKW;
#pragma unroll
            for (int k = 0; k < R; k++) ws[k] = Qs[k];
            ws[R] = botq; ws[R + 1] = diagq;
        }
        if (l == 0) {
            wrec w;
            w.tbase = tbase; w.jlo = jlo; w.jhi = jhi;
            w.flags = (screened ? 1 : 0) | (c0 > 0 ? 2 : 0) | clustered;
            ((wrec *)a.wrec)[slot] = w;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// pass W: the exact window
// ---------------------------------------------------------------------------------------------
// (R > 16: the lane's motif rows stay in LDS -- one ds_read_b64 per cell on the idle LDS pipe -- instead of 2R more
// VGPRs, which would leave two waves per SIMD; the 8 lanes of a group read 8 addresses R doubles apart, all groups
// the same ones: no bank conflict)
template <int L, int R, int FEED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(R <= 13 ? 4 : (R <= 26 ? 3 : 2), 8)))
void k_sdtw_w(const sdtw_kargs a)
{
    constexpr int G = 64 / L;
    constexpr bool XLDS = R > 16;
    constexpr int SHR = (L == 64) ? DPP_WAVE_SHR1 : DPP_ROW_SHR1;
    constexpr int CKW = R + 2;
    constexpr int U = (L < 16) ? L : 8;             // steps per unrolled run (samples prefetched from LDS)
    const double INF = __builtin_huge_val();

    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int g = lane / L, l = lane % L;
    int slot = wave * G + g;
    int nreads = a.nreads;
    if (a.wl_count) {                               // second tier: the list length is only known on the device
        nreads = min(*a.wl_count, a.nreads);
        if (nreads <= 0) return;                    // (block-uniform)
    }
    const bool live = slot < nreads;
    if (!live) slot = nreads - 1;
    const int r = a.tier2 == 2 ? ((const sibrec *)a.sib)[slot].r : a.wl_list ? a.wl_list[slot] : a.read0 + slot;

    int n, flags = 0;
    double center = 0.0, scale = 1.0;
    const int16_t *s16 = nullptr;
    const double  *s64 = nullptr;
    if constexpr (FEED == SK_FEED_I16) {
        const sk_prep pr = a.prep[r];
        n = pr.n; flags = pr.flags & SK_FLAG_PUBLIC; center = pr.center; scale = pr.scale;
        s16 = (const int16_t *)a.samples + (int64_t)r * a.stride;
    } else if constexpr (FEED == SK_FEED_F64_NORM) {
        const sk_prep pr = a.prep[r];
        n = pr.n; flags = pr.flags & SK_FLAG_PUBLIC; center = pr.center; scale = pr.scale;
        s64 = (const double *)(((pr.flags & SK_IFLAG_INPLACE) && a.samples_raw) ? a.samples_raw : a.samples) + a.off[r];
    } else {
        n = (int)(a.off[r + 1] - a.off[r]);
        if (n == 0) flags = SK_FLAG_EMPTY;
        s64 = (const double *)a.samples + a.off[r];
    }
    if (!live) n = 0;

    const wrec rec = ((const wrec *)a.wrec)[slot];
    const bool screened = live && (rec.flags & 1);
    const bool restored = screened && (rec.flags & 2);
    const int tbase = screened ? rec.tbase : 0;
    const int jlo = rec.jlo, jhi = rec.jhi;
    const int tlast = screened ? jhi + L - 1 : -1;
    const bool shortlane = l < a.P;
    const unsigned notfirst = (l == 0) ? 0u : 0xffffffffu;
    const unsigned first = ~notfirst;

    int nsteps = tlast - tbase + 1;
    if (nsteps < 0) nsteps = 0;
    const int own_steps = nsteps;                   // what MY read asks for (before the wavefront's maximum)
#pragma unroll
    for (int d = L; d < 64; d <<= 1) nsteps = max(nsteps, __shfl_xor(nsteps, d));
    nsteps = __builtin_amdgcn_readfirstlane(nsteps);
    const int nblk = (nsteps + L - 1) / L;
    if (a.wsteps) {                                 // diagnostic: the pass's issue roof is counted in these (bench.py)
        if (lane == 0 && nblk > 0) atomicAdd(&a.wsteps[0], (unsigned long long)(nblk * L));
        if (live && l == 0 && own_steps > 0) atomicAdd(&a.wsteps[1], (unsigned long long)own_steps);
    }

    double x[XLDS ? 1 : R];
    constexpr int RP = R | 1;                       // odd row pitch: the lanes of a group land in different LDS banks
    __shared__ double lds_x[XLDS ? L * RP : 1];     // (R = 32, L = 16: a pitch of 256 bytes put all 16 on one bank)
    if constexpr (XLDS) {
        for (int i = threadIdx.x; i < L * R; i += blockDim.x) lds_x[(i / R) * RP + i % R] = a.xlay[i];
        __syncthreads();
    } else {
#pragma unroll
        for (int k = 0; k < R; k++) x[k] = a.xlay[l * R + k];
    }
    const double *xl = lds_x + (XLDS ? l * RP : 0);

    const int nlast = max(n - 1, 0);
    if (n == 0) { s16 = (const int16_t *)a.xlay; s64 = (const double *)a.xlay; }       // any valid address
    // (unconditional load from a clamped index: a load under a branch is waited for at the join)
    auto fetch = [&](int idx) -> double {
        const int ci = min(max(idx, 0), nlast);
        double v;
        if constexpr (FEED == SK_FEED_I16)           v = ((double)s16[ci] - center) / scale;
        else if constexpr (FEED == SK_FEED_F64_NORM) v = (s64[ci] - center) / scale;
        else                                         v = s64[ci];
        return (idx < 0 || idx >= n) ? INF : v;
    };
    // lower bound (in signal units) of a cell whose screening cost is q
    auto lb = [&](unsigned q) -> double { return (double)(q > a.qerr ? q - a.qerr : 0u) * QUNIT; };

    double D[R];
    int    S[R];
#pragma unroll
    for (int k = 0; k < R; k++) { D[k] = INF; S[k] = -1; }
    double botD = (R == 1 && l == 0 && shortlane) ? 0.0 : INF;
    int    botS = (R == 1 && l == 0 && shortlane) ? 0 : -1;
    double diagD = (l == 0) ? 0.0 : INF;
    int    diagS = (l == 0) ? tbase : -1;
    if (restored) {
        const unsigned *ws = a.wstate + ((int64_t)slot * L + l) * CKW;
#pragma unroll
        for (int k = 0; k < R; k++) D[k] = lb(ws[k]);
        botD = lb(ws[R]);
        if (l > 0) diagD = lb(ws[R + 1]);           // lane 0's diag is the virtual row: exactly 0
        if (R == 1 && l == 0 && shortlane) { botD = 0.0; botS = tbase; }   // forwards the virtual row
    }
    double best = INF;  int bestS = -1, bestJ = -1;
    // the running argmin is only wanted where lane L-1 sits on a candidate column -- the last few steps of the window:
    // first step (relative to tbase) at which some read group of the wave needs it
    int s0 = screened ? max(jlo - tbase + L - 1, 0) : 0x7fffffff;
#pragma unroll
    for (int d = L; d < 64; d <<= 1) s0 = min(s0, __shfl_xor(s0, d));
    s0 = __builtin_amdgcn_readfirstlane(s0);

    // The normalised samples go through an LDS ring per read group, as in pass Q (doubles here):
    //   ybuf[0,L) even blocks | ybuf[L,2L) odd blocks | ybuf[2L,3L) copy of [0,L)
    __shared__ double lds_w[4][3 * 64];
    double *ybuf = lds_w[threadIdx.x >> 6] + g * 3 * L;
    ybuf[L + l] = fetch(tbase - L + l);             // the block before the first: lane l starts on column tbase - l
    double F = fetch(tbase + l);
    for (int blk = 0; blk < nblk; blk++) {
        const double Fnext = fetch(tbase + (blk + 1) * L + l);
        const double *yr;
        if (blk & 1) { ybuf[L + l] = F; yr = ybuf + L - l; }
        else         { ybuf[l] = F; ybuf[2 * L + l] = F; yr = ybuf + 2 * L - l; }
#pragma unroll 1
        for (int qq = 0; qq < L; qq += U) {
            double yv[U];
#pragma unroll
            for (int q = 0; q < U; q++) yv[q] = yr[qq + q];
#pragma unroll
            for (int q = 0; q < U; q++) {
                const int t = tbase + blk * L + qq + q;
                const double y = yv[q];
                double upD;  int upS;
                if constexpr (L >= 16) {
                    upD = dpp_f64<SHR>(0.0, botD);
                    upS = dpp_i32<SHR>(t + 1, botS);
                } else {
                    const unsigned lo = shr0<L>((unsigned)__double2loint(botD), notfirst);
                    const unsigned hi = shr0<L>((unsigned)__double2hiint(botD), notfirst);
                    upD = __hiloint2double((int)hi, (int)lo);
                    upS = (int)(shr0<L>((unsigned)botS, notfirst) | (first & (unsigned)(t + 1)));
                }
                double dgD = diagD;  int dgS = diagS;
                double uD = upD;     int uS = upS;
#pragma unroll
                for (int k = 0; k < R; k++) {
                    const double lfD = D[k];  const int lfS = S[k];
                    const double xk = XLDS ? xl[k] : x[k];
                    const double c = fabs(xk - y);
                    // (no NaN can reach this pass -- pass Q sends reads with a non-finite or out-of-range sample to the
                    // exact single pass -- so v_min_f64 IS the ternary select of the reference's min3; written as a
                    // select the compiler spends two v_cndmask per double: 10 instructions per cell instead of 8)
                    const bool lt1 = lfD < dgD;
                    const double m1 = vmin(lfD, dgD);
                    const int    s1 = lt1 ? lfS : dgS;
                    const bool lt2 = uD < m1;
                    const double m = vmin(uD, m1);
                    const int    s = lt2 ? uS : s1;
                    const double nd = c + m;
                    dgD = lfD;  dgS = lfS;
                    D[k] = nd;  S[k] = s;
                    uD = nd;    uS = s;
                }
                diagD = upD;  diagS = upS;
                if constexpr (R >= 2) {
                    botD = shortlane ? D[R - 2] : D[R - 1];
                    botS = shortlane ? S[R - 2] : S[R - 1];
                } else {
                    botD = shortlane ? upD : D[0];
                    botS = shortlane ? upS : S[0];
                }
                if (blk * L + qq + q >= s0) {               // (wave-uniform)
                    const int j = t - l;
                    if (j >= jlo && j <= jhi && D[R - 1] < best) { best = D[R - 1]; bestS = S[R - 1]; bestJ = j; }
                }
            }
        }
        F = Fnext;
    }

    // Guard (round 5).  The certificate is a theorem ABOUT the screening values: it assumes |Dq - D| <= E in every
    // cell, which is argued (E's derivation, the sample image) and which nothing used to check -- the round-4 fuzz found
    // it silently false for float64 reads after three rounds of green tests.  A certified winner is an exact cell of the
    // last row, so the assumption can be tested where it is about to be relied on: the exact distance must lie within
    // E of the screening value of ITS OWN column, and not more than E above the screening minimum of the whole row
    // (the column that holds that minimum has an exact cost of at most minimum + E, and the winner beats it).  Two loads
    // per read.  A violation is counted, raises the call's alarm (the whole call is then redone by the exact pass,
    // sk_launch_sdtw) and sends the read to the exact retry.
    auto premise_holds = [&](double bestD, int j) -> bool {
        if (!a.guard) return true;
        const unsigned lq = a.lastq[(int64_t)(r - a.read0) * a.lq_stride + j + L];   // (rows start L early)
        const unsigned bq = (unsigned)a.qflag[r - a.read0];
        const double u = bestD * QSCALE, e = (double)a.qerr;
        // (one of two clusters: the row's minimum may sit in the other one -- this window's candidates only promise a
        // screening cost within 2 E of it, hence an exact one within 3 E)
        const double e2 = (rec.flags & 4) ? 3.0 * e : e;
        const bool ok = fabs(u - (double)lq) <= e && u <= (double)bq + e2;
        if (!ok) { atomicAdd(&a.guard[SK_GUARD_VIOL], 1); atomicAdd(&a.guard[SK_GUARD_ALARM], 1); }
        return ok;
    };
    if (live && l == L - 1) {
        sk_hit h;
        h.n = n; h.flags = flags;
        // tuning / sensitivity runs only: send a share of the reads to the exact retry whatever the window found
        const bool forced = a.force_retry && (((unsigned)r * 2654435761u) >> 22) < (unsigned)a.force_retry;
        bool certified = false;
        if (a.tier2 == 2) {
            // a second cluster: its result goes to the sibling's own record, the combine kernel merges (k_sib_combine)
            h.dist = best; h.start = bestS; h.end = bestJ;
            h.n = (n > 0 && screened && bestS >= 0 && !forced && premise_holds(best, bestJ)) ? 1 : 0;   // 1: certified
            a.sib_out[slot] = h;
        } else if (n <= 0) {
            h.dist = __builtin_nan(""); h.start = -1; h.end = -1;
            a.out[r] = h;
        } else if (screened && bestS >= 0 && !forced && (certified = premise_holds(best, bestJ))) {
            h.dist = best; h.start = bestS; h.end = bestJ;
            a.out[r] = h;
        } else if (!screened && a.early_cnt && !a.tier2) {
            // (pass Q listed this read for the early exact retry, which writes its record -- perhaps right now.  In the second
            // tier an unscreened read is one whose FIRST cluster did not certify -- the tier looks at all its candidates
            // again and finds them too far apart: it falls through to the exact pass)
        } else {
            h.dist = __builtin_nan(""); h.start = -1; h.end = -1;    // overwritten by a later pass
            a.out[r] = h;
            const bool violated = screened && bestS >= 0 && !forced && !certified;     // (a wider look-back cannot mend that)
            if (screened && a.soft && !forced && !violated) a.soft[atomicAdd(a.soft_cnt, 1)] = r;   // path wider than this look-back: next tier
            else                                            a.retry[atomicAdd(a.retry_cnt, 1)] = r; // the exact single pass
        }
    }
}

typedef void (*sdtw_fn)(const sdtw_kargs);

template <int L, int FEED, int WHICH>
sdtw_fn pick_r(int R)
{
#define SK_CASE(RR) case RR: return WHICH == 0 ? (sdtw_fn)k_sdtw_q<L, RR, FEED, false> : \
                                   WHICH == 3 ? (sdtw_fn)k_sdtw_q<L, RR, FEED, (RR >= 2)> : \
                                   WHICH == 1 ? (sdtw_fn)k_sdtw_p<L, RR, FEED> : (sdtw_fn)k_sdtw_w<L, RR, FEED>;
    switch (R) {
        SK_CASE(1) SK_CASE(2) SK_CASE(3) SK_CASE(4) SK_CASE(5) SK_CASE(6) SK_CASE(7) SK_CASE(8)
        SK_CASE(9) SK_CASE(10) SK_CASE(11) SK_CASE(12) SK_CASE(13) SK_CASE(14) SK_CASE(15) SK_CASE(16)
    }
    if constexpr (L != 64) {
        switch (R) {
            SK_CASE(17) SK_CASE(18) SK_CASE(19) SK_CASE(20) SK_CASE(21) SK_CASE(22) SK_CASE(23) SK_CASE(24)
            SK_CASE(25) SK_CASE(26) SK_CASE(27) SK_CASE(28) SK_CASE(29) SK_CASE(30) SK_CASE(31) SK_CASE(32)
        }
    }
#undef SK_CASE
    return nullptr;
}

template <int WHICH, int FEED>
sdtw_fn pick_l(int L, int R)
{
    if (L == 8)  return pick_r<8, FEED, WHICH>(R);
    if (L == 16) return pick_r<16, FEED, WHICH>(R);
    return pick_r<64, FEED, WHICH>(R);
}

} // namespace

// (which: 0 = Q, 1 = P, 2 = W)
#if SK_SDTWQ_FEED == 0
void *sk_sdtwq_pick_feed0(int which, int L, int R)
#elif SK_SDTWQ_FEED == 1
void *sk_sdtwq_pick_feed1(int which, int L, int R)
#else
void *sk_sdtwq_pick_feed2(int which, int L, int R)
#endif
{
    switch (which) {
    case 0: return (void *)pick_l<0, SK_SDTWQ_FEED>(L, R);
    case 1: return (void *)pick_l<1, SK_SDTWQ_FEED>(L, R);
    case 3: return (void *)pick_l<3, SK_SDTWQ_FEED>(L, R);       // pass Q without short lanes
    default: return (void *)pick_l<2, SK_SDTWQ_FEED>(L, R);
    }
}

#ifdef SK_SDTWQ_MAIN
void *sk_sdtwq_pick_feed1(int which, int L, int R);
void *sk_sdtwq_pick_feed2(int which, int L, int R);

// lanes per read of the screening scheme for an N-point motif (SK_DTW_QL = 8 / 16 / 64: A/B runs).  The long lanes
// of L = 8 pay off once the batch fills the chip with them (8 reads per wavefront: measured 1.75 / 1.45 / 1.50 ms
// with 8 / 16 / 64 lanes at 10 000 reads x 163 points, 2.06 / 1.80 / 2.45 at 20 000 x 200, level from 40 000 on,
// 8 ahead at 1 M).
static void screen_layout(int N, int64_t nreads, int64_t maxlen, int *L, int *R)
{
    // (round 6, float64 reads of 20 000 / 37 000 samples vs 163 points: 50 000 reads 15.9 ms with 8 lanes against 15.5 with
    // 16, 25 000 reads 15.3 against 14.0 -- while 62 500 reads of 4 000 samples vs 200 points take 4.39 ms with 8 lanes and
    // 4.54 with 16: long reads switch at 65 536 reads, short ones at 49 152 as before)
    int l = (N <= 8 * 32 && nreads >= (maxlen > 8192 ? 65536 : 49152)) ? 8 : (N <= 16 * 32) ? 16 : 64;
    if (const char *e = sk_tune("SK_DTW_QL")) {
        const int v = atoi(e);
        if ((v == 8 && N <= 8 * 32) || (v == 16 && N <= 16 * 32) || v == 64) l = v;
    }
    *L = l;
    *R = (N + l - 1) / l;
}

// ---------------------------------------------------------------------------------------------
// the order in which the window passes take a chunk's reads
// ---------------------------------------------------------------------------------------------
// The read groups of a wavefront step together: pass P runs as many pre-roll blocks, pass W as many window blocks, as
// the neediest of its 8 (4, 1) reads asks for.  In file order that is 13.6 pre-roll blocks where a read needs 7.3 on
// average (the distance from the checkpoint to the window start is uniform in [0, ck)) and 29.8 window blocks for 28.2
// (1 M C4 reads).  A counting sort of the reads by (window blocks, pre-roll blocks) -- three small kernels on pass Q's
// epilogue records -- puts reads with the same needs into the same wavefront; nothing else changes, the reads are
// independent and every pass addresses them through the list.
constexpr int ORDER_BINS = 1024;                       // 64 classes of window blocks x 16 of pre-roll blocks

struct order_args {
    const wrec    *rec;                                // pass Q's epilogue records, [read - read0]
    const int32_t *qflag;
    int nreads, read0, span, ck, nck, L, wmax;
    int32_t *hist;                                     // [ORDER_BINS] counts, then (in place) running cursors
    int32_t *order;                                    // out: read indices (absolute), sorted by key
};

__device__ __forceinline__ int order_key(const order_args &a, int i)
{
    const wrec q = a.rec[i];
    const unsigned b = (unsigned)a.qflag[i];
    // Longest first (round 5): the sorted list used to end with the reads that need the most window blocks, so the last
    // wavefronts to start were the longest-running ones; in descending order the short ones fill the tail of the launch
    // (window passes 2.03 -> 1.78 ms at 125 000 reads per call, 12.1 -> 11.9 at 1 M).
    if (!((b < QSAFE) && (q.jhi >= q.jlo) && (q.jhi - q.jlo <= a.wmax))) return ORDER_BINS - 1;   // not screened: no work in P / W, last
    const int tx = max(0, q.jlo - a.span);             // (pass P's arithmetic)
    int c0 = tx / a.ck;
    if (c0 > a.nck) c0 = a.nck;
    const int npre = c0 > 0 ? (tx - c0 * a.ck) / a.L : 0;
    const int tbase = c0 * a.ck + npre * a.L;
    const int nblk = (q.jhi + a.L - tbase + a.L - 1) / a.L;
    return ORDER_BINS - 1 - (min(max(nblk, 1), 63) * 16 + min(npre, 15));
}

// (1 024 threads x 4 reads per workgroup: the global reservations -- a few dozen hot bins -- serialise in the L2, so
// there should be few of them: 4 096 reads per reservation round instead of 256 took the two kernels from 53 to a
// few microseconds per 500 000 reads)
constexpr int ORDER_TPB = 1024, ORDER_IPT = 4;

__global__ __launch_bounds__(ORDER_TPB) void k_order_count(const order_args a)
{
    __shared__ int h[ORDER_BINS];
    for (int b = threadIdx.x; b < ORDER_BINS; b += ORDER_TPB) h[b] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < ORDER_IPT; u++) {
        const int i = (blockIdx.x * ORDER_IPT + u) * ORDER_TPB + threadIdx.x;
        if (i < a.nreads) atomicAdd(&h[order_key(a, i)], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < ORDER_BINS; b += ORDER_TPB)
        if (h[b]) atomicAdd(&a.hist[b], h[b]);
}

__global__ __launch_bounds__(ORDER_BINS) void k_order_scan(int32_t *hist)   // counts -> exclusive prefix sums, in place
{
    __shared__ int s[ORDER_BINS];
    const int t = threadIdx.x;
    const int v = hist[t];
    s[t] = v;
    __syncthreads();
    for (int d = 1; d < ORDER_BINS; d <<= 1) {
        const int add = t >= d ? s[t - d] : 0;
        __syncthreads();
        s[t] += add;
        __syncthreads();
    }
    hist[t] = s[t] - v;
}

__global__ __launch_bounds__(ORDER_TPB) void k_order_scatter(const order_args a)
{
    __shared__ int h[ORDER_BINS];                      // the block's counts, then its base per bin
    for (int b = threadIdx.x; b < ORDER_BINS; b += ORDER_TPB) h[b] = 0;
    __syncthreads();
    int key[ORDER_IPT], rank[ORDER_IPT];
#pragma unroll
    for (int u = 0; u < ORDER_IPT; u++) {
        const int i = (blockIdx.x * ORDER_IPT + u) * ORDER_TPB + threadIdx.x;
        key[u] = 0; rank[u] = 0;
        if (i < a.nreads) { key[u] = order_key(a, i); rank[u] = atomicAdd(&h[key[u]], 1); }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < ORDER_BINS; b += ORDER_TPB)
        if (h[b]) h[b] = atomicAdd(&a.hist[b], h[b]);  // one global reservation per non-empty bin and block
    __syncthreads();
#pragma unroll
    for (int u = 0; u < ORDER_IPT; u++) {
        const int i = (blockIdx.x * ORDER_IPT + u) * ORDER_TPB + threadIdx.x;
        if (i < a.nreads) a.order[h[key[u]] + rank[u]] = a.read0 + i;
    }
}

// second clusters -> the reads' records (struct sibrec)
__global__ void k_sib_combine(const sibrec *sib, const int32_t *cnt, int cap, const sk_hit *so, sk_hit *out,
                              int32_t *retry, int32_t *retry_cnt, int32_t *nsib_total)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s == 0 && nsib_total) atomicAdd(nsib_total, min(*cnt, cap));     // (diagnostic: sk_last_dtw_guard out[6])
    if (s >= min(*cnt, cap)) return;
    const int r = sib[s].r;
    sk_hit A = out[r];
    const sk_hit B = so[s];
    if (A.start < 0) return;                            // the first cluster did not certify: the read is on its way to the exact pass
    if (B.n == 1) {
        if (B.dist < A.dist) { A.dist = B.dist; A.start = B.start; A.end = B.end; out[r] = A; }   // (a tie keeps the lower column)
    } else {                                            // the second cluster could not be certified: exact pass after all
        A.dist = __builtin_nan(""); A.start = -1; A.end = -1;
        out[r] = A;
        retry[atomicAdd(retry_cnt, 1)] = r;
    }
}

// Screening + certified window over all reads; fills out[] and the retry list (device).
// The caller (sk_launch_sdtw) runs the exact pass on the listed reads.
// span / span2: look-back of the window pass's first tier (every read) and of its second tier (the reads whose
// optimal path turned out wider than `span`); span2 <= span: one tier only.
int sk_launch_sdtw_screen(sk_ctx *c, const sk_sdtw_args *a, int ck, int span, int span2,
                          int32_t *d_retry_cnt, int32_t *d_retry, int32_t *d_early_cnt, int32_t *d_early)
{
    const int N = a->nmotif;
    int L, R;
    screen_layout(N, a->nreads, a->max_len, &L, &R);
    const int P = L * R - N;
    // quantised motif and the exact one in this scheme's per-lane layout (the exact kernels keep their own);
    // resident like them (the caller invalidates when the motif changes), so a call makes no host-side synchronisation
    int rc;
    if (!c->motifq_valid || c->motifq_L != L) {
        SK_HIP(hipStreamSynchronize(c->stream));        // an earlier launch may still read the old one
        std::vector<unsigned> &layq = c->motifq_host;
        std::vector<double> &layw = c->motifw_host;
        layq.assign((size_t)L * R, 0x80000000u);
        layw.assign((size_t)L * R, 0.0);
        int row = 0;
        for (int l = 0; l < L; l++) {
            const int cnt = (l < P) ? R - 1 : R;
            for (int k = 0; k < cnt; k++, row++) {
                layq[(size_t)l * R + k] = (unsigned)((int)rint(a->motif[row] * QSCALE)) + 0x80000000u;
                layw[(size_t)l * R + k] = a->motif[row];
            }
        }
        if ((rc = sk_reserve(c, &c->motifq, layq.size() * sizeof(unsigned)))) return rc;
        if ((rc = sk_reserve(c, &c->motifw, layw.size() * sizeof(double)))) return rc;
        SK_HIP(hipMemcpyAsync(c->motifq.p, layq.data(), layq.size() * sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
        SK_HIP(hipMemcpyAsync(c->motifw.p, layw.data(), layw.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        c->motifq_valid = true;
        c->motifq_L = L;
    }

    const int64_t maxlen = a->max_len;
    const int nck = (int)((maxlen + L - 1) / ck);
    const size_t lq_stride = (size_t)((maxlen + 3 * L + 7) & ~(int64_t)3);      // L columns of padding in front, 2 L behind
    const size_t state_bytes = (size_t)L * (R + 2) * sizeof(unsigned);
    const size_t ck_bytes = (size_t)L * ((R + 3) / 2) * sizeof(unsigned);          // a checkpoint: the state's high halves
    const size_t per_read = (size_t)(nck > 0 ? nck : 1) * ck_bytes + lq_stride * sizeof(unsigned) + state_bytes +
                            (size_t)(nck + 1) * L * sizeof(unsigned) + 64;
    int64_t chunk;
    while (true) {                                      // (a device short of memory: halve the budget and try again)
        chunk = sk_dtw_chunk_reads(per_read, a->nreads);
        rc = sk_reserve(c, &c->ckpt, (size_t)chunk * (size_t)(nck > 0 ? nck : 1) * ck_bytes);
        if (!rc) rc = sk_reserve(c, &c->lastq, (size_t)chunk * lq_stride * sizeof(unsigned));
        if (!rc) rc = sk_reserve(c, &c->wstate, (size_t)chunk * state_bytes);
        if (!rc) rc = sk_reserve(c, &c->lsum, (size_t)chunk * (size_t)(nck + 1) * L * sizeof(unsigned));
        if (rc != SK_ERR_NOMEM || chunk <= 1024) break;
        sk_dtw_scratch_shrink(0);
    }
    if (rc) return rc;
    if ((rc = sk_reserve(c, &c->qflag, (size_t)chunk * sizeof(int32_t)))) return rc;
    if ((rc = sk_reserve(c, &c->wrec, (size_t)chunk * sizeof(wrec)))) return rc;
    if ((rc = sk_reserve(c, &c->wrecq, (size_t)chunk * sizeof(wrec)))) return rc;
    // the window passes take large chunks in sorted order (SK_DTW_NOSORT=1: file order; small ones: not worth 3 launches)
    int sort_min = 32768;
    if (const char *e = sk_tune("SK_DTW_SORT_MIN")) { const int v = atoi(e); if (v > 0) sort_min = v; }   // (tests: small batches too)
    const bool sorted = chunk >= sort_min && sk_tune("SK_DTW_NOSORT") == nullptr;
    if (sorted && (rc = sk_reserve(c, &c->order, ((size_t)chunk + ORDER_BINS) * sizeof(int32_t)))) return rc;
    const bool tiers = span2 > span;
    if (tiers && (rc = sk_reserve(c, &c->wsoft, ((size_t)chunk + 1) * sizeof(int32_t)))) return rc;
    const bool siblings = sk_tune("SK_DTW_NO_SIBLINGS") == nullptr;
    // (their list is short: a launch sized for the whole chunk costs 0.2 ms in workgroups that only look at the count)
    const int sib_cap = (int)(chunk / 64 + 1024);
    if (siblings) {
        if ((rc = sk_reserve(c, &c->sib, 16 + (size_t)sib_cap * sizeof(sibrec)))) return rc;
        if ((rc = sk_reserve(c, &c->sibout, (size_t)sib_cap * sizeof(sk_hit)))) return rc;
        if ((rc = sk_reserve(c, &c->sibstate, (size_t)sib_cap * (state_bytes + sizeof(wrec))))) return rc;
        if (c->stream3 && !c->ev_s[0])
            for (int i = 0; i < 2; i++) SK_HIP(hipEventCreateWithFlags(&c->ev_s[i], hipEventDisableTiming));
    }

    typedef void *(*pick_fn)(int, int, int);
    const pick_fn pk = a->feed == SK_FEED_I16 ? (pick_fn)sk_sdtwq_pick_feed0
                     : a->feed == SK_FEED_F64_NORM ? (pick_fn)sk_sdtwq_pick_feed1 : (pick_fn)sk_sdtwq_pick_feed2;
    sdtw_fn fq = (sdtw_fn)pk(P == 0 ? 3 : 0, L, R), fp = (sdtw_fn)pk(1, L, R), fw = (sdtw_fn)pk(2, L, R);
    if (!fq || !fp || !fw) return sk_fail(SK_ERR_UNSUPPORTED, "no screening kernel for L=%d R=%d", L, R);

    sdtw_kargs k;
    memset(&k, 0, sizeof k);
    k.samples = a->samples; k.samples_raw = a->samples_raw; k.stride = a->stride; k.off = a->off; k.prep = a->prep;
    k.xlay = (const double *)c->motifw.p; k.xlayq = (const unsigned *)c->motifq.p; k.P = P; k.out = a->out;
    k.nck = nck; k.ck = ck; k.span = span; k.retry = d_retry; k.retry_cnt = d_retry_cnt;
    k.ckq = (unsigned *)c->ckpt.p; k.lastq = (unsigned *)c->lastq.p; k.lq_stride = (int64_t)lq_stride;
    k.qflag = (int32_t *)c->qflag.p; k.lsum = (unsigned *)c->lsum.p;
    k.wstate = (unsigned *)c->wstate.p; k.wrec = c->wrec.p; k.wrec_q = c->wrecq.p;
    k.early_cnt = d_early_cnt; k.early = d_early;
    k.qerr = (unsigned)(N + maxlen + 2);
    k.wmax = 4 * ck;
    k.lds_wave_words = (64 / L) * ck;                   // pass Q, per wave: the interval's last-row values of its read groups
    k.guard = sk_tune("SK_DTW_NOGUARD") ? nullptr : (int32_t *)c->dtwcnt.p + 8;
    k.wsteps = (unsigned long long *)((char *)c->dtwcnt.p + 64);
    if (const char *e = sk_tune("SK_DTW_HOLE")) {                 // tests: a known hole back in, for the guard to find
        k.hole = strcmp(e, "qerr1") == 0 ? SK_HOLE_QERR1 : strcmp(e, "fma64") == 0 ? SK_HOLE_FMA64
               : strcmp(e, "fma64x") == 0 ? SK_HOLE_FMA64_UNGUARDED : SK_HOLE_NONE;
        if (k.hole == SK_HOLE_QERR1) k.qerr = 1;                  // "E = 1": lower bounds that are none, too few candidates
    }
    if (const char *e = sk_tune("SK_DTW_FORCE_RETRY_PM")) {       // sensitivity runs: per-mille of reads sent to the retry
        const int pm = atoi(e);
        if (pm > 0) k.force_retry = pm >= 1000 ? 1024 : (pm * 1024 + 999) / 1000;
    }

    // filter + medmad fused into pass Q (the caller checked the limits: sk_sdtw_fuse_ok)
    size_t fz_lds = 0;
    const sk_prep_fuse fz = a->fuse ? *a->fuse : sk_prep_fuse();
    if (a->fuse) {                                      // (or the prologue's histogram / its LDS copy of the compacted read)
        const int need = fz.mode == SK_PREP_ZSCALE ? (int)((a->stride * 2 + 15) / 16 * 4) : (((fz.hi - fz.lo - 1) + 3) & ~3);
        if (need > k.lds_wave_words) k.lds_wave_words = need;
    }
    fz_lds = (size_t)4 * (size_t)k.lds_wave_words * sizeof(unsigned);
    // (64 KB per workgroup without an opt-in attribute, 5 KB of it static: only the tuning variable SK_DTW_CK can ask
    // for more -- L = 8 with ck >= 512 -- and gets a message instead of a failed launch)
    if (fz_lds + 5 * 1024 > 64 * 1024)
        return sk_fail(SK_ERR_INVALID, "SK_DTW_CK=%d with %d lanes per read needs %zu bytes of LDS per workgroup (limit 65536): "
                       "use a smaller multiple of 64", ck, L, fz_lds + 5 * 1024);

    const size_t nchunks = (size_t)((a->nreads + chunk - 1) / chunk);
    while (c->evpool.size() < 3 * nchunks) {
        hipEvent_t e;
        SK_HIP(hipEventCreate(&e));
        c->evpool.push_back(e);
    }
    c->prof_chunks = 0;
    const int reads_per_block = 4 * (64 / L);
    for (int64_t r0 = 0; r0 < a->nreads; r0 += chunk) {
        k.read0 = (int)r0;
        k.nreads = (int)((a->nreads - r0 < chunk) ? a->nreads - r0 : chunk);
        const int grid = (k.nreads + reads_per_block - 1) / reads_per_block;
        if (a->fuse) {
            k.fz_raw = fz.raw; k.fz_len = fz.len; k.fz_lo = fz.lo; k.fz_hi = fz.hi;
            k.fz_mode = fz.mode == SK_PREP_ZSCALE ? 1 : 0;
            k.fz_vec = ((((uintptr_t)fz.raw & 15) == 0 && (a->stride % 8) == 0) ? 1 : 0) |
                       ((((uintptr_t)a->samples & 15) == 0 && (a->stride % 8) == 0) ? 2 : 0);
        }
        hipEvent_t *ev = &c->evpool[3 * (size_t)c->prof_chunks];
        k.sib = nullptr; k.sib_cnt = nullptr; k.sib_out = nullptr; k.tier2 = 0;
        if (siblings) {
            SK_HIP(hipMemsetAsync(c->sib.p, 0, 16, c->stream));
            k.sib_cnt = (int32_t *)c->sib.p; k.sib = (char *)c->sib.p + 16; k.sib_out = (sk_hit *)c->sibout.p;
            k.sib_cap = sib_cap;
        }
        SK_HIP(hipEventRecord(ev[0], c->stream));
        k.clk = (r0 == 0) ? (unsigned long long *)((char *)c->dtwcnt.p + 16) : nullptr;
        hipLaunchKernelGGL(fq, dim3(grid), dim3(256), fz_lds, c->stream, k);
        k.fz_raw = nullptr;                                 // (the later passes only read)
        if (d_early_cnt && r0 + chunk >= a->nreads) SK_HIP(hipEventRecord(c->ev_r[0], c->stream));   // last pass Q done
        SK_HIP(hipGetLastError());
        if (siblings) {
            // the second clusters (struct sibrec): one more short round of both window passes over the sibling list, with
            // the wider look-back (there is no second tier for them).  A handful of reads -- pure latency (pre-roll +
            // window, 0.3-0.9 ms): on the third stream, beside the main window passes, with state and records of its own
            sdtw_kargs ks = k;
            ks.tier2 = 2; ks.wl_list = nullptr; ks.wl_count = k.sib_cnt; ks.soft = nullptr; ks.soft_cnt = nullptr;
            ks.span = tiers ? span2 : span; ks.total_ptr = nullptr; ks.nreads = sib_cap;
            ks.wstate = (unsigned *)c->sibstate.p; ks.wrec = (char *)c->sibstate.p + (size_t)sib_cap * state_bytes;
            const int sgrid = (sib_cap + reads_per_block - 1) / reads_per_block;
            hipStream_t ss = c->stream3 ? c->stream3 : c->stream;
            if (c->stream3) {
                SK_HIP(hipEventRecord(c->ev_s[0], c->stream));
                SK_HIP(hipStreamWaitEvent(c->stream3, c->ev_s[0], 0));
            }
            hipLaunchKernelGGL(fp, dim3(sgrid), dim3(256), 0, ss, ks);
            hipLaunchKernelGGL(fw, dim3(sgrid), dim3(256), 0, ss, ks);
            SK_HIP(hipGetLastError());
            if (c->stream3) SK_HIP(hipEventRecord(c->ev_s[1], c->stream3));
        }
        SK_HIP(hipEventRecord(ev[1], c->stream));
        const int32_t *order = nullptr;
        if (sorted && k.nreads >= sort_min) {
            order_args oa;
            oa.rec = (const wrec *)c->wrecq.p; oa.qflag = (const int32_t *)c->qflag.p;
            oa.nreads = k.nreads; oa.read0 = k.read0; oa.span = span; oa.ck = ck; oa.nck = nck; oa.L = L; oa.wmax = k.wmax;
            oa.hist = (int32_t *)c->order.p; oa.order = oa.hist + ORDER_BINS;
            SK_HIP(hipMemsetAsync(oa.hist, 0, ORDER_BINS * sizeof(int32_t), c->stream));
            const int og = (k.nreads + ORDER_TPB * ORDER_IPT - 1) / (ORDER_TPB * ORDER_IPT);
            hipLaunchKernelGGL(k_order_count, dim3(og), dim3(ORDER_TPB), 0, c->stream, oa);
            hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(ORDER_BINS), 0, c->stream, oa.hist);
            hipLaunchKernelGGL(k_order_scatter, dim3(og), dim3(ORDER_TPB), 0, c->stream, oa);
            SK_HIP(hipGetLastError());
            order = oa.order;
        }
        k.tier2 = 0;
        if (tiers) {
            int32_t *soft = (int32_t *)c->wsoft.p;
            SK_HIP(hipMemsetAsync(soft, 0, sizeof(int32_t), c->stream));
            k.span = span; k.wl_list = order; k.wl_count = nullptr; k.soft = soft + 1; k.soft_cnt = soft;
            hipLaunchKernelGGL(fp, dim3(grid), dim3(256), 0, c->stream, k);
            hipLaunchKernelGGL(fw, dim3(grid), dim3(256), 0, c->stream, k);
            SK_HIP(hipGetLastError());
            k.span = span2; k.wl_list = soft + 1; k.wl_count = soft; k.soft = nullptr; k.soft_cnt = nullptr;
            k.tier2 = 1;
            k.total_ptr = (int32_t *)c->dtwcnt.p + 1;       // reads that needed the second tier (sk_last_dtw_tier2)
        } else {
            k.wl_list = order; k.wl_count = nullptr;
        }
        hipLaunchKernelGGL(fp, dim3(grid), dim3(256), 0, c->stream, k);
        k.total_ptr = nullptr;
        hipLaunchKernelGGL(fw, dim3(grid), dim3(256), 0, c->stream, k);
        SK_HIP(hipGetLastError());
        if (siblings) {                                     // the merge of the second clusters' records, behind both rounds
            if (c->stream3) SK_HIP(hipStreamWaitEvent(c->stream, c->ev_s[1], 0));
            hipLaunchKernelGGL(k_sib_combine, dim3((sib_cap + 255) / 256), dim3(256), 0, c->stream,
                               (const sibrec *)k.sib, (const int32_t *)k.sib_cnt, sib_cap, (const sk_hit *)k.sib_out, k.out,
                               d_retry, d_retry_cnt, (int32_t *)c->dtwcnt.p + 8 + 6);
            SK_HIP(hipGetLastError());
        }
        SK_HIP(hipEventRecord(ev[2], c->stream));
        c->prof_reads[c->prof_chunks < 64 ? c->prof_chunks : 63] = k.nreads;
        c->prof_chunks++;
    }
    return SK_OK;
}
#endif
