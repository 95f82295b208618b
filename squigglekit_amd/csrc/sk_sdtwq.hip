// sk_sdtwq.hip -- subsequence DTW by fixed-point screening + a certified exact window.
//
// The exact FP64 recurrence costs 16 VALU cycles per cell (28 with start tracking), and it has to
// be exact: MotifSeq prints the distance with 17 digits and the path's start/end columns
// (/root/reference/MotifSeq.py:437-449).  But almost all of those cells only serve to show that
// they do NOT hold the minimum.  So:
//
//  pass Q  (k_sdtw_q)   the same wave-systolic sweep in 32-bit fixed point (1 unit = 2^-22):
//          v_min3_u32 + v_sad_u32(clamp) = 2 instructions, 4 cycles per cell.  Its cost matrix Dq
//          differs from the exact one by at most E = N + n + 2 units in any cell (each local cost
//          |q(x)-q(y)| is within one unit of |x-y|, a path has at most N + n cells, FP64 rounding
//          is 10^-10 of a unit).  It stores the last row and, every CK steps, its systolic state.
//  pass W  (k_sdtw_w)   per read: columns whose screening cost is within 2E of the screening
//          minimum are the only ones that can hold the exact minimum [jlo..jhi].  Restart from the
//          checkpoint `span` columns before jlo with every restored cell set to a LOWER BOUND of
//          its exact value ((Dq - E) units) and S = -1, run the exact FP64 recurrence with start
//          tracking up to jhi, take the first exact argmin inside [jlo, jhi].
//          Certificate: every cell computed from lower bounds is itself a lower bound (the
//          recurrence is monotone, also after rounding).  If the tie-ordered back-trace of the
//          winning cell never touches a restored cell (S >= 0), then along it lower bound == exact
//          value, cell by cell from row 0 up, and every rejected predecessor is rejected in the
//          exact matrix too -- so distance, end and start are the reference's, bit for bit.
//          Otherwise (S = -1), or if the range is too wide, a sample left the fixed-point range,
//          or the minimum may have saturated: the read goes to the retry list.
//  retry   the exact single pass (k_sdtw FULL) on the listed reads only.
#include "sk_sdtw_dev.h"
#include <math.h>
#include <string.h>
#include <vector>

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

// ---------------------------------------------------------------------------------------------
// pass Q
// ---------------------------------------------------------------------------------------------
template <int L, int R, int FEED>
__global__ __launch_bounds__(256)
void k_sdtw_q(const sdtw_kargs a)
{
    constexpr int G = 64 / L;
    constexpr int SHR = (L == 16) ? DPP_ROW_SHR1 : DPP_WAVE_SHR1;
    constexpr int CKW = R + 2;

    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int g = lane / L, l = lane % L;
    int slot = wave * G + g;
    const bool live = slot < a.nreads;
    if (!live) slot = a.nreads - 1;
    const int r = a.read0 + slot;

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
        s64 = (const double *)a.samples + a.off[r];
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

    // the sample feed in two halves, so that the load for the next block is in flight during this
    // block's L steps and only converted afterwards
    // (unconditional load from a clamped index: a load under a branch is waited for at the join)
    const int nlast = max(n - 1, 0);
    if (n == 0) { s16 = (const int16_t *)a.xlayq; s64 = (const double *)a.xlayq; }   // any valid address
    auto loadraw = [&](int idx) {
        if constexpr (FEED == SK_FEED_I16) return (int)s16[min(idx, nlast)];
        else                               return s64[min(idx, nlast)];
    };
    auto toq = [&](auto raw_, int idx) -> unsigned {
        if (idx >= n) return QINF;
        const double raw = (double)raw_;
        double v = raw;
        // screening only: (x - c) * (1/s) instead of the reference's division -- the two differ by
        // < 1e-6 of a fixed-point unit, covered by the slack in E (pass W divides, exactly)
        if constexpr (FEED != SK_FEED_F64_RAW) v = (raw - center) * inv_scale;
        const bool ok = fabs(v) < QLIM;              // false for NaN / inf too
        bad |= ok ? 0 : 1;
        return ok ? qimg(v) : QINF;
    };

    const unsigned smask = shortlane ? 0xffffffffu : 0u;
    unsigned *lastq = a.lastq + (int64_t)(r - a.read0) * a.lq_stride;
    // Per read group, in LDS: the sample images of the previous and the current block (lane l
    // needs sample t - l at step t: one ds_read with an immediate offset instead of a DPP shift
    // chain on the vector ALU, which is the bottleneck), and the last-row values lane L-1 produces.
    //   ybuf[0,L) even blocks | ybuf[L,2L) odd blocks | ybuf[2L,3L) copy of [0,L)
    // so that sample t0 + q - l of an even block is ybuf[2L + q - l] and of an odd one ybuf[L + q - l].
    __shared__ unsigned lds_all[4][5 * 64];
    unsigned *ybuf = lds_all[threadIdx.x >> 6] + g * 5 * L;
    unsigned *hbuf = ybuf + 3 * L;
    unsigned *hw = (l == L - 1) ? hbuf : hbuf + L;  // every other lane writes to a dump row
    ybuf[L + l] = QINF;                             // columns before the read
    unsigned F = toq(loadraw(l), l);
    // one step: lane l-1's bottom row comes in by DPP, run the column old -> nw
    auto step = [&](const unsigned (&old)[R], unsigned (&nw)[R], unsigned yq, unsigned *hslot) {
        const unsigned upq = (unsigned)__builtin_amdgcn_update_dpp(0, (int)botq, SHR, 0xF, 0xF, true);
        qcolumn<R>(old, nw, xq, yq, diagq, upq);
        diagq = upq;
        // (a bit select, v_bitop3_b32: 2.6 cycles of issue against v_cndmask's 4.6)
        if constexpr (R >= 2) asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe4" : "=v"(botq) : "v"(nw[R - 2]), "v"(nw[R - 1]), "v"(smask));
        else                  asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe4" : "=v"(botq) : "v"(upq), "v"(nw[0]), "v"(smask));
        *hslot = nw[R - 1];                         // (only lane L-1's lands in hbuf)
    };
    for (int blk = 0; blk < nblk; blk++) {
        const auto rawnext = loadraw((blk + 1) * L + l);
        const int t0 = blk * L;
        if (t0 > 0 && (t0 % a.ck) == 0 && t0 / a.ck <= a.nck && live) {
            unsigned *cp = a.ckq + (((int64_t)(r - a.read0) * a.nck + (t0 / a.ck - 1)) * L + l) * CKW;
#pragma unroll
            for (int k = 0; k < R; k++) cp[k] = Da[k];
            cp[R] = botq; cp[R + 1] = diagq;
        }
        const unsigned *yr;
        if (blk & 1) { ybuf[L + l] = F; yr = ybuf + L - l; }
        else         { ybuf[l] = F; ybuf[2 * L + l] = F; yr = ybuf + 2 * L - l; }
#pragma unroll 1
        for (int qq = 0; qq < L; qq += 16) {
            unsigned yv[16];
#pragma unroll
            for (int q = 0; q < 16; q++) yv[q] = yr[qq + q];
#pragma unroll
            for (int q = 0; q < 16; q += 2) {       // after two steps the ping-pong roles are back
                step(Da, Db, yv[q], hw + qq + q);
                step(Db, Da, yv[q + 1], hw + qq + q + 1);
            }
        }
        F = toq(rawnext, (blk + 1) * L + l);
        const int j = t0 + l - (L - 1);             // lane L-1's column at step t0 + l
        if (j >= 0 && j < n) { const unsigned hv = hbuf[l]; lastq[j] = hv; qmin = min(qmin, hv); }
    }
    // the read's screening minimum for pass W; a sample outside the fixed-point range anywhere in the
    // read disqualifies the screening (reported as an infinite minimum)
#pragma unroll
    for (int d = 1; d < L; d <<= 1) {
        bad |= __shfl_xor(bad, d);
        qmin = min(qmin, (unsigned)__shfl_xor((int)qmin, d));
    }
    if (live && l == 0) a.qflag[r - a.read0] = (int32_t)(bad ? QINF : qmin);
}

// ---------------------------------------------------------------------------------------------
// pass W
// ---------------------------------------------------------------------------------------------
// (four waves per SIMD: the two phases together want ~140 VGPRs; capping at 96 spills into the loops)
template <int L, int R, int FEED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(R <= 13 ? 4 : 3, 8)))
void k_sdtw_w(const sdtw_kargs a)
{
    constexpr int G = 64 / L;
    constexpr int SHR = (L == 16) ? DPP_ROW_SHR1 : DPP_WAVE_SHR1;
    constexpr int ROL = (L == 16) ? DPP_ROW_ROL1 : DPP_WAVE_ROL1;
    constexpr int CKW = R + 2;
    const double INF = __builtin_huge_val();

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
    const int r = a.wl_list ? a.wl_list[slot] : a.read0 + slot;

    int n, flags = 0;
    double center = 0.0, scale = 1.0;
    const int16_t *s16 = nullptr;
    const double  *s64 = nullptr;
    if constexpr (FEED == SK_FEED_I16) {
        const sk_prep pr = a.prep[r];
        n = pr.n; flags = pr.flags; center = pr.center; scale = pr.scale;
        s16 = (const int16_t *)a.samples + (int64_t)r * a.stride;
    } else if constexpr (FEED == SK_FEED_F64_NORM) {
        const sk_prep pr = a.prep[r];
        n = pr.n; flags = pr.flags; center = pr.center; scale = pr.scale;
        s64 = (const double *)a.samples + a.off[r];
    } else {
        n = (int)(a.off[r + 1] - a.off[r]);
        if (n == 0) flags = SK_FLAG_EMPTY;
        s64 = (const double *)a.samples + a.off[r];
    }
    if (!live) n = 0;

    // ---- candidate columns: screening cost within 2E of the screening minimum ----------------
    const unsigned *lastq = a.lastq + (int64_t)(r - a.read0) * a.lq_stride;
    // (rows are 16-byte aligned and padded to a multiple of 4 columns: four columns per load)
    const uint4 *lq4 = (const uint4 *)lastq;
    const int n4 = (n + 3) >> 2;
    const unsigned b = (unsigned)a.qflag[r - a.read0];   // the screening minimum (pass Q), QINF: not usable
    const unsigned thr = (b > QINF - 2u * a.qerr) ? QINF : b + 2u * a.qerr;
    int jlo = 0x7fffffff, jhi = -1;
    for (int q4 = l; q4 < n4; q4 += L) {
        const uint4 v = lq4[q4];
        const int j = q4 * 4;
        if (v.x <= thr) { jlo = min(jlo, j); jhi = max(jhi, j); }
        if (j + 1 < n && v.y <= thr) { jlo = min(jlo, j + 1); jhi = max(jhi, j + 1); }
        if (j + 2 < n && v.z <= thr) { jlo = min(jlo, j + 2); jhi = max(jhi, j + 2); }
        if (j + 3 < n && v.w <= thr) { jlo = min(jlo, j + 3); jhi = max(jhi, j + 3); }
    }
#pragma unroll
    for (int d = 1; d < L; d <<= 1) { jlo = min(jlo, __shfl_xor(jlo, d)); jhi = max(jhi, __shfl_xor(jhi, d)); }
    const bool screened = (n > 0) && (b < QSAFE) && (jhi >= jlo) &&
                          (jhi - jlo <= a.wmax);

    int tbase = 0, tlast = -1, c0 = 0, npre = 0;
    if (screened) {
        const int tx = max(0, jlo - a.span);        // where the exact recurrence has to start
        c0 = tx / a.ck;
        if (c0 > a.nck) c0 = a.nck;
        tbase = c0 * a.ck;
        tlast = jhi + L - 1;
        if (c0 > 0) npre = (tx - tbase) / L;        // whole blocks between the checkpoint and tx
    }
    const bool shortlane = l < a.P;

    // ---- pre-roll: from the checkpoint to the block that holds tx in fixed point -------------------
    // The checkpoints are ck steps apart; running the remaining (on average ck/2) steps with the
    // exact FP64 recurrence would cost four times what the screening recurrence does, so the
    // systolic state is first advanced in fixed point.  Groups of one wave need different numbers of
    // blocks: a group loads its checkpoint in the iteration in which its turn starts (what it
    // computed before on never-initialised state is overwritten), so all groups finish together.
    unsigned Qs[R];                                 // screening state: R cells, lane l-1's row, diagonal
    unsigned botq = QINF, diagq = QINF;
#pragma unroll
    for (int k = 0; k < R; k++) Qs[k] = QINF;
    const unsigned *cp = a.ckq + (((int64_t)(r - a.read0) * a.nck + (c0 > 0 ? c0 - 1 : 0)) * L + l) * CKW;
    auto load_ckpt = [&]() {
#pragma unroll
        for (int k = 0; k < R; k++) Qs[k] = cp[k];
        botq = cp[R]; diagq = cp[R + 1];
    };
    int maxpre = npre;
#pragma unroll
    for (int d = L; d < 64; d <<= 1) maxpre = max(maxpre, __shfl_xor(maxpre, d));
    maxpre = __builtin_amdgcn_readfirstlane(maxpre);
    if (maxpre > 0) {
        const double inv_scale = 1.0 / scale;
        const int nlast = max(n - 1, 0);
        auto toq = [&](int idx) -> unsigned {       // the screening pass's image of sample idx
            if (idx < 0 || idx >= n) return QINF;
            double v;
            if constexpr (FEED == SK_FEED_I16)           v = ((double)s16[min(idx, nlast)] - center) * inv_scale;
            else if constexpr (FEED == SK_FEED_F64_NORM) v = (s64[min(idx, nlast)] - center) * inv_scale;
            else                                         v = s64[min(idx, nlast)];
            return (fabs(v) < QLIM) ? qimg(v) : QINF;
        };
        unsigned xq[R];
#pragma unroll
        for (int k = 0; k < R; k++) xq[k] = a.xlayq[l * R + k];
        unsigned Qn[R];
#pragma unroll
        for (int k = 0; k < R; k++) Qn[k] = QINF;
        unsigned yq = QINF, Fq = QINF;
        const int startblk = maxpre - npre;         // my group's first block
        auto qstep = [&](const unsigned (&old)[R], unsigned (&nw)[R]) {
            yq = (unsigned)dpp_i32<SHR>((int)Fq, (int)yq);
            Fq = (unsigned)dpp_i32<ROL>((int)Fq, (int)Fq);
            const unsigned upq = (unsigned)__builtin_amdgcn_update_dpp(0, (int)botq, SHR, 0xF, 0xF, true);
            qcolumn<R>(old, nw, xq, yq, diagq, upq);
            diagq = upq;
            if constexpr (R >= 2) botq = shortlane ? nw[R - 2] : nw[R - 1];
            else                  botq = shortlane ? upq : nw[0];
        };
        for (int pb = 0; pb < maxpre; pb++) {
            const int t0 = tbase + (pb - startblk) * L;
            if (pb == startblk && npre > 0) {
                load_ckpt();
                yq = toq(t0 - 1 - l);               // the sample this lane held after step t0 - 1
                Fq = toq(t0 + l);
            }
            const unsigned Fnext = toq(t0 + L + l);
#pragma unroll 1
            for (int q = 0; q < L; q += 2) {        // L is even: after two steps the roles are back
                qstep(Qs, Qn);
                qstep(Qn, Qs);
            }
            Fq = Fnext;
        }
    }
    if (c0 > 0 && npre == 0) load_ckpt();
    tbase += npre * L;
    asm volatile("" ::: "memory");                  // keep the exact phase's loads (and registers) below

    int nsteps = tlast - tbase + 1;
    if (nsteps < 0) nsteps = 0;
#pragma unroll
    for (int d = L; d < 64; d <<= 1) nsteps = max(nsteps, __shfl_xor(nsteps, d));
    nsteps = __builtin_amdgcn_readfirstlane(nsteps);
    const int nblk = (nsteps + L - 1) / L;

    double x[R];
#pragma unroll
    for (int k = 0; k < R; k++) x[k] = a.xlay[l * R + k];

    auto fetch = [&](int idx) -> double {
        if (idx < 0 || idx >= n) return INF;
        if constexpr (FEED == SK_FEED_I16)           return ((double)s16[idx] - center) / scale;
        else if constexpr (FEED == SK_FEED_F64_NORM) return (s64[idx] - center) / scale;
        else                                         return s64[idx];
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
    double y = INF;
    if (c0 > 0) {
#pragma unroll
        for (int k = 0; k < R; k++) D[k] = lb(Qs[k]);
        botD = lb(botq);
        if (l > 0) diagD = lb(diagq);               // lane 0's diag is the virtual row: exactly 0
        if (R == 1 && l == 0 && shortlane) { botD = 0.0; botS = tbase; }   // forwards the virtual row
        y = fetch(tbase - 1 - l);                   // the sample this lane held after step tbase-1
    }
    double best = INF;  int bestS = -1, bestJ = -1;
    // the running argmin is only wanted where lane L-1 sits on a candidate column -- the last few steps of the window:
    // first step (relative to tbase) at which some read group of the wave needs it
    int s0 = screened ? max(jlo - tbase + L - 1, 0) : 0x7fffffff;
#pragma unroll
    for (int d = L; d < 64; d <<= 1) s0 = min(s0, __shfl_xor(s0, d));
    s0 = __builtin_amdgcn_readfirstlane(s0);

    double F = fetch(tbase + l);
    for (int blk = 0; blk < nblk; blk++) {
        const double Fnext = fetch(tbase + (blk + 1) * L + l);
#pragma unroll 2
        for (int q = 0; q < L; q++) {
            const int t = tbase + blk * L + q;
            y = dpp_f64<SHR>(F, y);
            F = dpp_f64<ROL>(F, F);
            const double upD = dpp_f64<SHR>(0.0, botD);
            const int    upS = dpp_i32<SHR>(t + 1, botS);
            double dgD = diagD;  int dgS = diagS;
            double uD = upD;     int uS = upS;
#pragma unroll
            for (int k = 0; k < R; k++) {
                const double lfD = D[k];  const int lfS = S[k];
                const double c = fabs(x[k] - y);
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
            if (blk * L + q >= s0) {                       // (wave-uniform)
                const int j = t - l;
                if (j >= jlo && j <= jhi && D[R - 1] < best) { best = D[R - 1]; bestS = S[R - 1]; bestJ = j; }
            }
        }
        F = Fnext;
    }

    if (live && l == L - 1) {
        sk_hit h;
        h.n = n; h.flags = flags;
        if (n <= 0) {
            h.dist = __builtin_nan(""); h.start = -1; h.end = -1;
            a.out[r] = h;
        } else if (screened && bestS >= 0) {
            h.dist = best; h.start = bestS; h.end = bestJ;
            a.out[r] = h;
        } else {
            h.dist = __builtin_nan(""); h.start = -1; h.end = -1;    // overwritten by a later pass
            a.out[r] = h;
            if (screened && a.soft) a.soft[atomicAdd(a.soft_cnt, 1)] = r;   // path wider than this look-back: next tier
            else                    a.retry[atomicAdd(a.retry_cnt, 1)] = r; // the exact single pass
        }
    }
}

typedef void (*sdtw_fn)(const sdtw_kargs);

template <int L, int FEED, int WHICH>
sdtw_fn pick_r(int R)
{
    switch (R) {
#define SK_CASE(RR) case RR: return WHICH == 0 ? (sdtw_fn)k_sdtw_q<L, RR, FEED> : (sdtw_fn)k_sdtw_w<L, RR, FEED>;
        SK_CASE(1) SK_CASE(2) SK_CASE(3) SK_CASE(4) SK_CASE(5) SK_CASE(6) SK_CASE(7) SK_CASE(8)
        SK_CASE(9) SK_CASE(10) SK_CASE(11) SK_CASE(12) SK_CASE(13) SK_CASE(14) SK_CASE(15) SK_CASE(16)
#undef SK_CASE
    }
    return nullptr;
}

template <int WHICH>
sdtw_fn pick(int feed, int L, int R)
{
    if (L == 16) {
        if (feed == SK_FEED_I16) return pick_r<16, SK_FEED_I16, WHICH>(R);
        if (feed == SK_FEED_F64_NORM) return pick_r<16, SK_FEED_F64_NORM, WHICH>(R);
        return pick_r<16, SK_FEED_F64_RAW, WHICH>(R);
    }
    if (feed == SK_FEED_I16) return pick_r<64, SK_FEED_I16, WHICH>(R);
    if (feed == SK_FEED_F64_NORM) return pick_r<64, SK_FEED_F64_NORM, WHICH>(R);
    return pick_r<64, SK_FEED_F64_RAW, WHICH>(R);
}

} // namespace

// Screening + certified window over all reads; fills out[] and the retry list (device).
// The caller (sk_launch_sdtw) reads the retry count and runs the exact pass on those reads.
// span / span2: look-back of the window pass's first tier (every read) and of its second tier (the reads whose
// optimal path turned out wider than `span`); span2 <= span: one tier only.
int sk_launch_sdtw_screen(sk_ctx *c, const sk_sdtw_args *a, int L, int R, int P, int ck, int span, int span2,
                          int32_t *d_retry_cnt, int32_t *d_retry)
{
    const int N = a->nmotif;
    // quantised motif in the same per-lane layout as the exact one; resident like the exact layout (the caller
    // invalidates it when the motif or its layout changes), so a call makes no host-side synchronisation
    int rc;
    if (!c->motifq_valid) {
        SK_HIP(hipStreamSynchronize(c->stream));        // an earlier launch may still read the old one
        std::vector<unsigned> &layq = c->motifq_host;
        layq.assign((size_t)L * R, 0x80000000u);
        int row = 0;
        for (int l = 0; l < L; l++) {
            const int cnt = (l < P) ? R - 1 : R;
            for (int k = 0; k < cnt; k++)
                layq[(size_t)l * R + k] = (unsigned)((int)rint(a->motif[row++] * QSCALE)) + 0x80000000u;
        }
        if ((rc = sk_reserve(c, &c->motifq, layq.size() * sizeof(unsigned)))) return rc;
        SK_HIP(hipMemcpyAsync(c->motifq.p, layq.data(), layq.size() * sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
        c->motifq_valid = true;
    }

    const int64_t maxlen = a->max_len;
    const int nck = (int)((maxlen + L - 1) / ck);
    const size_t lq_stride = (size_t)((maxlen + 3) & ~(int64_t)3);
    const size_t per_read = (size_t)(nck > 0 ? nck : 1) * L * (R + 2) * sizeof(unsigned) +
                            lq_stride * sizeof(unsigned) + sizeof(int32_t);
    int64_t chunk;
    while (true) {                                      // (a device short of memory: halve the budget and try again)
        chunk = sk_dtw_chunk_reads(per_read, a->nreads);
        rc = sk_reserve(c, &c->ckpt, (size_t)chunk * (size_t)(nck > 0 ? nck : 1) * L * (R + 2) * sizeof(unsigned));
        if (!rc) rc = sk_reserve(c, &c->lastq, (size_t)chunk * lq_stride * sizeof(unsigned));
        if (rc != SK_ERR_NOMEM || chunk <= 1024) break;
        sk_dtw_scratch_shrink(0);
    }
    if (rc) return rc;
    if ((rc = sk_reserve(c, &c->qflag, (size_t)chunk * sizeof(int32_t)))) return rc;
    const bool tiers = span2 > span;
    if (tiers && (rc = sk_reserve(c, &c->wsoft, ((size_t)chunk + 1) * sizeof(int32_t)))) return rc;

    sdtw_fn fq = pick<0>(a->feed, L, R), fw = pick<1>(a->feed, L, R);
    if (!fq || !fw) return sk_fail(SK_ERR_UNSUPPORTED, "no screening kernel for L=%d R=%d", L, R);

    sdtw_kargs k;
    memset(&k, 0, sizeof k);
    k.samples = a->samples; k.stride = a->stride; k.off = a->off; k.prep = a->prep;
    k.xlay = (const double *)c->motif.p; k.xlayq = (const unsigned *)c->motifq.p; k.P = P; k.out = a->out;
    k.nck = nck; k.ck = ck; k.span = span; k.retry = d_retry; k.retry_cnt = d_retry_cnt;
    k.ckq = (unsigned *)c->ckpt.p; k.lastq = (unsigned *)c->lastq.p; k.lq_stride = (int64_t)lq_stride;
    k.qflag = (int32_t *)c->qflag.p;
    k.qerr = (unsigned)(N + maxlen + 2);
    k.wmax = 4 * ck;

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
        hipEvent_t *ev = &c->evpool[3 * (size_t)c->prof_chunks];
        SK_HIP(hipEventRecord(ev[0], c->stream));
        hipLaunchKernelGGL(fq, dim3(grid), dim3(256), 0, c->stream, k);
        SK_HIP(hipGetLastError());
        SK_HIP(hipEventRecord(ev[1], c->stream));
        if (tiers) {
            int32_t *soft = (int32_t *)c->wsoft.p;
            SK_HIP(hipMemsetAsync(soft, 0, sizeof(int32_t), c->stream));
            k.span = span; k.wl_list = nullptr; k.wl_count = nullptr; k.soft = soft + 1; k.soft_cnt = soft;
            hipLaunchKernelGGL(fw, dim3(grid), dim3(256), 0, c->stream, k);
            SK_HIP(hipGetLastError());
            k.span = span2; k.wl_list = soft + 1; k.wl_count = soft; k.soft = nullptr; k.soft_cnt = nullptr;
            k.total_ptr = (int32_t *)c->dtwcnt.p + 1;       // reads that needed the second tier (sk_last_dtw_tier2)
        }
        hipLaunchKernelGGL(fw, dim3(grid), dim3(256), 0, c->stream, k);
        SK_HIP(hipGetLastError());
        k.total_ptr = nullptr;
        SK_HIP(hipEventRecord(ev[2], c->stream));
        c->prof_reads[c->prof_chunks < 64 ? c->prof_chunks : 63] = k.nreads;
        c->prof_chunks++;
    }
    return SK_OK;
}
