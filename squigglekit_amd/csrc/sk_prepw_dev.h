// sk_prepw_dev.h -- one read through scale_outliers + medmad on ONE wavefront (device code shared by
// k_prepw_medmad, sk_prepw.hip, and the fused prologue of the screening pass, sk_sdtwq.hip).
// Reference: scale_outliers MotifSeq.py:317-324, medmad MotifSeq.py:192-200.
#pragma once
#include "sk_common.h"

namespace {

typedef short i16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned clamp_pk_i16(unsigned q, unsigned lo2, unsigned hi2)
{
    const i16x2 x = __builtin_bit_cast(i16x2, q);
    const i16x2 c = __builtin_elementwise_min(__builtin_elementwise_max(x, __builtin_bit_cast(i16x2, lo2)),
                                              __builtin_bit_cast(i16x2, hi2));
    return __builtin_bit_cast(unsigned, c);
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

__device__ __forceinline__ int bcast_from(int v, int src_lane)        // src_lane wave-uniform
{
    return __builtin_amdgcn_readlane(v, src_lane);
}

// eight consecutive samples (four packed pairs) to dst; al = (element offset of dst) mod 8 when the
// row base is 16-byte aligned, odd when nothing is known
__device__ __forceinline__ void put8(int16_t *dst, const unsigned (&q)[4], int al)
{
    if (al == 0) {
        *(uint4 *)dst = make_uint4(q[0], q[1], q[2], q[3]);
    } else if ((al & 1) == 0) {
        unsigned *d = (unsigned *)dst;
        d[0] = q[0]; d[1] = q[1]; d[2] = q[2]; d[3] = q[3];
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            dst[2 * k] = (int16_t)(q[k] & 0xffffu);
            dst[2 * k + 1] = (int16_t)(q[k] >> 16);
        }
    }
}

__device__ __forceinline__ int sample_of(const unsigned (&q)[4], int k)
{
    return (k & 1) ? (int)q[k >> 1] >> 16 : (int)(short)(q[k >> 1] & 0xffffu);
}

// Rank select on a histogram held in registers: lane l owns cnt[i] = count of bin l*4*NQ + i.
// Every lane gets the bins holding ranks k1 <= k2.
template <int NQ>
__device__ __forceinline__ void rank2(const unsigned (&cnt)[4 * NQ], int lane, int k1, int k2, int &b1, int &b2,
                                      int *pre_out = nullptr)
{
    const int b0 = lane * 4 * NQ;
    int local = 0;
#pragma unroll
    for (int i = 0; i < 4 * NQ; i++) local += (int)cnt[i];
    const int inc = wave_incl_scan(local);
    const int pre = inc - local;
    if (pre_out) *pre_out = pre;                            // samples in the bins below this lane's
    int i1 = b0, i2 = b0, acc = pre;
#pragma unroll
    for (int i = 0; i < 4 * NQ; i++) {
        acc += (int)cnt[i];
        i1 += (acc <= k1) ? 1 : 0;
        i2 += (acc <= k2) ? 1 : 0;
    }
    const unsigned long long own1 = __ballot(local > 0 && k1 >= pre && k1 < pre + local);
    const unsigned long long own2 = __ballot(local > 0 && k2 >= pre && k2 < pre + local);
    b1 = bcast_from(i1, own1 ? (int)__builtin_ctzll(own1) : 0);
    b2 = bcast_from(i2, own2 ? (int)__builtin_ctzll(own2) : 0);
}

// Per-wave constants of the filter + histogram code.
struct prepw_env {
    unsigned *hist;        // this wave's value histogram in LDS: nb4 words, zero between reads
    int lo, hi, nbins, nb4;
    int hb0;               // first bin this lane owns
    bool in_vec, out_vec;  // 16-byte loads of the input rows / stores of the compacted rows are possible
    unsigned lo2, hi2;     // the keep range as packed int16 pairs
};

template <int NQ>
__device__ __forceinline__ prepw_env prepw_setup(unsigned *hist, int lane, int lo, int hi, int vec_ok)
{
    prepw_env e;
    e.hist = hist; e.lo = lo; e.hi = hi;
    e.nbins = hi - lo - 1;                                   // >= 1 (host)
    e.nb4 = (e.nbins + 3) & ~3;
    e.hb0 = lane * 4 * NQ;
    e.in_vec = (vec_ok & 1) != 0; e.out_vec = (vec_ok & 2) != 0;
    const int lo1 = max(lo + 1, -32768), hi1 = min(hi - 1, 32767);
    e.lo2 = (unsigned)(lo1 & 0xffff) * 0x10001u; e.hi2 = (unsigned)(hi1 & 0xffff) * 0x10001u;
    // the histogram starts zeroed; every lane clears the bins it owns after use
#pragma unroll
    for (int j = 0; j < NQ; j++)
        if (e.hb0 + 4 * j < e.nb4) *(uint4 *)(hist + e.hb0 + 4 * j) = make_uint4(0u, 0u, 0u, 0u);
    return e;
}

// Read r: filter, compacted samples -> comp row, statistics -> prep[r] (lane 0 stores) and returned in every lane.
// NQ: 16-byte chunks of histogram per lane (bins <= 256 NQ).
template <int NQ>
__device__ __forceinline__ sk_prep prepw_read(const prepw_env &E, const int16_t *__restrict__ sig, int64_t stride,
                                              const int32_t *__restrict__ len, int r, int lane,
                                              int16_t *__restrict__ comp, sk_prep *__restrict__ prep)
{
    unsigned *const hist = E.hist;
    unsigned *const hist_v = hist - (E.lo + 1);            // hist_v[x] counts value x
    const int lo = E.lo, hi = E.hi, nbins = E.nbins, nb4 = E.nb4, hb0 = E.hb0;
    const bool in_vec = E.in_vec, out_vec = E.out_vec;
    const unsigned lo2 = E.lo2, hi2 = E.hi2;
    auto load8 = [&](const int16_t *row, int M, int i0, unsigned (&q)[4]) {
        if (in_vec && i0 + 8 <= M) {
            const uint4 t = *(const uint4 *)(row + i0);
            q[0] = t.x; q[1] = t.y; q[2] = t.z; q[3] = t.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned a = (i0 + 2 * k < M) ? (unsigned short)row[i0 + 2 * k] : 0u;
                const unsigned b = (i0 + 2 * k + 1 < M) ? (unsigned short)row[i0 + 2 * k + 1] : 0u;
                q[k] = a | (b << 16);
            }
        }
    };
    {
        const int M = min(max(len[r], 0), (int)min(stride, (int64_t)0x7fffff00));   // never past the row
        const int16_t *row = sig + (int64_t)r * stride;
        int16_t *crow = comp + (int64_t)r * stride;

        // ---- pass 1: filter, compact (order preserving), histogram ------------------------------
        int run = 0;
        {
            unsigned v[4], vn[4];
            load8(row, M, lane * 8, v);
            for (int base = 0; base < M; base += 512) {
                const int i0 = base + lane * 8;
                if (base + 512 < M) load8(row, M, i0 + 512, vn);           // next tile in flight
                // Outliers are rare: when all 512 samples survive (clamping the packed pairs to the
                // keep range changes nothing) there is nothing to scan or to test.
                unsigned changed = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) changed |= clamp_pk_i16(v[k], lo2, hi2) ^ v[k];
                if (__all(i0 + 8 <= M && changed == 0u)) {
                    put8(crow + run + lane * 8, v, out_vec ? (run & 7) : 1);
#pragma unroll
                    for (int k = 0; k < 8; k++) atomicAdd(&hist_v[sample_of(v, k)], 1u);
                    run += 512;
                } else {
                    unsigned keep = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int x = sample_of(v, k);
                        if (i0 + k < M && x > lo && x < hi) keep |= 1u << k;
                    }
                    const int cnt = __popc(keep);
                    const int inc = wave_incl_scan(cnt);
                    int o = run + inc - cnt;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (keep & (1u << k)) {
                            const int x = sample_of(v, k);
                            crow[o++] = (int16_t)x;
                            atomicAdd(&hist_v[x], 1u);
                        }
                    }
                    run += bcast_from(inc, 63);
                }
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = vn[k];
            }
        }
        const int n = run;

        sk_prep pr;
        pr.n = n; pr.flags = 0; pr.center = 0.0; pr.scale = 1.0; pr.top = 0.0; pr.bot = 0.0;
        if (n == 0) {                                       // (nothing was counted: histogram still zero)
            pr.flags = SK_FLAG_EMPTY;
            const double qnan = __builtin_nan("");
            pr.center = qnan; pr.scale = qnan; pr.top = qnan; pr.bot = qnan;
            if (lane == 0) prep[r] = pr;
            return pr;
        }

        // ---- median: ranks (n-1)/2 and n/2 of the value histogram, from registers -------------------
        unsigned cnt[4 * NQ];
#pragma unroll
        for (int j = 0; j < NQ; j++) {
            uint4 q = make_uint4(0u, 0u, 0u, 0u);
            if (hb0 + 4 * j < nb4) q = *(const uint4 *)(hist + hb0 + 4 * j);
            cnt[4 * j] = q.x; cnt[4 * j + 1] = q.y; cnt[4 * j + 2] = q.z; cnt[4 * j + 3] = q.w;
        }
        int b1, b2, pre_med;
        rank2<NQ>(cnt, lane, (n - 1) / 2, n / 2, b1, b2, &pre_med);
        const int med2 = (b1 + lo + 1) + (b2 + lo + 1);                    // 2 * median, exact

        // ---- MAD = median of |x - med| ---------------------------------------------------------------------------
        // |2x - med2| takes the values 2t (med2 even) or 2t + 1 (odd), and the number of samples within deviation t is
        //     C(t) = P[cr + t] - P[cl - t - 1]           P = inclusive prefix counts of the value histogram
        // (cl / cr: the bins just below / above the median, equal when it is an integer).  The prefix counts replace
        // the histogram in LDS (each lane has its bins' counts and the scan of the median select in registers); the
        // smallest t with C(t) > k is then found for both middle ranks in two 64-way steps -- lane l probes the end
        // of block l, then the lanes of each half probe one block's members -- instead of folding the histogram
        // around the median bin by bin.
        const int odd = med2 & 1;
        const int cl = ((med2 - odd) >> 1) - (lo + 1);                     // bin just below / at the median
        const int cr = cl + odd;
        {
            int acc = pre_med;
#pragma unroll
            for (int j = 0; j < NQ; j++) {
                uint4 q;
                acc += (int)cnt[4 * j];     q.x = (unsigned)acc;
                acc += (int)cnt[4 * j + 1]; q.y = (unsigned)acc;
                acc += (int)cnt[4 * j + 2]; q.z = (unsigned)acc;
                acc += (int)cnt[4 * j + 3]; q.w = (unsigned)acc;
                if (hb0 + 4 * j < nb4) *(uint4 *)(hist + hb0 + 4 * j) = q;
            }
        }
        auto within = [&](int t) -> int {                                  // C(t), t >= 0
            const int hi_i = min(cr + t, nbins - 1), lo_i = cl - t - 1;
            const int a = (int)hist[hi_i];
            const int b = (lo_i >= 0) ? (int)hist[max(lo_i, 0)] : 0;
            return a - b;
        };
        const int k1 = (n - 1) / 2, k2 = n / 2;
        const int step = (nbins + 63) >> 6;                                // <= 32 (nbins <= 2048)
        const int cend = within(lane * step + step - 1);                   // non-decreasing in the lane index
        const unsigned long long ge1 = __ballot(cend > k1), ge2 = __ballot(cend > k2);
        const int B1 = ge1 ? (int)__builtin_ctzll(ge1) : 63, B2 = ge2 ? (int)__builtin_ctzll(ge2) : 63;
        const int half = lane >> 5, li = lane & 31;
        const int tprobe = (half ? B2 : B1) * step + li;
        const int cin = within(tprobe);
        const unsigned long long hit = __ballot(li < step && cin > (half ? k2 : k1));
        const unsigned h1 = (unsigned)hit, h2 = (unsigned)(hit >> 32);
        const int t1 = B1 * step + (h1 ? (int)__builtin_ctz(h1) : step - 1);
        const int t2 = B2 * step + (h2 ? (int)__builtin_ctz(h2) : step - 1);
        const double mad = (double)((2 * t1 + odd) + (2 * t2 + odd)) * 0.25;   // (d1/2 + d2/2) / 2, exact
        pr.center = (double)med2 * 0.5;
        pr.scale = mad * 1.4826;                                           // MotifSeq.py:196
        if (mad == 0.0) pr.flags |= SK_FLAG_DEGENERATE;
        if (lane == 0) prep[r] = pr;
        // clear the bins I own (all my reads of the histogram are done: LDS ops of a wave are in order)
#pragma unroll
        for (int j = 0; j < NQ; j++)
            if (hb0 + 4 * j < nb4) *(uint4 *)(hist + hb0 + 4 * j) = make_uint4(0u, 0u, 0u, 0u);
        return pr;
    }
}


// ------------------------------------------------------------------------------------------------------------------
// zscale on one wavefront (round 5; the fused prologue of the screening pass for `-l zscale`, reads of up to 4 096
// samples).  sklearn.preprocessing.scale as MotifSeq.py:186-191 calls it: (x - np.mean(x)) / np.std(x).  The mean of
// integer samples is an exact integer sum and one division; np.std's sum of (x - mean)^2 has to be added up in NUMPY'S
// ORDER, because the value itself -- not a comparison against it -- goes into every normalised sample: the pairwise
// tree of np.add.reduce (a node longer than 128 splits at (len / 2) rounded down to a multiple of 8; a leaf is eight
// strided accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a serial tail), which the workgroup kernel
// (sk_prep.hip pairwise_chunk) builds as a table.  Here: every leaf is at least 64 long and starts at a multiple of 8,
// so lane l probes position 64 l, walks down from the root to the leaf that holds it (<= 6 steps of integer
// arithmetic) and sums that leaf if it is the first probe inside it -- one leaf per lane, from the compacted samples
// the wave keeps in LDS (16-byte reads); the leaf sums meet in a 128-entry heap in LDS, folded bottom-up level by level.
// ------------------------------------------------------------------------------------------------------------------
struct zs_env {
    int16_t *lcomp;        // this wave's compacted samples in LDS (16-byte aligned, >= stride int16)
    double  *nodes;        // 128 doubles of LDS: the pairwise tree's partial sums, heap order
    int lo, hi;
    bool in_vec, out_vec;
    unsigned lo2, hi2;
};

__device__ __forceinline__ zs_env zs_setup(int16_t *lcomp, double *nodes, int lo, int hi, int vec_ok)
{
    zs_env e;
    e.lcomp = lcomp; e.nodes = nodes; e.lo = lo; e.hi = hi;
    e.in_vec = (vec_ok & 1) != 0; e.out_vec = (vec_ok & 2) != 0;
    const int lo1 = max(lo + 1, -32768), hi1 = min(hi - 1, 32767);
    e.lo2 = (unsigned)(lo1 & 0xffff) * 0x10001u; e.hi2 = (unsigned)(hi1 & 0xffff) * 0x10001u;
    return e;
}

// sum over i < m of sq(lcomp[i]) in np.add.reduce's order, m <= 8192 (one reduction chunk); sq(v) >= 0
template <typename Sq>
__device__ __forceinline__ double wave_pairwise_sq(int m, const int16_t *lcomp, double *nodes, int lane, Sq sq)
{
    if (m < 8) {                                            // numpy: a plain serial loop
        double res = 0.0;
        for (int i = 0; i < m; i++) res += sq((int)lcomp[i]);
        return res;
    }
    const int p = lane * 64;
    int s = 0, len = m, id = 1;
#pragma unroll 1
    for (int it = 0; it < 7; it++) {
        if (len > 128) {
            int n2 = len / 2;
            n2 -= n2 % 8;
            if (p < s + n2) { len = n2; id = 2 * id; }
            else            { s += n2; len -= n2; id = 2 * id + 1; }
        }
    }
    const bool active = p < m && (lane == 0 || s > p - 64);     // the first probe inside its leaf
    nodes[lane] = -1.0; nodes[lane + 64] = -1.0;                // "no such node" (the sums are >= 0)
    __builtin_amdgcn_wave_barrier();
    if (active) {
        const int16_t *q = lcomp + s;                            // (s is a multiple of 8: 16-byte aligned)
        double r[8];
        {
            const uint4 t = *(const uint4 *)q;
            const unsigned w4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = sq((j & 1) ? (int)w4[j >> 1] >> 16 : (int)(short)(w4[j >> 1] & 0xffffu));
        }
        const int full = len - (len % 8);
        for (int i = 8; i < full; i += 8) {
            const uint4 t = *(const uint4 *)(q + i);
            const unsigned w4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] += sq((j & 1) ? (int)w4[j >> 1] >> 16 : (int)(short)(w4[j >> 1] & 0xffffu));
        }
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (int i = full; i < len; i++) res += sq((int)q[i]);
        nodes[id] = res;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int d = 6; d >= 1; d--) {                              // parents at depth d - 1 from their children at depth d
        const int pid = (1 << (d - 1)) + lane;
        if (lane < (1 << (d - 1))) {
            const double a = nodes[2 * pid], b = nodes[2 * pid + 1];
            if (a >= 0.0 && b >= 0.0) nodes[pid] = a + b;
        }
        __builtin_amdgcn_wave_barrier();
    }
    return nodes[1];
}

// Read r: filter, compacted samples -> comp row (and LDS), mean / std -> prep[r] (lane 0 stores), returned in every lane.
__device__ __forceinline__ sk_prep zs_read(const zs_env &E, const int16_t *__restrict__ sig, int64_t stride,
                                           const int32_t *__restrict__ len, int r, int lane,
                                           int16_t *__restrict__ comp, sk_prep *__restrict__ prep)
{
    const int lo = E.lo, hi = E.hi;
    const unsigned lo2 = E.lo2, hi2 = E.hi2;
    const int M = min(max(len[r], 0), (int)min(stride, (int64_t)0x7fffff00));
    const int16_t *row = sig + (int64_t)r * stride;
    int16_t *crow = comp + (int64_t)r * stride;
    int16_t *lcomp = E.lcomp;
    auto load8 = [&](int i0, unsigned (&q)[4]) {
        if (E.in_vec && i0 + 8 <= M) {
            const uint4 t = *(const uint4 *)(row + i0);
            q[0] = t.x; q[1] = t.y; q[2] = t.z; q[3] = t.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned a = (i0 + 2 * k < M) ? (unsigned short)row[i0 + 2 * k] : 0u;
                const unsigned b = (i0 + 2 * k + 1 < M) ? (unsigned short)row[i0 + 2 * k + 1] : 0u;
                q[k] = a | (b << 16);
            }
        }
    };
    int run = 0, isum = 0;                                  // (|sum| <= 4 096 * 32 768 < 2^31)
    unsigned v[4], vn[4];
    load8(lane * 8, v);
    for (int base = 0; base < M; base += 512) {
        const int i0 = base + lane * 8;
        if (base + 512 < M) load8(i0 + 512, vn);            // next tile in flight
        unsigned changed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) changed |= clamp_pk_i16(v[k], lo2, hi2) ^ v[k];
        if (__all(i0 + 8 <= M && changed == 0u)) {
            put8(crow + run + lane * 8, v, E.out_vec ? (run & 7) : 1);
            put8(lcomp + run + lane * 8, v, run & 7);
#pragma unroll
            for (int k = 0; k < 8; k++) isum += sample_of(v, k);
            run += 512;
        } else {
            unsigned keep = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int x = sample_of(v, k);
                if (i0 + k < M && x > lo && x < hi) keep |= 1u << k;
            }
            const int cnt = __popc(keep);
            const int inc = wave_incl_scan(cnt);
            int o = run + inc - cnt;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (keep & (1u << k)) {
                    const int x = sample_of(v, k);
                    crow[o] = (int16_t)x;
                    lcomp[o] = (int16_t)x;
                    isum += x;
                    o++;
                }
            }
            run += bcast_from(inc, 63);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = vn[k];
    }
    const int n = run;
    sk_prep pr;
    pr.n = n; pr.flags = 0; pr.center = 0.0; pr.scale = 1.0; pr.top = 0.0; pr.bot = 0.0;
    if (n == 0) {
        pr.flags = SK_FLAG_EMPTY;
        const double qnan = __builtin_nan("");
        pr.center = qnan; pr.scale = qnan; pr.top = qnan; pr.bot = qnan;
        if (lane == 0) prep[r] = pr;
        return pr;
    }
    const int S = bcast_from(wave_incl_scan(isum), 63);
    const double mean = (double)S / (double)n;
    __builtin_amdgcn_wave_barrier();                        // (the LDS copy is complete: LDS operations of a wave are in order)
    const double ssq = wave_pairwise_sq(n, lcomp, E.nodes, lane, [&](int x) { const double d = (double)x - mean; return d * d; });
    const double sd = sqrt(ssq / (double)n);
    pr.center = mean;
    pr.scale = (sd == 0.0) ? 1.0 : sd;                      // sklearn _handle_zeros_in_scale
    if (lane == 0) prep[r] = pr;
    return pr;
}

} // namespace
