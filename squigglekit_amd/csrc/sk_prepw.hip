// sk_prepw.hip -- filter + medmad statistics with ONE WAVEFRONT PER READ (int16 reads, MotifSeq path).
//
// Same results as the medmad instantiation of k_prep_i16 (sk_prep.hip), different shape.  The
// workgroup-per-read kernel is bound by the latency of its barrier-separated phases with 8 reads in
// flight per CU.  medmad needs no second look at the samples, so here a read belongs to one wavefront
// from the first load to the last store: no workgroup barrier, 32 reads in flight per CU, and the only
// LDS a wave owns is its value histogram (<= 8 KB).
//
//   pass 1   stream the read (16-byte loads, 8 samples per lane, packed): filter, write the survivors
//            in order for the DTW kernels, histogram (LDS atomics)
//   median   rank select on the histogram from registers (lane l owns bins [l*B, (l+1)*B)), DPP scan
//   MAD      rank select on the histogram folded around the median -- no second histogram
//
// 6.7 -> 3.9 ms per 1 M reads x 4 000 samples.  (Mean/std + mask variants of this shape were built and
// measured too, bit-exact but slower than the workgroup kernel's 7.5 ms: 13-22 ms when the std / mask
// passes re-read the samples from global memory with per-lane 2-byte loads, 10.4 ms with the compacted
// read in LDS -- 12 KB per wave leave 13 waves per CU.  zscale and the segmenter stay on k_prep_i16.)
//
// Reference: scale_outliers MotifSeq.py:317-324, medmad MotifSeq.py:192-200.
#include "sk_common.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int WPB = 4;            // wavefronts (= reads in flight) per workgroup

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

// NQ: 16-byte chunks of histogram per lane (bins <= 256 NQ).
template <int NQ>
__global__ __launch_bounds__(64 * WPB, 8)
void k_prepw_medmad(const int16_t *__restrict__ sig, int64_t stride, const int32_t *__restrict__ len, int nreads,
                    int lo, int hi, int vec_ok, int16_t *__restrict__ comp, sk_prep *__restrict__ prep)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nbins = hi - lo - 1;                         // >= 1 (host)
    const int nb4 = (nbins + 3) & ~3;
    unsigned *hist = (unsigned *)lds_raw + (size_t)w * nb4;
    unsigned *hist_v = hist - (lo + 1);                    // hist_v[x] counts value x
    const int hb0 = lane * 4 * NQ;                         // first bin this lane owns

    const bool in_vec = (vec_ok & 1) != 0, out_vec = (vec_ok & 2) != 0;
    // the keep range as packed int16 pairs, for the "all eight samples survive" test
    const int lo1 = max(lo + 1, -32768), hi1 = min(hi - 1, 32767);
    const unsigned lo2 = (unsigned)(lo1 & 0xffff) * 0x10001u, hi2 = (unsigned)(hi1 & 0xffff) * 0x10001u;

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

    // my histogram starts zeroed; every lane clears the bins it owns after use
#pragma unroll
    for (int j = 0; j < NQ; j++)
        if (hb0 + 4 * j < nb4) *(uint4 *)(hist + hb0 + 4 * j) = make_uint4(0u, 0u, 0u, 0u);

    const int nwaves = gridDim.x * WPB;
    for (int r = blockIdx.x * WPB + w; r < nreads; r += nwaves) {
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
            continue;
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
    }
}

typedef void (*prepw_fn)(const int16_t *, int64_t, const int32_t *, int, int, int, int, int16_t *, sk_prep *);

} // namespace

// Returns SK_OK after launching, or 1 when this kernel does not apply (caller uses k_prep_i16).
int sk_launch_prepw_medmad(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len,
                           int32_t nreads, int32_t lo, int32_t hi, int16_t *d_comp, sk_prep *d_prep)
{
    if (getenv("SK_PREP_BLOCK")) return 1;                  // tuning / A-B switch: workgroup-per-read kernel
    const int64_t nbins = (int64_t)hi - lo - 1;
    if (nbins < 1 || nbins > 2048) return 1;
    const int per_lane = (int)((nbins + 63) / 64);
    prepw_fn fn = per_lane <= 16 ? k_prepw_medmad<4> : per_lane <= 20 ? k_prepw_medmad<5> : k_prepw_medmad<8>;
    const size_t nb4 = (size_t)((nbins + 3) & ~(int64_t)3);
    const size_t lds = (size_t)WPB * nb4 * 4;
    int per_cu = (int)((160 * 1024) / (lds + 256));
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) return 1;
    const int vec_ok = ((((uintptr_t)d_sig & 15) == 0 && (stride % 8) == 0) ? 1 : 0) |
                       ((((uintptr_t)d_comp & 15) == 0 && (stride % 8) == 0) ? 2 : 0);
    int rounds = 4;
    if (const char *e = getenv("SK_PREP_ROUNDS")) { int v = atoi(e); if (v > 0) rounds = v; }
    const long long g = (long long)c->num_cu * per_cu * rounds;
    const long long need = ((long long)nreads + WPB - 1) / WPB;
    const int grid = (int)(g > need ? need : g);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * WPB), lds, c->stream, d_sig, stride, d_len, nreads, lo, hi,
                       vec_ok, d_comp, d_prep);
    SK_HIP(hipGetLastError());
    return SK_OK;
}
