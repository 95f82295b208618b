// sk_prep.hip -- per-read filter + statistics ("prep") kernels for gfx950.
//
// Covers, per read (one 256-thread workgroup per read):
//   scale_outliers      segmenter.py:311-318 / MotifSeq.py:317-324   strict lo < x < hi, order kept
//   np.median           segmenter.py:410, MotifSeq.py:194            LDS counting histogram + rank select
//   MAD                 MotifSeq.py:195-196                          rank select on the histogram folded around the median
//   np.mean / np.std    segmenter.py:412, sklearn.scale (MotifSeq.py:187)
//                       summed in numpy's exact order: 8192-element chunks accumulated serially,
//                       each chunk by the pairwise tree (leaves <= 128: eight strided accumulators,
//                       ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), serial tail) so top/bot are bit-exact
//   top / bot           segmenter.py:413-414
//   in-band bit mask    the `a < top and a > bot` test of segmenter.py:431 for every sample
//
// int16 samples are integers, so median and MAD are exact in half-integer arithmetic and the
// sum for the mean is exact in int64; only sum((x-mean)^2) depends on the order of additions.
// HBM traffic per read: M*2 B in, n*2 B out (compacted samples, needed in order by the DTW /
// segment kernels), 48 B of statistics, n/8 B of mask (segmenter only).
#include "sk_common.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int TPB = 256;
constexpr int NWAVE = TPB / 64;
constexpr int NPY_BUFSIZE = 8192;     // numpy ufunc buffer: reduction chunk
constexpr int PW_BLOCK = 128;         // numpy pairwise leaf size
constexpr int MAX_NODES = 256;        // heap of the pairwise tree of one chunk (8 levels)

// Fixed scratch at the front of dynamic LDS.
struct Scratch {
    int       wsum[2][NWAVE];         // per-wave counts (double buffered by iteration parity)
    long long wred[NWAVE];
    int       sel[4];                 // rank-select results
    double    node_sum[MAX_NODES];    // pairwise-tree partial sums, heap order (node 1 = chunk)
    double    bcast[2];
    // numpy's pairwise split tree of a chunk of tree_m elements, heap order: start << 16 | len,
    // 0 = no such node.  Built by wave 0 and kept across reads (a persistent workgroup mostly
    // sees one or two distinct lengths).
    unsigned  tab[MAX_NODES];
    unsigned short leaf[MAX_NODES / 2];   // heap ids of the leaves (len <= PW_BLOCK), any order
    int       tree_m, tree_depth, nleaf, pad_;
};
static_assert(sizeof(Scratch) % 16 == 0, "the histograms / sample buffer behind Scratch are 16-byte aligned");

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, which would
// serialise the global loads we keep in flight across the statistics phase.  Use it wherever the
// barrier protects LDS data; keep __syncthreads() where global writes must become visible.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Inclusive scan across the wavefront on the vector ALU (DPP): four shifts inside each row of 16,
// then lane 15 of rows 0/2 into rows 1/3 and lane 31 into the upper half.  No LDS round trips.
__device__ __forceinline__ int wave_incl_scan(int v, int /*lane*/)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2, 3
    return v;
}

// lane i <- lane i + N inside a row of 16 (lanes without a source keep their own value)
template <int N>
__device__ __forceinline__ double dpp_shl_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x100 + N, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x100 + N, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// Block-wide exclusive scan of one int per thread; returns exclusive prefix, *total = block sum.
__device__ __forceinline__ int block_excl_scan(int v, int *wsum /*[NWAVE]*/, int *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = wave_incl_scan(v, lane);
    if (lane == 63) wsum[w] = inc;
    lds_barrier();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NWAVE; i++) {
        int s = wsum[i];
        if (i < w) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc - v;
}

// Find the bins holding ranks k1 <= k2 of a histogram (counts sum to n).  Result in sc->sel[0..1].
// Every thread scans a contiguous slice of bins.
template <typename Count>
__device__ __forceinline__ void rank_select2_fn(Count count, int nbins, int k1, int k2, Scratch *sc, int parity)
{
    const int per = (nbins + TPB - 1) / TPB;
    const int b0 = threadIdx.x * per;
    const int b1 = min(nbins, b0 + per);
    int local = 0;
    for (int b = b0; b < b1; b++) local += (int)count(b);
    int total;
    int pre = block_excl_scan(local, sc->wsum[parity], &total);
    if (local > 0) {
        if (k1 >= pre && k1 < pre + local) {
            int acc = pre;
            for (int b = b0; b < b1; b++) { acc += (int)count(b); if (k1 < acc) { sc->sel[0] = b; break; } }
        }
        if (k2 >= pre && k2 < pre + local) {
            int acc = pre;
            for (int b = b0; b < b1; b++) { acc += (int)count(b); if (k2 < acc) { sc->sel[1] = b; break; } }
        }
    }
    lds_barrier();
}

__device__ void rank_select2(const unsigned *hist, int nbins, int k1, int k2, Scratch *sc, int parity)
{
    rank_select2_fn([&](int b) { return hist[b]; }, nbins, k1, k2, sc, parity);
}

// The same for histograms of up to 1024 * NQ bins, held in registers: thread t owns bins
// [4 NQ t, 4 NQ (t+1)) (NQ 16-byte LDS reads), so the prefix scan, the search inside the owning
// thread and the weighted sum sum(count * bin) all run without touching LDS again.  cnt[] is
// left to the caller (MAD histogram) and *wsum_out gets sum(count * bin) when `weighted`;
// that sum is kept in 32 bits: the caller guarantees samples * bins < 2^31.
// `nb4` = number of allocated bins, a multiple of 4 (bins past the real ones are zero).
template <int NQ>
__device__ __forceinline__ void rank_select2_regs(const unsigned *hist, int nb4, int k1, int k2, Scratch *sc, int parity,
                                                  bool weighted, long long *wsum_out, unsigned (&cnt)[4 * NQ])
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b0 = tid * 4 * NQ;
#pragma unroll
    for (int j = 0; j < NQ; j++) {
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        if (b0 + 4 * j < nb4) q = *(const uint4 *)(hist + b0 + 4 * j);
        cnt[4 * j] = q.x; cnt[4 * j + 1] = q.y; cnt[4 * j + 2] = q.z; cnt[4 * j + 3] = q.w;
    }
    int local = 0, wloc = 0;
#pragma unroll
    for (int i = 0; i < 4 * NQ; i++) { local += (int)cnt[i]; wloc += (int)cnt[i] * (b0 + i); }
    const int inc = wave_incl_scan(local, lane);
    int winc = 0;
    if (weighted) winc = wave_incl_scan(wloc, lane);
    if (lane == 63) { sc->wsum[parity][w] = inc; sc->wred[w] = (long long)winc; }
    lds_barrier();
    int base = 0;
    long long wtot = 0;
#pragma unroll
    for (int i = 0; i < NWAVE; i++) {
        const int t = sc->wsum[parity][i];
        if (i < w) base += t;
        wtot += sc->wred[i];
    }
    if (wsum_out) *wsum_out = wtot;
    const int pre = base + inc - local;
    if (local > 0) {
        if (k1 >= pre && k1 < pre + local) {
            int acc = pre, idx = b0;
#pragma unroll
            for (int i = 0; i < 4 * NQ; i++) { acc += (int)cnt[i]; idx += (acc <= k1) ? 1 : 0; }
            sc->sel[0] = idx;
        }
        if (k2 >= pre && k2 < pre + local) {
            int acc = pre, idx = b0;
#pragma unroll
            for (int i = 0; i < 4 * NQ; i++) { acc += (int)cnt[i]; idx += (acc <= k2) ? 1 : 0; }
            sc->sel[1] = idx;
        }
    }
    lds_barrier();
}

// numpy's pairwise split tree of a chunk of m elements (8 <= m <= 8192): a node longer than
// PW_BLOCK splits at n2 = (len/2) rounded down to a multiple of 8, at most 7 levels deep.  Wave 0
// fills sc->tab level by level (children from parents) and lists the leaves.
__device__ void build_tree(int m, Scratch *sc)
{
    const int lane = threadIdx.x;
    volatile unsigned *tab = sc->tab;
    if (lane == 0) { tab[1] = (unsigned)m; sc->nleaf = 0; }
    int depth = 0;
    for (int lvl = 1; lvl <= 7; lvl++) {
        __builtin_amdgcn_wave_barrier();
        bool any = false;
        for (int id = (1 << lvl) + lane; id < (2 << lvl); id += 64) {
            const unsigned par = tab[id >> 1];
            const int plen = (int)(par & 0xffffu), ps = (int)(par >> 16);
            unsigned me = 0u;
            if (plen > PW_BLOCK) {
                int n2 = plen / 2;
                n2 -= n2 % 8;
                me = (id & 1) ? ((unsigned)(ps + n2) << 16 | (unsigned)(plen - n2)) : ((unsigned)ps << 16 | (unsigned)n2);
            }
            tab[id] = me;
            any |= (me != 0u);
        }
        if (__ballot(any)) depth = lvl;
    }
    __builtin_amdgcn_wave_barrier();
    for (int id = 1 + lane; id < MAX_NODES; id += 64) {
        const unsigned e = tab[id];
        if (e != 0u && (int)(e & 0xffffu) <= PW_BLOCK) sc->leaf[atomicAdd(&sc->nleaf, 1)] = (unsigned short)id;
    }
    if (lane == 0) { sc->tree_m = m; sc->tree_depth = depth; }
}

// Sum of term(i), i in [0, m), in the order numpy's pairwise_sum uses (m <= 8192).
// term(i) must be a pure function.  All 256 threads call; result returned to every thread.
// Leaves (<= 128 elements) are summed by 8-lane groups -- lane j owns numpy's accumulator
// r[j], the butterfly shfl_xor 1,2,4 reproduces ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) -- then
// wave 0 folds the partial sums bottom-up (parent = left + right), wave-synchronously.
// BCAST = false: only wavefront 0 gets the result (the caller continues in thread 0), which saves the
// two barriers of the broadcast.
template <bool BCAST, int NT = TPB, typename Term>
__device__ double pairwise_chunk(int m, Scratch *sc, Term term)
{
    const int tid = threadIdx.x;
    if (m < 8) {                                   // numpy: plain serial loop from 0.0
        double res = 0.0;
        if (tid < 64) for (int i = 0; i < m; i++) res += term(i);
        if (!BCAST) return res;
        if (tid == 0) sc->bcast[0] = res;
        lds_barrier();
        double r = sc->bcast[0];
        lds_barrier();
        return r;
    }
    if (sc->tree_m != m) {                         // (block-uniform: written before the last barrier)
        lds_barrier();                             // nobody still reads the old tree
        if (tid < 64) build_tree(m, sc);
        lds_barrier();
    }
    const int grp = tid >> 3, j = tid & 7;
    const int nleaf = sc->nleaf;
    for (int li = grp; li < nleaf; li += NT / 8) {
        const int id = sc->leaf[li];
        const unsigned e = sc->tab[id];
        const int s = (int)(e >> 16), len = (int)(e & 0xffffu);
        double r = term(s + j);
        const int full = len - (len % 8);
        for (int i = 8; i < full; i += 8) r += term(s + i + j);
        // lane j = 0 of the group needs ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)): three DPP shifts
        r += dpp_shl_f64<1>(r);                     // (r0+r1) . (r2+r3) . (r4+r5) . (r6+r7) .
        r += dpp_shl_f64<2>(r);                     // ((r0+r1)+(r2+r3)) . . . ((r4+r5)+(r6+r7)) ...
        r += dpp_shl_f64<4>(r);
        if (j == 0) {
            for (int i = full; i < len; i++) r += term(s + i);
            sc->node_sum[id] = r;
        }
    }
    lds_barrier();
    if (tid < 64) {
        volatile double *ns = sc->node_sum;
        for (int lvl = sc->tree_depth - 1; lvl >= 0; lvl--) {
            for (int id = (1 << lvl) + tid; id < (2 << lvl); id += 64)
                if ((int)(sc->tab[id] & 0xffffu) > PW_BLOCK) ns[id] = ns[2 * id] + ns[2 * id + 1];
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (!BCAST) return (tid < 64) ? ((volatile double *)sc->node_sum)[1] : 0.0;
    lds_barrier();
    double res = sc->node_sum[1];
    lds_barrier();
    return res;
}

// np.add.reduce order over n terms: serial over 8192-chunks of pairwise sums.
template <bool BCAST = true, typename Term>
__device__ double numpy_sum(int n, Scratch *sc, Term term)
{
    double res = 0.0;
    for (int base = 0; base < n; base += NPY_BUFSIZE) {
        const int m = min(NPY_BUFSIZE, n - base);
        if (!BCAST && base > 0) lds_barrier();     // wave 0 is done with the previous chunk's partial sums
        res += pairwise_chunk<BCAST>(m, sc, [&](int i) { return term(base + i); });
    }
    return res;
}

// ------------------------------------------------------------------------------------------
// int16 reads
// ------------------------------------------------------------------------------------------
// LDSCOMP: the compacted samples also live in LDS (reads up to lds_cap samples), so the std
// leaf sums and the in-band classification never go back to global memory; the segmenter
// variant then writes no compacted samples to HBM at all (the walk kernel only needs the mask).
typedef short i16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned clamp_pk_i16(unsigned q, unsigned lo2, unsigned hi2)
{
    const i16x2 x = __builtin_bit_cast(i16x2, q);
    const i16x2 c = __builtin_elementwise_min(__builtin_elementwise_max(x, __builtin_bit_cast(i16x2, lo2)),
                                              __builtin_bit_cast(i16x2, hi2));
    return __builtin_bit_cast(unsigned, c);
}

// eight consecutive samples (four packed pairs) to dst; al = (element offset of dst) mod 8 when the
// row base is 16-byte aligned, odd when nothing is known
template <typename P>
__device__ __forceinline__ void put8(P *dst, const unsigned (&q)[4], int al)
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

// WINDOWED: statistics over filtered samples [t0, t1) only (dRNA_segmenter.py:109-110)
// MEDMAD: the medmad variant (its own instantiation: it shares no statistics code with the
// mean/std variants, and one kernel carrying both runs out of registers)
// LISTED: the reads are list[0 .. *count) (the streaming segmenter's uncertified reads, sk_segstat.hip); the
// mask goes out as {in band, kept} bytes in RAW sample coordinates into that path's per-read entries.
struct ListedArgs {
    const int32_t *list;
    const int32_t *count;
    unsigned char *mask2;
    int            row16;
};

template <bool LDSCOMP, bool WINDOWED, bool MEDMAD, bool LISTED = false>
__global__ __launch_bounds__(TPB, 8) __attribute__((amdgpu_num_sgpr(80)))
void k_prep_i16(const int16_t *__restrict__ sig, int64_t stride, const int32_t *__restrict__ len, int nreads,
                int lo, int hi, int mode, double std_scale, int vec_ok, int t0, int t1,
                int16_t *__restrict__ comp, sk_prep *__restrict__ prep,
                uint64_t *__restrict__ maskT, int64_t mask_rows, ListedArgs la)
{
    if constexpr (LISTED) nreads = *la.count;
    auto rid = [&](int k) -> int { if constexpr (LISTED) return la.list[k]; else return k; };
    extern __shared__ __align__(16) unsigned char lds_raw[];
    Scratch *sc = (Scratch *)lds_raw;
    unsigned *hist = (unsigned *)(lds_raw + sizeof(Scratch));
    const int nbins = max(0, hi - lo - 1);                 // values lo+1 .. hi-1
    const int nb4 = (nbins + 3) & ~3;                      // (every LDS array starts 16-byte aligned)
    int16_t *lcomp = (int16_t *)(hist + nb4);
    // histograms small enough to be ranked from registers (rank_select2_regs)
    const bool small_hist = nbins <= 2048;
    bool dirty = true;                                     // histograms need zeroing before use
    unsigned *hist_v = hist - (lo + 1);                    // hist_v[x] counts value x

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool to_global = MEDMAD || !(LDSCOMP && (mode == SK_PREP_SEGMENT || mode == SK_PREP_DRNA));
    if (!WINDOWED) { t0 = 0; t1 = 0x7fffffff; }
    // bit 0: sig rows 16-byte aligned, bit 1: comp rows too
    const bool in_vec = (vec_ok & 1) != 0, out_vec = (vec_ok & 2) != 0;
    // the keep range as packed int16 pairs, for the "all eight samples survive" test
    const int lo1 = max(lo + 1, -32768), hi1 = min(hi - 1, 32767);
    const bool range_ok = lo1 <= hi1;
    const unsigned lo2 = (unsigned)(lo1 & 0xffff) * 0x10001u, hi2 = (unsigned)(hi1 & 0xffff) * 0x10001u;

    // Persistent workgroups: each walks reads r, r + grid, ... and has the first tile of its
    // next read in flight while it does the statistics of the current one.  Samples stay packed
    // (two per register) until they are scattered.
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
    auto sample = [](const unsigned (&q)[4], int k) -> int {
        return (k & 1) ? (int)q[k >> 1] >> 16 : (int)(short)(q[k >> 1] & 0xffffu);
    };

    if (tid == 0) sc->tree_m = -1;                         // no pairwise tree cached yet
    unsigned v[4], vn[4];
    // a length outside [0, stride] would walk into the neighbouring rows: clamp (the host entry points
    // reject such input; device-resident callers get the clamp)
    const int maxM = (int)min(stride, (int64_t)0x7fffff00);
    auto rdlen = [&](int rr) { return min(max(len[rr], 0), maxM); };
    int rnext = ((int)blockIdx.x < nreads) ? rid(blockIdx.x) : 0;
    int Mnext = ((int)blockIdx.x < nreads) ? rdlen(rnext) : 0;
    if ((int)blockIdx.x < nreads) load8(sig + (int64_t)rnext * stride, Mnext, tid * 8, v);
    for (int rk = blockIdx.x; rk < nreads; rk += gridDim.x) {
    const int r = rnext;                                   // the read (rk: its position in the launch / list)
    const int M = Mnext;
    const int16_t *row = sig + (int64_t)r * stride;
    // (LISTED without the LDS-resident copy: one scratch row per persistent workgroup is enough)
    int16_t *crow = comp + (int64_t)(LISTED ? (int)blockIdx.x : r) * stride;
    if (dirty) for (int b = tid; b < nb4; b += TPB) hist[b] = 0u;
    dirty = false;
    if (tid < 4) sc->sel[tid] = 0;
    lds_barrier();

    // ---- pass A: filter, compact (order preserving), histogram ----------------------------
    int run = 0;                                           // survivors so far (block uniform)
    int parity = 0;
    for (int base = 0; base < M; base += TPB * 8, parity ^= 1) {
        const int i0 = base + tid * 8;
        if (base + TPB * 8 < M) load8(row, M, i0 + TPB * 8, vn);   // prefetch the next tile
        // Outliers are rare: when every sample of the wavefront's 512 survives (clamping the
        // packed pairs to the keep range changes nothing) there is nothing to scan or to test.
        unsigned changed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) changed |= clamp_pk_i16(v[k], lo2, hi2) ^ v[k];
        const bool wave_all = __all(range_ok && i0 + 8 <= M && changed == 0u);
        unsigned keep = 0xffu;
        int cnt = 8, inc = 8 * (lane + 1);
        if (!wave_all) {
            keep = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int x = sample(v, k);
                if (i0 + k < M && x > lo && x < hi) keep |= 1u << k;
            }
            cnt = __popc(keep);
            inc = wave_incl_scan(cnt, lane);
        }
        if (lane == 63) sc->wsum[parity][w] = inc;
        lds_barrier();
        int wbase = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < NWAVE; i++) {
            const int s = sc->wsum[parity][i];
            if (i < w) wbase += s;
            tot += s;
        }
        int o = run + wbase + inc - cnt;
        if (wave_all) {
            const int al = (run + wbase) & 7;              // wave uniform: o = run + wbase + 8 * lane
            if (LDSCOMP) put8(lcomp + o, v, al);
            if (to_global) put8(crow + o, v, out_vec ? al : 1);
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (!WINDOWED || (o + k >= t0 && o + k < t1)) atomicAdd(&hist_v[sample(v, k)], 1u);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (keep & (1u << k)) {
                    const int x = sample(v, k);
                    if (LDSCOMP) lcomp[o] = (int16_t)x;
                    if (to_global) crow[o] = (int16_t)x;
                    if (!WINDOWED || (o >= t0 && o < t1)) atomicAdd(&hist_v[x], 1u);   // statistics window
                    o++;
                }
            }
        }
        run += tot;
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = vn[k];
    }
    const int n = run;
    {                                                      // first tile of my next read
        const int rn = rk + gridDim.x;
        rnext = (rn < nreads) ? rid(rn) : 0;
        Mnext = (rn < nreads) ? rdlen(rnext) : 0;
        if (rn < nreads) load8(sig + (int64_t)rnext * stride, Mnext, tid * 8, v);
    }
    __syncthreads();                                       // histogram + compacted samples complete

    sk_prep pr;
    pr.n = n; pr.flags = 0; pr.center = 0.0; pr.scale = 1.0; pr.top = 0.0; pr.bot = 0.0;
    if (n == 0) {
        pr.flags = SK_FLAG_EMPTY;
        const double qnan = __builtin_nan("");
        pr.center = qnan; pr.scale = qnan; pr.top = qnan; pr.bot = qnan;
        if (tid == 0) prep[r] = pr;
        lds_barrier();
        continue;
    }

    // statistics are taken over filtered samples [w0, w0 + ns): everything, or the slice
    // sig[t_start:t_end] of dRNA_segmenter.py:109-110
    const int w0 = min(n, t0);
    const int ns = min(n, t1) - w0;
    if (ns <= 0) {                                         // empty slice: numpy gives NaN, band is empty
        const double qnan = __builtin_nan("");
        pr.center = qnan; pr.scale = qnan; pr.top = qnan; pr.bot = qnan;
        if (tid == 0) prep[r] = pr;
        if (maskT != nullptr)
            for (int wi = tid; wi * 64 < n; wi += TPB) maskT[(int64_t)wi * mask_rows + r] = 0ull;
        lds_barrier();
        continue;
    }

    // ---- median: ranks (ns-1)/2 and ns/2 of the value histogram -----------------------
    // (registers when the histograms are small; every thread then clears the bins it owns, so the
    // next read starts from zeroed histograms without a separate pass)
    const bool regs = small_hist && M < (1 << 19);         // sum(count * bin) stays below 2^31
    long long S = 0;
    unsigned cnt[8];
    const int hb0 = tid * 8;                               // first bin this thread owns
    if (regs) {
        long long wsum;
        rank_select2_regs<2>(hist, nb4, (ns - 1) / 2, ns / 2, sc, 0, !MEDMAD, &wsum, cnt);
        S = wsum + (long long)ns * (lo + 1);
        if (!MEDMAD) {                                     // (medmad reads the histogram once more)
#pragma unroll
            for (int j = 0; j < 2; j++)
                if (hb0 + 4 * j < nb4) *(uint4 *)(hist + hb0 + 4 * j) = make_uint4(0u, 0u, 0u, 0u);
        }
    } else {
        rank_select2(hist, nbins, (ns - 1) / 2, ns / 2, sc, 0);
        dirty = true;
    }
    const int med2 = (sc->sel[0] + lo + 1) + (sc->sel[1] + lo + 1);    // 2 * median, exact
    const double median = (double)med2 * 0.5;
    if (MEDMAD) lds_barrier();                             // (the MAD select below rewrites sc->sel)

    if constexpr (MEDMAD) {
        // MAD = median of |x - med|: rank select on the value histogram folded around the median.
        // |2x - med2| takes the values 2t (med2 even) or 2t + 1 (odd); f(t) = count(left) + count(right).
        const int odd = med2 & 1;
        const int cl = ((med2 - odd) >> 1) - (lo + 1);                 // bin just below / at the median
        const int cr = cl + odd;
        rank_select2_fn([&](int t) -> unsigned {
            const int bl = cl - t, br = cr + t;
            unsigned c = 0u;
            if (bl >= 0 && bl < nbins) c += hist[bl];
            if (br >= 0 && br < nbins && (odd || t > 0)) c += hist[br];
            return c;
        }, nbins, (ns - 1) / 2, ns / 2, sc, 1);
        if (regs) {                                                    // all reads done (barrier above)
#pragma unroll
            for (int j = 0; j < 2; j++)
                if (hb0 + 4 * j < nb4) *(uint4 *)(hist + hb0 + 4 * j) = make_uint4(0u, 0u, 0u, 0u);
        }
        const double mad = (double)((2 * sc->sel[0] + odd) + (2 * sc->sel[1] + odd)) * 0.25;   // (d1/2 + d2/2) / 2
        pr.center = median;
        pr.scale = mad * 1.4826;                                       // MotifSeq.py:196
        if (mad == 0.0) pr.flags |= SK_FLAG_DEGENERATE;
        if (tid == 0) prep[r] = pr;
        lds_barrier();
        continue;
    }

    if constexpr (!MEDMAD) {
    // ---- mean (exact integer sum, taken from the histogram) and numpy-order std -----------
    if (!regs) {
        long long isum = 0;
        for (int b = tid; b < nbins; b += TPB) isum += (long long)hist[b] * (long long)(b + lo + 1);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) isum += __shfl_xor(isum, d);
        if (lane == 0) sc->wred[w] = isum;
        lds_barrier();
#pragma unroll
        for (int i = 0; i < NWAVE; i++) S += sc->wred[i];
    }
    const double mean = (double)S / (double)ns;
    const int16_t *src = LDSCOMP ? (const int16_t *)lcomp : (const int16_t *)crow;
    const double ssq = numpy_sum<false>(ns, sc, [&](int i) {      // (result in wavefront 0 only)
        const double d = (double)src[w0 + i] - mean;
        return d * d;
    });
    // sqrt / divisions / thresholds once per read (thread 0), not once per wavefront
    if (tid == 0) {
        const double sd = sqrt(ssq / (double)ns);
        if (mode == SK_PREP_ZSCALE) {
            pr.center = mean;
            pr.scale = (sd == 0.0) ? 1.0 : sd;             // sklearn _handle_zeros_in_scale
        } else {
            // ---- segmenter thresholds (segmenter.py:413-414) ---------------------------------
            const double spread = sd * std_scale;
            const double top = median + spread;
            // dRNA_segmenter.py:111,114 tests `a < top` only
            const double bot = (mode == SK_PREP_DRNA) ? -__builtin_huge_val() : median - spread;
            pr.center = median; pr.scale = sd; pr.top = top; pr.bot = bot;
            // Samples are integers, so `a < top and a > bot` (segmenter.py:431) is the integer test
            // floor(bot) < a < ceil(top); clamped just outside the int16 range, NaN -> empty band.
            const double ct = ceil(top), fb = floor(bot);
            int itop = (top == top) ? (ct > 32768.0 ? 32768 : (ct < -32768.0 ? -32768 : (int)ct)) : -32768;
            int ibot = (bot == bot) ? (fb > 32767.0 ? 32767 : (fb < -32769.0 ? -32769 : (int)fb)) : 32767;
            sc->sel[2] = ibot + 1;                         // first in-band value
            sc->sel[3] = max(itop - ibot - 1, 0);          // number of in-band values
        }
        prep[r] = pr;
    }
    if (mode == SK_PREP_ZSCALE) {
        lds_barrier();
        continue;
    }
    lds_barrier();
    // ---- in-band mask: one compare per sample, the lane mask of the compare is the word -------
    const int first = sc->sel[2];
    const unsigned width = (unsigned)sc->sel[3];
    if constexpr (LISTED) {                                // raw coordinates: one byte of each mask per 8 samples
        unsigned char *mb = la.mask2 + (int64_t)r * la.row16 * 16;
        for (int base = 0; base < M; base += TPB * 8) {
            const int i0 = base + tid * 8;
            if (i0 < M) {
                unsigned q[4];
                load8(row, M, i0, q);
                unsigned in8 = 0, kp8 = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int x = sample(q, k);
                    const bool kept = i0 + k < M && x > lo && x < hi;
                    kp8 |= (kept ? 1u : 0u) << k;
                    in8 |= ((kept && (unsigned)(x - first) < width) ? 1u : 0u) << k;
                }
                mb[(i0 >> 6) * 16 + ((i0 & 63) >> 3)] = (unsigned char)in8;
                mb[(i0 >> 6) * 16 + 8 + ((i0 & 63) >> 3)] = (unsigned char)kp8;
            }
        }
        lds_barrier();                                     // LDS is reused by the next read
        continue;
    }
    for (int base = 0; base < n; base += TPB) {
        const int i = base + tid;
        bool in = false;
        if (i < n) in = (unsigned)((int)src[i] - first) < width;
        const unsigned long long bits = __ballot(in);
        if (lane == 0) maskT[(int64_t)(i >> 6) * mask_rows + r] = bits;
    }
    lds_barrier();                                       // LDS is reused by the next read
    }
    }
}


// ------------------------------------------------------------------------------------------
// dRNA_segmenter.py, slow5 branch (:85-176): statistics window + one-sided band, ONE look at the read (round 5)
// ------------------------------------------------------------------------------------------
// The branch takes median and std from the filtered samples [t_start, t_end) = [1 000, 5 000) and then tests `a < top`
// over the whole read (:109-114).  k_prep_i16<WINDOWED> does that as the general kernel does everything: compact the
// read to memory, statistics, then read the compacted samples back for the band test -- 6 bytes of traffic per sample
// for 2 bytes of input (10.2 ms per 250 000 reads of 17 300 samples, 0.10 of HBM).  But the statistics are complete as
// soon as t_end samples have survived the filter: this kernel collects those in LDS (they are needed there for numpy's
// summation order anyway), computes top, classifies what it has collected from LDS and every later tile straight from
// the registers it was loaded into.  The band bits are compacted into an LDS bit row (one atomic OR per thread and tile)
// and leave as the transposed words the scan reads.  Nothing is written but the mask and the 48-byte record.
// LDS: Scratch | hist[nb4] | lbits[ceil(stride / 64) * 2] | lcomp[t_end + 8 TPB] int16.
__global__ __launch_bounds__(TPB, 5)
void k_drna_stats(const int16_t *__restrict__ sig, int64_t stride, const int32_t *__restrict__ len, int nreads,
                  int lo, int hi, double std_scale, int vec_ok, int t0, int t1, int lbits_words,
                  sk_prep *__restrict__ prep, uint64_t *__restrict__ maskT, int64_t mask_rows)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    Scratch *sc = (Scratch *)lds_raw;
    unsigned *hist = (unsigned *)(lds_raw + sizeof(Scratch));
    const int nbins = max(0, hi - lo - 1);
    const int nb4 = (nbins + 3) & ~3;
    unsigned *lbits = hist + nb4;                           // the read's band bits, filtered coordinates
    int16_t *lcomp = (int16_t *)(lbits + lbits_words);      // the first t1 (+ one tile) filtered samples
    unsigned *hist_v = hist - (lo + 1);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool in_vec = (vec_ok & 1) != 0;
    const int lo1 = max(lo + 1, -32768), hi1 = min(hi - 1, 32767);
    const bool range_ok = lo1 <= hi1;
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
    auto sample = [](const unsigned (&q)[4], int k) -> int {
        return (k & 1) ? (int)q[k >> 1] >> 16 : (int)(short)(q[k >> 1] & 0xffffu);
    };
    if (tid == 0) sc->tree_m = -1;
    for (int b = tid; b < nb4; b += TPB) hist[b] = 0u;
    const int maxM = (int)min(stride, (int64_t)0x7fffff00);

    for (int r = blockIdx.x; r < nreads; r += gridDim.x) {
        const int M = min(max(len[r], 0), maxM);
        const int16_t *row = sig + (int64_t)r * stride;
        for (int i = tid; i < lbits_words; i += TPB) lbits[i] = 0u;
        if (tid < 4) sc->sel[tid] = 0;
        lds_barrier();

        sk_prep pr;
        pr.n = 0; pr.flags = 0; pr.center = 0.0; pr.scale = 1.0; pr.top = 0.0; pr.bot = 0.0;
        int run = 0, parity = 0;
        bool have = false;                                  // the statistics exist
        int itop = -32768;                                  // in band <=> x < itop
        unsigned v[4], vn[4];
        load8(row, M, tid * 8, v);
        // statistics + the band bits of everything collected so far (block-uniform call)
        auto finish_stats = [&](int nsofar) {
            __syncthreads();                                // histogram + collected samples complete
            const int w0 = min(nsofar, t0);
            const int ns = min(nsofar, t1) - w0;
            const double qnan = __builtin_nan("");
            if (ns <= 0) {                                  // empty slice: numpy gives NaN, the band is empty
                pr.center = qnan; pr.scale = qnan; pr.top = qnan; pr.bot = qnan;
                itop = -32768;
            } else {
                long long wsum;
                unsigned cnt[8];
                rank_select2_regs<2>(hist, nb4, (ns - 1) / 2, ns / 2, sc, 0, true, &wsum, cnt);
                const long long S = wsum + (long long)ns * (lo + 1);
                const int hb0 = tid * 8;
#pragma unroll
                for (int j = 0; j < 2; j++)
                    if (hb0 + 4 * j < nb4) *(uint4 *)(hist + hb0 + 4 * j) = make_uint4(0u, 0u, 0u, 0u);
                const int med2 = (sc->sel[0] + lo + 1) + (sc->sel[1] + lo + 1);
                const double median = (double)med2 * 0.5;
                const double mean = (double)S / (double)ns;
                const double ssq = numpy_sum<false>(ns, sc, [&](int i) {
                    const double d = (double)lcomp[w0 + i] - mean;
                    return d * d;
                });
                if (tid == 0) {
                    const double sd = sqrt(ssq / (double)ns);
                    const double top = median + sd * std_scale;          // dRNA_segmenter.py:111
                    sc->bcast[0] = median; sc->bcast[1] = sd;
                    const double ct = ceil(top);
                    sc->sel[2] = (top == top) ? (ct > 32768.0 ? 32768 : (ct < -32768.0 ? -32768 : (int)ct)) : -32768;
                }
                lds_barrier();
                itop = sc->sel[2];
                const double median_b = sc->bcast[0], sd_b = sc->bcast[1];
                pr.center = median_b; pr.scale = sd_b; pr.top = median_b + sd_b * std_scale; pr.bot = -__builtin_huge_val();
            }
            // the samples collected so far, from LDS: 64 consecutive ones per wavefront and ballot
            for (int base = 0; base < nsofar; base += TPB) {
                const int i = base + tid;
                const bool in = i < nsofar && (int)lcomp[i] < itop;
                const unsigned long long bits = __ballot(in);
                if (lane == 0) { lbits[2 * (i >> 6)] = (unsigned)bits; lbits[2 * (i >> 6) + 1] = (unsigned)(bits >> 32); }
            }
            have = true;
        };
        for (int base = 0; base < M; base += TPB * 8, parity ^= 1) {
            const int i0 = base + tid * 8;
            if (base + TPB * 8 < M) load8(row, M, i0 + TPB * 8, vn);
            unsigned changed = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) changed |= clamp_pk_i16(v[k], lo2, hi2) ^ v[k];
            const bool wave_all = __all(range_ok && i0 + 8 <= M && changed == 0u);
            unsigned keep = 0xffu;
            int cnt = 8, inc = 8 * (lane + 1);
            if (!wave_all) {
                keep = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int x = sample(v, k);
                    if (i0 + k < M && x > lo && x < hi) keep |= 1u << k;
                }
                cnt = __popc(keep);
                inc = wave_incl_scan(cnt, lane);
            }
            if (lane == 63) sc->wsum[parity][w] = inc;
            lds_barrier();
            int wbase = 0, tot = 0;
#pragma unroll
            for (int i = 0; i < NWAVE; i++) {
                const int s2 = sc->wsum[parity][i];
                if (i < w) wbase += s2;
                tot += s2;
            }
            int o = run + wbase + inc - cnt;
            if (!have) {                                    // collecting: samples to LDS, window samples into the histogram
                int oo = o;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (keep & (1u << k)) {
                        const int x = sample(v, k);
                        lcomp[oo] = (int16_t)x;
                        if (oo >= t0 && oo < t1) atomicAdd(&hist_v[x], 1u);
                        oo++;
                    }
                }
                run += tot;
                if (run >= t1 || base + TPB * 8 >= M) finish_stats(run);     // (block-uniform)
            } else {                                        // classifying: this tile's kept samples, straight from registers
                unsigned bits = 0;
                int nb = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (keep & (1u << k)) {
                        bits |= ((sample(v, k) < itop) ? 1u : 0u) << nb;
                        nb++;
                    }
                }
                if (bits) {
                    const int sh = o & 31;
                    atomicOr(&lbits[o >> 5], bits << sh);
                    if (sh + nb > 32) atomicOr(&lbits[(o >> 5) + 1], bits >> (32 - sh));
                }
                run += tot;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = vn[k];
        }
        const int n = run;
        pr.n = n;
        if (n == 0) {
            pr.flags = SK_FLAG_EMPTY;
            const double qnan = __builtin_nan("");
            pr.center = qnan; pr.scale = qnan; pr.top = qnan; pr.bot = qnan;
        }
        lds_barrier();                                      // every OR into the bit row has landed
        if (tid == 0) prep[r] = pr;
        for (int wi = tid; wi * 64 < n; wi += TPB)
            maskT[(int64_t)wi * mask_rows + r] = (unsigned long long)lbits[2 * wi] | ((unsigned long long)lbits[2 * wi + 1] << 32);
        lds_barrier();                                      // LDS is reused by the next read
    }
}

// ------------------------------------------------------------------------------------------
// float64 reads (pA TSVs, segmenter.py:198-199 / MotifSeq.py:270; fast5 input converted to pA)
// ------------------------------------------------------------------------------------------
// Values are arbitrary doubles, so the median is found by an MSD radix select over the
// order-preserving 64-bit image of each double: 8-bit digits, one 256-bin LDS histogram per
// pass (one bin per thread), starting below the bits that min and max have in common.

__device__ __forceinline__ unsigned long long f64_key(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

__device__ void block_minmax_u64(unsigned long long &mn, unsigned long long &mx, unsigned long long *tmp /*[2*NWAVE]*/)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long a = __shfl_xor(mn, d), b = __shfl_xor(mx, d);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if (lane == 0) { tmp[w] = mn; tmp[NWAVE + w] = mx; }
    lds_barrier();
    mn = tmp[0]; mx = tmp[NWAVE];
#pragma unroll
    for (int i = 1; i < NWAVE; i++) {
        mn = tmp[i] < mn ? tmp[i] : mn;
        mx = tmp[NWAVE + i] > mx ? tmp[NWAVE + i] : mx;
    }
    lds_barrier();
}

constexpr int SEL_LIST = 1024;        // candidates kept in LDS once the selected bin is this small

// Key of rank k (0-based) among key(0..n-1).  kmin/kmax: block-uniform extremes of the keys.
// MSD radix select with 8-bit digits.  As soon as the bin that holds rank k has at most SEL_LIST
// members they are gathered into LDS (one more sweep over the data) and the remaining digits are
// resolved on that list, so a select costs two or three sweeps over the read instead of seven.
template <typename KeyFn>
__device__ unsigned long long radix_select(int n, int k, Scratch *sc, unsigned *hist256, unsigned long long *list,
                                           KeyFn key, unsigned long long kmin, unsigned long long kmax)
{
    if (kmin == kmax) return kmin;
    const int tid = threadIdx.x;
    const int top = 63 - __clzll((long long)(kmin ^ kmax));
    int shift = (top / 8) * 8;
    unsigned long long pmask = (shift + 8 >= 64) ? 0ull : ~((1ull << (shift + 8)) - 1ull);
    unsigned long long prefix = kmin & pmask;
    int m = -1;                                     // members of `list` (-1: still sweeping the data)
    for (; shift >= 0; shift -= 8) {
        hist256[tid] = 0u;
        lds_barrier();
        if (m < 0) {
            for (int i = tid; i < n; i += TPB) {
                const unsigned long long kk = key(i);
                if ((kk & pmask) == prefix) atomicAdd(&hist256[(unsigned)(kk >> shift) & 255u], 1u);
            }
        } else {
            for (int i = tid; i < m; i += TPB) {
                const unsigned long long kk = list[i];
                if ((kk & pmask) == prefix) atomicAdd(&hist256[(unsigned)(kk >> shift) & 255u], 1u);
            }
        }
        lds_barrier();
        const int c = (int)hist256[tid];
        int total;
        const int excl = block_excl_scan(c, sc->wsum[0], &total);
        if (c > 0 && k >= excl && k < excl + c) { sc->sel[0] = tid; sc->sel[1] = k - excl; sc->sel[2] = c; }
        lds_barrier();
        prefix |= (unsigned long long)(unsigned)sc->sel[0] << shift;
        pmask |= 0xffull << shift;
        k = sc->sel[1];
        const int members = sc->sel[2];
        lds_barrier();
        if (m < 0 && shift > 0 && members <= SEL_LIST) {        // gather the bin's members once
            if (tid == 0) sc->sel[3] = 0;
            lds_barrier();
            for (int i = tid; i < n; i += TPB) {
                const unsigned long long kk = key(i);
                if ((kk & pmask) == prefix) list[atomicAdd(&sc->sel[3], 1)] = kk;
            }
            lds_barrier();
            m = members;
        }
    }
    return prefix;
}

// Median of n values given as keys; returns (a+b)/2 for even n like np.median.  The upper middle
// element is the lower one again if enough values are <= it, else the smallest value above it:
// one sweep (count, min) instead of a second select.
template <typename KeyFn>
__device__ double median_select(int n, Scratch *sc, unsigned *hist256, unsigned long long *list,
                                unsigned long long *mm, KeyFn key, unsigned long long kmin, unsigned long long kmax)
{
    const int k1 = (n - 1) / 2, k2 = n / 2;
    const unsigned long long ka = radix_select(n, k1, sc, hist256, list, key, kmin, kmax);
    const double a = key_f64(ka);
    if (k2 == k1) return a;
    int le = 0;
    unsigned long long above = ~0ull, dummy = 0ull;
    for (int i = threadIdx.x; i < n; i += TPB) {
        const unsigned long long kk = key(i);
        if (kk <= ka) le++;
        else above = kk < above ? kk : above;
    }
    int total;
    (void)block_excl_scan(le, sc->wsum[1], &total);
    block_minmax_u64(above, dummy, mm);
    const double b = (total > k2) ? a : key_f64(above);
    return (a + b) / 2.0;
}

// LISTED: the reads are list[0 .. *count) -- the streaming float64 kernel's uncertified reads (sk_f64stat.hip).
// Segmenter mode then compacts into one scratch row per workgroup and writes the masks as {in band, kept} bytes
// in RAW sample coordinates into that path's per-read entries; medmad mode writes comp / prep as usual.
struct ListedF64 {
    const int32_t *list;
    const int32_t *count;
    unsigned char *mask2;
    int            row16;
    int64_t        scratch_stride;      // doubles per scratch row (segmenter mode)
    const int32_t *len;                 // optional: read r is its first len[r] samples
};

struct PrepF64Shared {
    Scratch sc;
    unsigned hist256[256];
    unsigned long long mm[2 * NWAVE];
    unsigned long long sel_list[SEL_LIST];
};

// where a read's float64 samples come from: the caller's doubles, or (round 6) int16 raw samples through the pA
// conversion of segmenter.py:345-349 -- np.round((x + offset) * unit, 2) = rint(y * 100) / 100, value for value what
// k_rows_to_pa (sk_synth.hip) writes -- for the reads the raw-domain pA segmenter (k_seg_stats<.., PA>) cannot certify
struct RowF64 {
    const double *p;
    __device__ __forceinline__ double operator[](int i) const { return p[i]; }
};
struct RowPA {
    const int16_t *p;
    double offset, unit;
    __device__ __forceinline__ double operator[](int i) const
    {
        const double v = ((double)p[i] + offset) * unit;
        return rint(v * 100.0) / 100.0;
    }
};

template <bool LISTED, typename Row>
__device__ void prep_f64_core(PrepF64Shared *sh, int r, const Row row, int M, double *__restrict__ crow,
                              double lo, double hi, int mode, double std_scale, sk_prep *__restrict__ prep,
                              uint64_t *__restrict__ maskT, int64_t mask_rows, const ListedF64 &la);

template <bool LISTED>
__device__ void prep_f64_read(PrepF64Shared *sh, int r, const double *__restrict__ sig, const int64_t *__restrict__ off,
                              double lo, double hi, int mode, double std_scale,
                              double *__restrict__ comp, sk_prep *__restrict__ prep,
                              uint64_t *__restrict__ maskT, int64_t mask_rows, const ListedF64 &la)
{
    const int64_t o0 = off[r];
    int M = (int)(off[r + 1] - o0);
    if (la.len) M = min(M, max(la.len[r], 0));
    double *crow = (LISTED && mode == SK_PREP_SEGMENT) ? comp + (int64_t)blockIdx.x * la.scratch_stride : comp + o0;
    prep_f64_core<LISTED>(sh, r, RowF64{sig + o0}, M, crow, lo, hi, mode, std_scale, prep, maskT, mask_rows, la);
}

template <bool LISTED, typename Row>
__device__ void prep_f64_core(PrepF64Shared *sh, int r, const Row row, int M, double *__restrict__ crow,
                              double lo, double hi, int mode, double std_scale, sk_prep *__restrict__ prep,
                              uint64_t *__restrict__ maskT, int64_t mask_rows, const ListedF64 &la)
{
    Scratch *sc = &sh->sc;
    unsigned *hist256 = sh->hist256;
    unsigned long long *mm = sh->mm;
    unsigned long long *sel_list = sh->sel_list;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < 4) sc->sel[tid] = 0;
    if (tid == 0) sc->tree_m = -1;

    // ---- pass A: filter + order-preserving compaction + key extremes --------------------
    unsigned long long kmin = ~0ull, kmax = 0ull;
    int run = 0, parity = 0;
    for (int base = 0; base < M; base += TPB * 4, parity ^= 1) {
        const int i0 = base + tid * 4;
        double v[4];
        unsigned keep = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v[k] = (i0 + k < M) ? row[i0 + k] : 0.0;
            if (i0 + k < M && v[k] > lo && v[k] < hi) keep |= 1u << k;
        }
        const int cnt = __popc(keep);
        const int inc = wave_incl_scan(cnt, lane);
        if (lane == 63) sc->wsum[parity][w] = inc;
        lds_barrier();
        int wbase = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < NWAVE; i++) {
            const int s = sc->wsum[parity][i];
            if (i < w) wbase += s;
            tot += s;
        }
        int o = run + wbase + inc - cnt;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (keep & (1u << k)) {
                crow[o++] = v[k];
                const unsigned long long kk = f64_key(v[k]);
                kmin = kk < kmin ? kk : kmin;
                kmax = kk > kmax ? kk : kmax;
            }
        }
        run += tot;
    }
    const int n = run;
    __syncthreads();

    sk_prep pr;
    pr.n = n; pr.flags = 0; pr.center = 0.0; pr.scale = 1.0; pr.top = 0.0; pr.bot = 0.0;
    if (n == 0) {
        pr.flags = SK_FLAG_EMPTY;
        const double qnan = __builtin_nan("");
        pr.center = qnan; pr.scale = qnan; pr.top = qnan; pr.bot = qnan;
        if (tid == 0) prep[r] = pr;
        return;
    }
    block_minmax_u64(kmin, kmax, mm);

    double median = 0.0;
    if (mode != SK_PREP_ZSCALE)
        median = median_select(n, sc, hist256, sel_list, mm, [&](int i) { return f64_key(crow[i]); }, kmin, kmax);

    if (mode == SK_PREP_MEDMAD) {
        // MAD = median(|x - med|)   MotifSeq.py:195
        unsigned long long dmin = ~0ull, dmax = 0ull;
        for (int i = tid; i < n; i += TPB) {
            const unsigned long long kk = f64_key(fabs(crow[i] - median));
            dmin = kk < dmin ? kk : dmin;
            dmax = kk > dmax ? kk : dmax;
        }
        block_minmax_u64(dmin, dmax, mm);
        const double mad = median_select(n, sc, hist256, sel_list, mm,
                                         [&](int i) { return f64_key(fabs(crow[i] - median)); }, dmin, dmax);
        pr.center = median;
        pr.scale = mad * 1.4826;
        if (mad == 0.0) pr.flags |= SK_FLAG_DEGENERATE;
        if (tid == 0) prep[r] = pr;
        return;
    }

    // ---- numpy-order mean and std ---------------------------------------------------------
    const double mean = numpy_sum(n, sc, [&](int i) { return crow[i]; }) / (double)n;
    const double ssq = numpy_sum(n, sc, [&](int i) {
        const double d = crow[i] - mean;
        return d * d;
    });
    const double sd = sqrt(ssq / (double)n);
    if (mode == SK_PREP_ZSCALE) {
        pr.center = mean;
        pr.scale = (sd == 0.0) ? 1.0 : sd;
        // sklearn's scale() subtracts the residual mean again when the mean of the centred (and, later,
        // of the scaled) data is not within 1e-8 of zero (np.allclose).  With |mean error| <= ~60 eps
        // max|x| that needs max|x| >~ 7e5, or a scale below ~1.4e-6 max|x|: only such reads pay for the
        // two extra numpy-order sums.  pr.top / pr.bot carry the two corrections (0.0 when not applied)
        // to the sample feed: v = ((x - mean) - top) / scale - bot.
        const double maxabs = fmax(fabs(key_f64(kmin)), fabs(key_f64(kmax)));
        if (sd != 0.0 && (!(maxabs < 1e4) || !(sd > 1e-4 * maxabs))) {
            const double m1 = numpy_sum(n, sc, [&](int i) { return crow[i] - mean; }) / (double)n;
            const double c1 = (fabs(m1) <= 1e-8) ? 0.0 : m1;           // np.allclose(mean_1, 0)
            const double m2 = numpy_sum(n, sc, [&](int i) { return ((crow[i] - mean) - c1) / pr.scale; }) / (double)n;
            const double c2 = (fabs(m2) <= 1e-8) ? 0.0 : m2;
            pr.top = c1; pr.bot = c2;
            if (c1 != 0.0 || c2 != 0.0) pr.flags |= SK_FLAG_RECENTRE;
        }
        if (tid == 0) prep[r] = pr;
        return;
    }
    const double spread = sd * std_scale;
    const double top = median + spread;
    const double bot = median - spread;
    pr.center = median; pr.scale = sd; pr.top = top; pr.bot = bot;
    if (tid == 0) prep[r] = pr;
    if constexpr (LISTED) {                                // raw coordinates: one byte of each mask per 8 samples
        unsigned char *mb = la.mask2 + (int64_t)r * la.row16 * 16;
        for (int base = 0; base < M; base += TPB * 8) {
            const int i0 = base + tid * 8;
            if (i0 < M) {
                unsigned in8 = 0, kp8 = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const double a = (i0 + k < M) ? row[i0 + k] : 0.0;
                    const bool kept = i0 + k < M && a > lo && a < hi;
                    kp8 |= (kept ? 1u : 0u) << k;
                    in8 |= ((kept && a < top && a > bot) ? 1u : 0u) << k;
                }
                mb[(i0 >> 6) * 16 + ((i0 & 63) >> 3)] = (unsigned char)in8;
                mb[(i0 >> 6) * 16 + 8 + ((i0 & 63) >> 3)] = (unsigned char)kp8;
            }
        }
        return;
    }
    for (int base = 0; base < n; base += TPB) {
        const int i = base + tid;
        bool in = false;
        if (i < n) {
            const double a = crow[i];
            in = (a < top) && (a > bot);
        }
        const unsigned long long bits = __ballot(in);
        if (lane == 0) maskT[(int64_t)(i >> 6) * mask_rows + r] = bits;
    }
}

template <bool LISTED>
__global__ __launch_bounds__(TPB)
void k_prep_f64(const double *__restrict__ sig, const int64_t *__restrict__ off, int nreads,
                double lo, double hi, int mode, double std_scale,
                double *__restrict__ comp, sk_prep *__restrict__ prep,
                uint64_t *__restrict__ maskT, int64_t mask_rows, ListedF64 la)
{
    __shared__ PrepF64Shared sh;
    if constexpr (LISTED) {
        const int cnt = *la.count;
        for (int k = blockIdx.x; k < cnt; k += gridDim.x) {
            __syncthreads();                               // the previous read is done with the shared scratch
            prep_f64_read<true>(&sh, la.list[k], sig, off, lo, hi, mode, std_scale, comp, prep, maskT, mask_rows, la);
        }
    } else {
        prep_f64_read<false>(&sh, blockIdx.x, sig, off, lo, hi, mode, std_scale, comp, prep, maskT, mask_rows, la);
    }
}

// The numpy-order redo of the raw-domain pA segmenter's uncertified reads (k_seg_stats<.., PA>, sk_segstat.hip): read
// list[k] is sig[r * stride .. + len[r]) through its own {offset, unit}; one scratch row of doubles per workgroup;
// prep records and the {in band, kept} entries of those reads are rewritten in place.
__global__ __launch_bounds__(TPB)
void k_prep_pa_listed(const int16_t *__restrict__ sig, int64_t stride, const int32_t *__restrict__ len,
                      const double *__restrict__ cal, double lo, double hi, double std_scale,
                      double *__restrict__ scratch, sk_prep *__restrict__ prep, ListedF64 la)
{
    __shared__ PrepF64Shared sh;
    const int cnt = *la.count;
    for (int k = blockIdx.x; k < cnt; k += gridDim.x) {
        __syncthreads();                                   // the previous read is done with the shared scratch
        const int r = la.list[k];
        const int M = (int)min((int64_t)max(len[r], 0), stride);
        prep_f64_core<true>(&sh, r, RowPA{sig + (int64_t)r * stride, cal[2 * r], cal[2 * r + 1]}, M,
                            scratch + (int64_t)blockIdx.x * la.scratch_stride, lo, hi, SK_PREP_SEGMENT, std_scale, prep,
                            (uint64_t *)nullptr, (int64_t)0, la);
    }
}

// ------------------------------------------------------------------------------------------
// dRNA_segmenter.py --signal branch (:272-326): rolling mean of the filtered signal
// ------------------------------------------------------------------------------------------
// t = Series(filtered).rolling(window=w).mean(): pandas' roll_mean keeps a Kahan-compensated running
// sum (add the entering sample, remove the leaving one) and divides by the count.  The samples are
// integers and every window sum is far below 2^53, so all of those additions are exact and
// t[i] = (double)(window sum) / w with ONE rounding -- whatever the order.  The kernel therefore takes
// the window sums from an exact int64 prefix sum.  mn = t.mean() and std = t.std() are pandas nanops:
// NaN (the first w - 1 entries) replaced by 0, numpy sums, ddof = 1 -- the numpy-order summation
// above.  Output: mn / std / bot in the read's record and the two bit masks t < bot, t > bot.
// PT: int64_t, or uint32_t when every window sum fits 31 bits (w < 65 536: the prefix sums wrap, their differences do
// not) -- the kernel is bound by the traffic of its prefix sums (one write, six reads per sample), and this halves it.
template <typename PT>
__global__ __launch_bounds__(TPB)
void k_roll_stats(const int16_t *__restrict__ comp, int64_t stride, sk_prep *__restrict__ prep, int nreads,
                  int w, double std_scale, void *__restrict__ psum_,
                  uint64_t *__restrict__ below, uint64_t *__restrict__ above, int64_t mask_rows)
{
    PT *psum = (PT *)psum_;
    __shared__ Scratch sc_;
    Scratch *sc = &sc_;
    const int r = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int n = prep[r].n;
    const int16_t *crow = comp + (int64_t)r * stride;
    PT *P = psum + (int64_t)r * (stride + 1);                // P[i] = sum of the first i filtered samples
    if (tid == 0) { sc->tree_m = -1; P[0] = 0; }

    long long carry = 0;
    int parity = 0;
    for (int base = 0; base < n; base += TPB * 8, parity ^= 1) {
        const int i0 = base + tid * 8;
        int v[8], tsum = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { v[k] = (i0 + k < n) ? (int)crow[i0 + k] : 0; tsum += v[k]; }
        int total;
        const int excl = block_excl_scan(tsum, sc->wsum[parity], &total);     // < 2^31: 2048 * 32767
        long long run = carry + excl;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            run += v[k];
            if (i0 + k < n) P[i0 + k + 1] = (PT)run;
        }
        carry += total;
    }
    __syncthreads();                                         // the prefix sums are read by everybody

    const long long cnt = (n >= w) ? (long long)n - w + 1 : 0;               // entries of t that are not NaN
    const double dw = (double)w;
    auto tval = [&](int i) -> double {                        // i >= w - 1
        if (sizeof(PT) == 4) return (double)(int)(P[i + 1] - P[i + 1 - w]) / dw;
        return (double)(P[i + 1] - P[i + 1 - w]) / dw;
    };
    const double mn = numpy_sum(n, sc, [&](int i) { return (i >= w - 1) ? tval(i) : 0.0; }) / (double)cnt;
    const double ss = numpy_sum(n, sc, [&](int i) {
        if (i < w - 1) return 0.0;
        const double d = mn - tval(i);
        return d * d;
    });
    const double sd = sqrt(ss / (double)(cnt - 1));          // ddof = 1 (NaN for a single entry, like pandas)
    const double bot = mn - (sd * std_scale);                // dRNA_segmenter.py:288
    if (tid == 0) {
        sk_prep pr = prep[r];
        pr.center = mn; pr.scale = sd; pr.top = bot; pr.bot = bot;
        prep[r] = pr;
    }
    for (int base = 0; base < n; base += TPB) {
        const int i = base + tid;
        bool lt = false, gt = false;
        if (i < n && i >= w - 1) {
            const double t = tval(i);
            lt = t < bot;                                    // :297 / :300
            gt = t > bot;                                    // :302
        }
        const unsigned long long bl = __ballot(lt), ba = __ballot(gt);
        if (lane == 0 && i < n) {                            // (words that start behind the read have no row in the masks)
            below[(int64_t)(i >> 6) * mask_rows + r] = bl;
            above[(int64_t)(i >> 6) * mask_rows + r] = ba;
        }
    }
}

// ---- the same in ONE look (round 5) ----------------------------------------------------------------------------------
// k_prep_i16 + k_roll_stats move every sample through HBM four times over (compacted copy out and in, 4-byte prefix sums
// out and six times in).  Here a workgroup of 1024 lanes keeps the read's prefix sums in LDS -- 4 bytes a sample, reads of
// up to ~37 000 samples in the CU's 160 KB -- and HBM sees the raw samples once and the two bit masks:
//   1. filter + order-preserving compaction + prefix sum in one sweep: a lane takes 8 raw samples, the block scans
//      (count kept, sum kept), the lane writes P[pos + 1 ..] for its kept samples;
//   2. mn / std of the rolling mean in numpy's order: the full 8192-chunks are regular trees (64 leaves of 128, halves
//      all the way up), so all their leaves are summed side by side by 8-lane groups and one wavefront per chunk folds
//      its 64 leaf sums with six shifts; the last, ragged chunk goes through pairwise_chunk;
//   t = RN(S / w) itself costs three FP64 operations, not the division's thirty (see tval below);
//   3. the masks t < bot, t > bot without one: RN(S / w) is monotone in the integer window sum S, so thread 0
//      finds the two integer thresholds around bot * w (eight candidates each, judged with the real division) and the
//      mask sweep compares integers.
// P is laid out with one word of padding behind every 8 (roll_idx): the sweep's lanes write runs of up to 8 entries
// 8 apart, and the 8-lane groups of a wavefront read leaves 128 samples apart -- both would fall on the same 8 LDS banks.
constexpr int ROLL_NT = 1024;
constexpr int ROLL_NW = ROLL_NT / 64;
constexpr int ROLL_MAXFULL = 5;                              // full 8192-chunks of a read that fits in LDS
struct RollShared {
    int       wcnt[2][ROLL_NW];                              // per-wave kept counts / sums (double buffered by tile parity)
    int       wsum[2][ROLL_NW];
    double    leaf[ROLL_MAXFULL][64];
    double    chunk[ROLL_MAXFULL + 1];
    long long thr[2];                                        // S < thr[0] <=> t < bot;  S > thr[1] <=> t > bot
};
static_assert(sizeof(RollShared) % 16 == 0, "P behind it is 16-byte aligned");
__host__ __device__ __forceinline__ int roll_idx(int i) { return i + (i >> 3); }

template <typename Term>
__device__ double roll_numpy_sum(int n, Scratch *sc, RollShared *rs, Term term)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nfull = n / NPY_BUFSIZE, mlast = n % NPY_BUFSIZE;
    const int grp = tid >> 3, j = tid & 7;
    double res = 0.0;
    for (int c0 = 0; c0 < nfull; c0 += ROLL_MAXFULL) {       // (one round for a read that fits in LDS)
        const int nb = min(ROLL_MAXFULL, nfull - c0);
        for (int L = grp; L < nb * 64; L += ROLL_NT / 8) {
            const int s = (c0 * 64 + L) * PW_BLOCK;
            double r = term(s + j);
#pragma unroll 4
            for (int i = 8; i < PW_BLOCK; i += 8) r += term(s + i + j);
            r += dpp_shl_f64<1>(r);
            r += dpp_shl_f64<2>(r);
            r += dpp_shl_f64<4>(r);
            if (j == 0) rs->leaf[L >> 6][L & 63] = r;
        }
        lds_barrier();
        if (wv < nb) {                                       // one wavefront per full chunk: parent = left + right, six levels
            double v = rs->leaf[wv][lane];
            v += dpp_shl_f64<1>(v);
            v += dpp_shl_f64<2>(v);
            v += dpp_shl_f64<4>(v);
            v += dpp_shl_f64<8>(v);
            v += __shfl_down(v, 16);
            v += __shfl_down(v, 32);
            if (lane == 0) rs->chunk[wv] = v;
        }
        lds_barrier();
        for (int c = 0; c < nb; c++) res += rs->chunk[c];
        lds_barrier();                                       // (rs->leaf / chunk are free for the next round / sum)
    }
    if (mlast > 0 || nfull == 0)
        res += pairwise_chunk<true, ROLL_NT>(mlast, sc, [&](int i) { return term(nfull * NPY_BUFSIZE + i); });
    return res;
}

// GLOBALP: the prefix sums of a row too long for LDS live in a scratch row of global memory, one per workgroup (only the
// list of k_roll_stream's uncertifiable reads comes this way: a handful of reads in a hundred million)
template <bool GLOBALP>
__global__ __launch_bounds__(ROLL_NT)
void k_roll_one(const int16_t *__restrict__ sig, int64_t stride, const int32_t *__restrict__ len, int nreads,
                int lo, int hi, int w, double std_scale, int vec_ok, sk_prep *__restrict__ prep,
                uint64_t *__restrict__ below, uint64_t *__restrict__ above, int64_t mask_rows, int64_t read_stride,
                const int32_t *__restrict__ list, const int32_t *__restrict__ list_count,
                unsigned *__restrict__ gscratch, int64_t grow_words)
{
    extern __shared__ __align__(16) unsigned char roll_lds[];
    Scratch *sc = (Scratch *)roll_lds;
    RollShared *rs = (RollShared *)(roll_lds + sizeof(Scratch));
    unsigned *P = GLOBALP ? gscratch + (int64_t)blockIdx.x * grow_words
                          : (unsigned *)(roll_lds + sizeof(Scratch) + sizeof(RollShared));
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { sc->tree_m = -1; P[0] = 0u; }
    auto row_len = [&](int r) -> int { const int m = len[r]; return m < 0 ? 0 : (m > stride ? (int)stride : m); };

    // A persistent workgroup (the LDS leaves room for one or two per CU, so nothing else hides a read's first load or
    // the launch of the next workgroup): the first 16 bytes of the NEXT read are requested before this read's statistics.
    // (list: the reads k_roll_stream could not certify; otherwise every read of the batch)
    const int total = list ? min(*list_count, nreads) : nreads;
    auto read_of = [&](int k) -> int { return list ? list[k] : k; };
    int k = blockIdx.x;
    int r = k < total ? read_of(k) : 0;
    int M = k < total ? row_len(r) : 0;
    uint4 qn = make_uint4(0u, 0u, 0u, 0u);
    if (k < total && vec_ok && tid * 8 + 8 <= M) qn = *(const uint4 *)(sig + (int64_t)r * stride + tid * 8);
    for (; k < total; k += gridDim.x) {
        const int16_t *row = sig + (int64_t)r * stride;

        // ---- 1. filter, compaction, prefix sums (wrapping uint32: w < 65 536, their differences do not wrap)
        int ccarry = 0;
        unsigned scarry = 0u;
        int parity = 0;
        for (int base = 0; base < M; base += ROLL_NT * 8, parity ^= 1) {
            const int i0 = base + tid * 8;
            int v[8];
            const uint4 q = qn;
            if (vec_ok && i0 + ROLL_NT * 8 + 8 <= M) qn = *(const uint4 *)(row + i0 + ROLL_NT * 8);    // the next tile
            if (vec_ok && i0 + 8 <= M) {
                const unsigned qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int k = 0; k < 4; k++) { v[2 * k] = (int)(short)(qq[k] & 0xffffu); v[2 * k + 1] = (int)(short)(qq[k] >> 16); }
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = (i0 + k < M) ? (int)row[i0 + k] : lo;     // (lo is not kept)
            }
            int c = 0, s = 0;
            unsigned keep = 0u;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const bool kept = v[k] > lo && v[k] < hi;                                   // scale_outliers: strictly inside
                c += kept ? 1 : 0;
                s += kept ? v[k] : 0;
                keep |= (kept ? 1u : 0u) << k;
            }
            const int inc_c = wave_incl_scan(c, lane), inc_s = wave_incl_scan(s, lane);
            if (lane == 63) { rs->wcnt[parity][wv] = inc_c; rs->wsum[parity][wv] = inc_s; }
            lds_barrier();
            int bc = 0, bs = 0, tc = 0, ts = 0;
#pragma unroll
            for (int i = 0; i < ROLL_NW; i++) {
                const int cc = rs->wcnt[parity][i], ss = rs->wsum[parity][i];
                if (i < wv) { bc += cc; bs += ss; }
                tc += cc; ts += ss;
            }
            int pos = ccarry + bc + inc_c - c;
            unsigned run = scarry + (unsigned)(bs + inc_s - s);
#pragma unroll
            for (int k = 0; k < 8; k++)
                if ((keep >> k) & 1u) { run += (unsigned)v[k]; pos++; P[roll_idx(pos)] = run; }
            ccarry += tc;
            scarry += (unsigned)ts;
        }
        const int n = ccarry;
        const int kn = k + gridDim.x;
        int rn = 0, Mn = 0;
        if (kn < total) {
            rn = read_of(kn);
            Mn = row_len(rn);
            if (vec_ok && tid * 8 + 8 <= Mn) qn = *(const uint4 *)(sig + (int64_t)rn * stride + tid * 8);
        }
        if (GLOBALP) __syncthreads();                        // (the prefix sums are global stores of other wavefronts)
        else lds_barrier();

        // ---- 2. numpy-order mean and std of the rolling mean (the expressions of k_roll_stats)
        const long long cnt = (n >= w) ? (long long)n - w + 1 : 0;
        const double dw = (double)w;
        auto wsum = [&](int i) -> int { return (int)(P[roll_idx(i + 1)] - P[roll_idx(i + 1 - w)]); };    // i >= w - 1
        // t = RN(S / w) in three operations: the quotient estimate, its exact remainder, one correction -- correctly
        // rounded for |S| < 2^31, w < 2^16 (tools/ubench/check_intdiv.c: the argument, and 1.3 G quotients against the division)
        const double inv_w = 1.0 / dw;
        auto tval = [&](int i) -> double {
            const double s = (double)wsum(i);
            const double q = s * inv_w;
            return fma(fma(-q, dw, s), inv_w, q);
        };
        const double mn = roll_numpy_sum(n, sc, rs, [&](int i) { return (i >= w - 1) ? tval(i) : 0.0; }) / (double)cnt;
        const double ss = roll_numpy_sum(n, sc, rs, [&](int i) {
            if (i < w - 1) return 0.0;
            const double d = mn - tval(i);
            return d * d;
        });
        const double sd = sqrt(ss / (double)(cnt - 1));
        const double bot = mn - (sd * std_scale);
        if (tid == 0) {
            sk_prep pr;
            pr.n = n; pr.flags = n > 0 ? 0 : SK_FLAG_EMPTY;
            pr.center = mn; pr.scale = sd; pr.top = bot; pr.bot = bot;
            prep[r] = pr;
        }
        // ---- 3. integer thresholds of the two comparisons: eight candidates around bot * w, one lane each
        if (wv == 0) {
            long long lt, gt;
            const double x = bot * dw;
            if (bot != bot) { lt = LLONG_MIN; gt = LLONG_MAX; }                 // NaN: neither t < bot nor t > bot
            else if (x >= 2147483652.0) { lt = LLONG_MAX; gt = LLONG_MAX; }     // every window sum is below
            else if (x <= -2147483653.0) { lt = LLONG_MIN; gt = LLONG_MIN; }    // every window sum is above
            else {
                const long long s0 = (long long)floor(x);
                const double t = (double)(s0 - 3 + (lane & 7)) / dw;             // (the real division: these decide)
                const int nlt = __popcll(__ballot(lane < 8 && t < bot)), ngt = __popcll(__ballot(lane < 8 && t > bot));
                lt = s0 - 3 + nlt;                                               // smallest S with t(S) >= bot (t is monotone in S)
                gt = s0 + 4 - ngt;                                               // largest S with t(S) <= bot
            }
            if (lane == 0) { rs->thr[0] = lt; rs->thr[1] = gt; }
        }
        lds_barrier();
        const long long thr_lt = rs->thr[0], thr_gt = rs->thr[1];
        for (int base = 0; base < n; base += ROLL_NT) {
            const int i = base + tid;
            bool lt = false, gt = false;
            if (i < n && i >= w - 1) {
                const long long S = (long long)wsum(i);
                lt = S < thr_lt;                             // :297 / :300
                gt = S > thr_gt;                             // :302
            }
            const unsigned long long bl = __ballot(lt), ba = __ballot(gt);
            if (lane == 0 && i < n) {
                below[(int64_t)(i >> 6) * mask_rows + (int64_t)r * read_stride] = bl;     // (word stride, read stride)
                above[(int64_t)(i >> 6) * mask_rows + (int64_t)r * read_stride] = ba;
            }
        }
        M = Mn; r = rn;
        if (GLOBALP) __syncthreads();
        else lds_barrier();                                  // (the next read writes P and the thresholds)
    }
}

// ---- ... and as a stream (round 5): a wavefront per read, certified thresholds, no numpy-order sum -------------------
// The branch uses mn and std of the rolling mean for ONE thing: bot = mn - std * std_scale, compared with every t
// (`t < bot`, `t > bot`, dRNA_segmenter.py:297-302).  t = RN(S / w) is monotone in the integer window sum S, so the
// comparisons are two integer thresholds on S (k_roll_one, step 3) -- and those only move when bot crosses one of the
// values k / w, which are 1 / w apart, while everything numpy's order of summation can do to bot is 10^-11.  So, as the
// segmenter does for its statistics (sk_segstat.hip): bot from EXACT integer sums (sum S, sum S^2: one streaming sweep
// with a few registers of state), a bound `delta` on how far the reference's floating-point value can be from it
// (roll_bot_delta: depth of numpy's summation tree times the unit roundoff times the magnitudes involved, doubled), the
// thresholds at bot - delta and at bot + delta; if they agree the masks are certified, if not (bot * w within ~10^-7 of
// an integer: about one read in ten million) the read goes on a list for k_roll_one, which computes in numpy's order.
// A sweep rebuilds the prefix sums tile by tile (512 raw samples, a lane takes 8, one wave scan, no barrier) into a ring
// of w + 576 entries in LDS (12 KB at w = 2 000: thirteen reads in flight per CU) and consumes the window sums 64
// outputs at a time.  Sweep 1: n, sum S, sum S^2.  Sweep 2 (the read comes from L2 the second time): the masks.
struct RollStreamArgs {
    const int16_t *sig; int64_t stride; const int32_t *len; int nreads;
    int lo, hi, w, vec_ok, ring;
    double std_scale, amax, delta_scale;
    sk_prep *prep; uint64_t *below, *above; int64_t mask_rows, read_stride;     // word stride, read stride of the masks
    int32_t *redo;                                           // [0] count, [2 ..] reads for k_roll_one
};

// How far the reference's bot (pandas nanops on the float64 rolling mean: numpy pairwise sums, ddof = 1) can lie from
// the one computed from exact sums.  u = 2^-53; D bounds the depth of any of numpy's summation trees over n terms
// (16 serial adds per accumulator, 3 to merge the eight, <= 7 levels, one add per 8192-chunk); A >= |t|.
//   mn:  |fl(sum t) / cnt - mean| <= (D + 3) u A              (t itself is rounded: + u A, the division: + u A)
//   std: the squares are summed about the ROUNDED mean: sqrt(var + cnt/(cnt-1) em^2) - sd <= 1.5 em; rounded t: 1.5 u A;
//        differences, squares, sum, division, root: (D + 8) u relative
//   bot: two more roundings.  All of it doubled.
__device__ __forceinline__ double roll_bot_delta(int n, double amax, double mn, double sd, double sc)
{
    const double u = 1.1102230246251565e-16, D = 40.0 + (double)(n >> 13), asc = fabs(sc);
    const double em = (D + 3.0) * u * amax;
    return 2.0 * (em * (1.0 + 1.5 * asc) + asc * ((D + 8.0) * u * sd + 1.5 * u * amax) + 12.0 * u * (fabs(mn) + 2.0 * asc * sd));
}

__global__ __launch_bounds__(64)
void k_roll_stream(const RollStreamArgs a)
{
    extern __shared__ unsigned roll_ring[];                  // ring of prefix sums
    const int lane = threadIdx.x;
    const int r = blockIdx.x;
    const int RS = a.ring, w = a.w;
    int M = a.len[r];
    M = M < 0 ? 0 : (M > a.stride ? (int)a.stride : M);
    const int16_t *row = a.sig + (int64_t)r * a.stride;
    auto wrap = [&](unsigned m) -> unsigned { return min(m, m - (unsigned)RS); };        // m < 2 RS
    // lo < x < hi for two packed samples at once: x == clamp(x, lo + 1, hi - 1) (limits outside int16 saturate, which changes nothing
    // for an int16 x; an empty range switches the packed test off)
    const int lo1 = max(a.lo + 1, -32768), hi1 = min(a.hi - 1, 32767);
    const bool pk_ok = lo1 <= hi1;
    const unsigned pk_lo = ((unsigned)lo1 & 0xffffu) * 0x10001u, pk_hi = ((unsigned)hi1 & 0xffffu) * 0x10001u;

    // one sweep over the raw samples; consume(done, slot of P[done + 1], g) is called for g <= 8 whole groups of 64 outputs
    // from `done` on, consume_tail(done, slot, avail) for the last avail < 64
    auto sweep = [&](auto &&consume, auto &&consume_tail) -> int {
        int ncar = 0, done = 0;
        unsigned scar = 0u;
        unsigned wslot = 1u;                                 // slot of P[ncar + 1] (uniform; P[0] = 0 sits in slot 0)
        unsigned dslot = 1u;                                 // slot of P[done + 1]
        if (lane == 0) roll_ring[0] = 0u;
        // four tiles (2 048 samples, 32 bytes a lane in flight beyond the ones being scanned): with one tile ahead the chip
        // has 3 MB outstanding, and 8 TB/s times the memory's latency is five times that
        uint4 qa[4], qb[4];
        auto fetch = [&](int b, uint4 (&q)[4]) {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int i0 = b + t * 512 + lane * 8;
                q[t] = make_uint4(0u, 0u, 0u, 0u);
                if (a.vec_ok && i0 + 8 <= M) q[t] = *(const uint4 *)(row + i0);
            }
        };
        auto tile = [&](int base, const uint4 q) {
            const int i0 = base + lane * 8;
            const bool vec = a.vec_ok && i0 + 8 <= M;
            const unsigned qq[4] = {q.x, q.y, q.z, q.w};
            // Nothing dropped in this tile -- the usual case, found with four packed clamps per lane: the counts are known,
            // the lane's sum is the last of its local prefix sums, and its 8 entries go out in a straight line with
            // immediate offsets unless the ring's end falls among them.
            bool lane_all = vec && pk_ok;
#pragma unroll
            for (int k = 0; k < 4; k++) lane_all = lane_all && clamp_pk_i16(qq[k], pk_lo, pk_hi) == qq[k];
            int tc, ts;
            if (__ballot(!lane_all) == 0ull) {
                unsigned pre[8];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    pre[2 * k] = (unsigned)(int)(short)(qq[k] & 0xffffu);
                    pre[2 * k + 1] = (unsigned)((int)qq[k] >> 16);
                }
#pragma unroll
                for (int k = 1; k < 8; k++) pre[k] += pre[k - 1];
                const int s = (int)pre[7];
                const int inc_s = wave_incl_scan(s, lane);
                ts = __builtin_amdgcn_readlane(inc_s, 63);
                tc = 512;
                const unsigned run0 = scar + (unsigned)(inc_s - s);
                unsigned slot = wrap(wslot + 8u * (unsigned)lane);
                if (__ballot(slot + 7u >= (unsigned)RS) == 0ull) {
                    unsigned *p = roll_ring + slot;
#pragma unroll
                    for (int k = 0; k < 8; k++) p[k] = run0 + pre[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 8; k++) { roll_ring[slot] = run0 + pre[k]; slot = wrap(slot + 1u); }
                }
            } else {
                int v[8];
                if (vec) {
#pragma unroll
                    for (int k = 0; k < 4; k++) { v[2 * k] = (int)(short)(qq[k] & 0xffffu); v[2 * k + 1] = (int)(short)(qq[k] >> 16); }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; k++) v[k] = (i0 + k < M) ? (int)row[i0 + k] : a.lo;      // (lo is not kept)
                }
                int c = 0, s = 0;
                unsigned keep = 0u;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const bool kept = v[k] > a.lo && v[k] < a.hi;                                   // scale_outliers: strictly inside
                    c += kept ? 1 : 0;
                    s += kept ? v[k] : 0;
                    keep |= (kept ? 1u : 0u) << k;
                }
                const int inc_s = wave_incl_scan(s, lane), inc_c = wave_incl_scan(c, lane);
                ts = __builtin_amdgcn_readlane(inc_s, 63);
                tc = __builtin_amdgcn_readlane(inc_c, 63);
                unsigned slot = wrap(wslot + (unsigned)(inc_c - c));       // (inc_c - c <= 504 < RS)
                unsigned run = scar + (unsigned)(inc_s - s);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if ((keep >> k) & 1u) {
                        run += (unsigned)v[k];
                        roll_ring[slot] = run;
                        slot = wrap(slot + 1u);
                    }
            }
            ncar += tc;
            scar += (unsigned)ts;
            wslot = wrap(wslot + (unsigned)tc);
            // whole groups of 64 outputs that are ready: up to 8, their LDS reads issued together
            const int g = (ncar - done) >> 6;
            if (g > 0) {
                consume(done, dslot, g);
                done += 64 * g;
                dslot = wrap(dslot + 64u * (unsigned)g);
            }
        };
        fetch(0, qa);
        for (int base = 0; base < M; base += 2048) {
            fetch(base + 2048, qb);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (base + t * 512 < M) tile(base + t * 512, qa[t]);      // (uniform)
                qa[t] = qb[t];
            }
        }
        if (done < ncar) consume_tail(done, dslot, ncar - done);
        return ncar;
    };
    // window sum of output i = done + lane: P[i + 1] - P[i + 1 - w]   (i < w - 1: the caller does not use it)
    const unsigned back = (unsigned)(RS - w);                // slot(i + 1 - w) = slot(i + 1) + RS - w (mod RS); w < RS
    auto wsum = [&](unsigned dslot) -> int {                 // dslot < 2 RS
        const unsigned sa = wrap(wrap(dslot) + (unsigned)lane);
        const unsigned sb = wrap(sa + back);
        return (int)(roll_ring[sa] - roll_ring[sb]);
    };
    // the window sums of up to 8 groups, all loads before any use (groups beyond g read stale entries: not used); when
    // neither stretch of the ring wraps, two addresses and sixteen loads with immediate offsets
    auto wsums = [&](unsigned dslot, int (&S)[8]) {
        const unsigned db = wrap(dslot + back);              // (uniform)
        if (dslot + 575u < (unsigned)RS && db + 575u < (unsigned)RS) {
            const unsigned *pa = roll_ring + dslot + lane, *pb = roll_ring + db + lane;
#pragma unroll
            for (int k = 0; k < 8; k++) S[k] = (int)(pa[64 * k] - pb[64 * k]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) S[k] = wsum(dslot + 64u * (unsigned)k);
        }
    };

    // ---- sweep 1: exact sums
    long long accS = 0;
    unsigned long long accQ = 0ull;
    auto add1 = [&](int S) { accS += (long long)S; accQ += (unsigned long long)((long long)S * (long long)S); };
    const int n = sweep([&](int done, unsigned dslot, int g) {
        int S[8];
        wsums(dslot, S);
        if (done >= w - 1) {                                 // (uniform: past the first w - 1 outputs every lane counts)
#pragma unroll
            for (int k = 0; k < 8; k++) add1(k < g ? S[k] : 0);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) add1((k < g && done + 64 * k + lane >= w - 1) ? S[k] : 0);
        }
    }, [&](int done, unsigned dslot, int avail) {
        const int S = wsum(dslot);
        add1((lane < avail && done + lane >= w - 1) ? S : 0);
    });
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { accS += __shfl_xor(accS, d); accQ += __shfl_xor(accQ, d); }

    // ---- bot from the exact sums, its uncertainty, the thresholds at both ends
    const long long cnt = (n >= w) ? (long long)n - w + 1 : 0;
    const double dw = (double)w;
    long long thr_lt = LLONG_MIN, thr_gt = LLONG_MAX;        // NaN bot: no bit in either mask
    double mn = __builtin_nan(""), sd = __builtin_nan(""), bot = __builtin_nan("");
    bool certified = true;
    if (cnt >= 1) mn = (double)accS / ((double)cnt * dw);
    if (cnt >= 2 && a.std_scale == a.std_scale) {
        // cnt sum S^2 - (sum S)^2 in 128 bits (exact, >= 0), then one conversion
        const unsigned long long ucnt = (unsigned long long)cnt, as = (unsigned long long)(accS < 0 ? -accS : accS);
        const unsigned long long hi1 = __umul64hi(ucnt, accQ), lo1 = ucnt * accQ;
        const unsigned long long hi2 = __umul64hi(as, as), lo2 = as * as;
        const unsigned long long lo = lo1 - lo2, hi = hi1 - hi2 - (lo1 < lo2 ? 1ull : 0ull);
        const double num = (double)hi * 18446744073709551616.0 + (double)lo;
        sd = sqrt(num / ((double)cnt * (double)(cnt - 1) * dw * dw));
        bot = mn - sd * a.std_scale;
        const double delta = a.delta_scale * roll_bot_delta(n, a.amax, mn, sd, a.std_scale);
        const double b0 = bot - delta, b1 = bot + delta;
        const double x0 = b0 * dw, x1 = b1 * dw;
        if (!(delta == delta) || !(x0 > -4.0e18) || !(x1 < 4.0e18)) certified = false;
        else {
            // lanes 0-7 judge the eight integers around b0 w against b0, lanes 8-15 those around b1 w against b1
            const double bb = (lane & 8) ? b1 : b0;
            const long long s0 = (long long)floor((lane & 8) ? x1 : x0);
            const double t = (double)(s0 - 3 + (lane & 7)) / dw;
            const unsigned long long mlt = __ballot(lane < 16 && t < bb), mgt = __ballot(lane < 16 && t > bb);
            const long long s00 = __shfl(s0, 0), s01 = __shfl(s0, 8);
            const long long lt0 = s00 - 3 + __popcll(mlt & 0xffull), lt1 = s01 - 3 + __popcll(mlt & 0xff00ull);
            const long long gt0 = s00 + 4 - __popcll(mgt & 0xffull), gt1 = s01 + 4 - __popcll(mgt & 0xff00ull);
            certified = (lt0 == lt1) && (gt0 == gt1);
            thr_lt = lt0; thr_gt = gt0;
        }
    }
    if (lane == 0) {
        sk_prep pr;
        pr.n = n; pr.flags = n > 0 ? 0 : SK_FLAG_EMPTY;
        pr.center = mn; pr.scale = sd; pr.top = bot; pr.bot = bot;   // (diagnostic values: within delta of the reference's)
        a.prep[r] = pr;
        if (!certified) a.redo[2 + atomicAdd(a.redo, 1)] = r;
    }
    if (!certified) return;                                  // k_roll_one writes this read's masks

    // ---- sweep 2: the masks
    unsigned long long *brow = (unsigned long long *)a.below + (int64_t)r * a.read_stride;
    unsigned long long *arow = (unsigned long long *)a.above + (int64_t)r * a.read_stride;
    // (|S| < 2^31 - 600 by the launch's check, so the two comparisons fit 32 bits: S < thr_lt <=> S <= le, S > thr_gt <=> S >= ge
    // with the thresholds saturated -- INT_MIN as `le` is "never", INT_MAX as `ge` is "never")
    const long long le64 = thr_lt == LLONG_MIN ? (long long)INT_MIN : thr_lt - 1, ge64 = thr_gt == LLONG_MAX ? (long long)INT_MAX : thr_gt + 1;
    const int le = (int)(le64 < INT_MIN ? INT_MIN : le64 > INT_MAX ? INT_MAX : le64);
    const int ge = (int)(ge64 < INT_MIN ? INT_MIN : ge64 > INT_MAX ? INT_MAX : ge64);
    (void)sweep([&](int done, unsigned dslot, int g) {
        int S[8];
        wsums(dslot, S);
        unsigned long long myb = 0ull, mya = 0ull;           // lane k keeps word k of this batch
        if (done >= w - 1) {                                 // (uniform: past the first w - 1 outputs every lane counts)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const unsigned long long bl = __ballot(S[k] <= le), ba = __ballot(S[k] >= ge);
                if (lane == k) { myb = bl; mya = ba; }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const bool valid = done + 64 * k + lane >= w - 1;
                const unsigned long long bl = __ballot(valid && S[k] <= le), ba = __ballot(valid && S[k] >= ge);
                if (lane == k) { myb = bl; mya = ba; }
            }
        }
        if (lane < g) {
            brow[(int64_t)((done >> 6) + lane) * a.mask_rows] = myb;
            arow[(int64_t)((done >> 6) + lane) * a.mask_rows] = mya;
        }
    }, [&](int done, unsigned dslot, int avail) {
        const int S = wsum(dslot);
        const bool valid = lane < avail && done + lane >= w - 1;
        const unsigned long long bl = __ballot(valid && S <= le), ba = __ballot(valid && S >= ge);
        if (lane == 0) {
            brow[(int64_t)(done >> 6) * a.mask_rows] = bl;
            arow[(int64_t)(done >> 6) * a.mask_rows] = ba;
        }
    });
}

} // namespace

// LDS of the one-look rolling-mean kernel for rows of `stride` samples (0: does not fit)
size_t sk_roll_one_lds(int64_t stride, int32_t w)
{
    if (w >= 65536 || stride > (1 << 17)) return 0;
    const size_t need = sizeof(Scratch) + sizeof(RollShared) + ((size_t)roll_idx((int)stride + 1) + 4) * sizeof(unsigned);
    return need <= 160 * 1024 ? need : 0;
}

int sk_launch_roll_one(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads, int32_t lo,
                       int32_t hi, int32_t w, double std_scale, sk_prep *d_prep, uint64_t *d_below, uint64_t *d_above)
{
    if (nreads <= 0) return SK_OK;
    const size_t lds = sk_roll_one_lds(stride, w);
    if (!lds) return sk_fail(SK_ERR_INVALID, "internal: the one-look rolling-mean kernel does not hold this read length");
    if (lds > 64 * 1024)
        SK_HIP(hipFuncSetAttribute((const void *)k_roll_one<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int vec_ok = (((uintptr_t)d_sig & 15) == 0 && (stride % 8) == 0) ? 1 : 0;
    int per_cu = (int)((160 * 1024) / (lds + 256));
    if (per_cu > 2) per_cu = 2;
    if (const char *e = sk_tune("SK_PREP_PERCU")) { int v = atoi(e); if (v > 0 && v < per_cu) per_cu = v; }
    const long long g = (long long)c->num_cu * per_cu;
    const int grid = g > nreads ? nreads : (int)g;
    hipLaunchKernelGGL(k_roll_one<false>, dim3(grid), dim3(ROLL_NT), lds, c->stream, d_sig, stride, d_len, nreads, lo, hi, w,
                       std_scale, vec_ok, d_prep, d_below, d_above, (int64_t)nreads, (int64_t)1, (const int32_t *)nullptr,
                       (const int32_t *)nullptr, (unsigned *)nullptr, (int64_t)0);
    SK_HIP(hipGetLastError());
    return SK_OK;
}

// can the streaming kernel take this call?  (the ring fits a wavefront's share of LDS; the sums of window sums and of
// their squares stay inside 64 bits; the reads it cannot certify fit k_roll_one)
bool sk_roll_stream_ok(int64_t stride, int32_t w, int32_t lo, int32_t hi)
{
    if (w > 12000 || stride > (1 << 21)) return false;
    const double amax = fmax(fabs((double)lo), fabs((double)hi));
    const double smax = amax * (double)w;
    return smax < 2147483000.0 && smax * smax * (double)stride < 9.0e18;
}

int sk_launch_roll_stream(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads, int32_t lo,
                          int32_t hi, int32_t w, double std_scale, sk_prep *d_prep, uint64_t *d_below, uint64_t *d_above,
                          int32_t *d_redo)
{
    if (nreads <= 0) return SK_OK;
    (void)d_above;                                           // (the two masks are interleaved in d_below's buffer, see below)
    RollStreamArgs a;
    a.sig = d_sig; a.stride = stride; a.len = d_len; a.nreads = nreads; a.lo = lo; a.hi = hi; a.w = w;
    a.vec_ok = (((uintptr_t)d_sig & 15) == 0 && (stride % 8) == 0) ? 1 : 0;
    a.ring = (w + 576 + 7) & ~7;
    a.std_scale = std_scale; a.amax = fmax(fabs((double)lo), fabs((double)hi));
    a.delta_scale = 1.0;
    if (const char *e = sk_tune("SK_ROLL_DELTA_SCALE")) { const double v = atof(e); if (v > 0.0) a.delta_scale = v; }
    // masks read-major here, the two interleaved ([read][word]{below, above}: a wavefront writes its read's words one
    // after the other, and the walk's two loads of a step fall into one line); word-major, as the other kernels lay them
    // out, every 8-byte word a wavefront writes is a line of its own
    a.prep = d_prep; a.below = d_below; a.above = d_below + 1; a.mask_rows = 2; a.read_stride = 2 * ((stride + 63) / 64);
    a.redo = d_redo;
    const size_t ring_lds = ((size_t)a.ring + 8) * sizeof(unsigned);
    SK_HIP(hipMemsetAsync(d_redo, 0, 2 * sizeof(int32_t), c->stream));
    if (ring_lds > 64 * 1024)
        SK_HIP(hipFuncSetAttribute((const void *)k_roll_stream, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring_lds));
    hipLaunchKernelGGL(k_roll_stream, dim3(nreads), dim3(64), ring_lds, c->stream, a);
    SK_HIP(hipGetLastError());
    // the reads it could not certify (normally none): numpy's order, k_roll_one over the list -- prefix sums in LDS when
    // the rows fit there, in a scratch row per workgroup otherwise
    const size_t lds = sk_roll_one_lds(stride, w);
    const bool all_listed = a.delta_scale > 1e6;             // (tests: everything is redone -- give the list more of the chip)
    int grid = all_listed ? (c->num_cu < nreads ? c->num_cu : nreads) : (nreads < 32 ? nreads : 32);
    if (lds) {
        if (lds > 64 * 1024)
            SK_HIP(hipFuncSetAttribute((const void *)k_roll_one<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_roll_one<false>, dim3(grid), dim3(ROLL_NT), lds, c->stream, d_sig, stride, d_len, nreads, lo, hi, w,
                           std_scale, a.vec_ok, d_prep, a.below, a.above, a.mask_rows, a.read_stride,
                           (const int32_t *)(d_redo + 2), (const int32_t *)d_redo, (unsigned *)nullptr, (int64_t)0);
    } else {
        const int64_t grow = ((int64_t)roll_idx((int)stride + 1) + 8) & ~(int64_t)3;
        while (grid > 1 && (size_t)grid * (size_t)grow * sizeof(unsigned) > ((size_t)1 << 30)) grid /= 2;
        int rc = sk_reserve(c, &c->comp, (size_t)grid * (size_t)grow * sizeof(unsigned));
        if (rc) return rc;
        hipLaunchKernelGGL(k_roll_one<true>, dim3(grid), dim3(ROLL_NT), sizeof(Scratch) + sizeof(RollShared) + 16, c->stream,
                           d_sig, stride, d_len, nreads, lo, hi, w, std_scale, a.vec_ok, d_prep, a.below, a.above,
                           a.mask_rows, a.read_stride, (const int32_t *)(d_redo + 2), (const int32_t *)d_redo,
                           (unsigned *)c->comp.p, grow);
    }
    SK_HIP(hipGetLastError());
    return SK_OK;
}

int sk_launch_roll_stats(sk_ctx *c, const int16_t *d_comp, int64_t stride, sk_prep *d_prep, int32_t nreads,
                         int32_t w, double std_scale, int64_t *d_psum, uint64_t *d_below, uint64_t *d_above)
{
    if (nreads <= 0) return SK_OK;
    if (w < 65536 && sk_tune("SK_DRNA_STEP") == nullptr)
        hipLaunchKernelGGL(k_roll_stats<uint32_t>, dim3(nreads), dim3(TPB), 0, c->stream, d_comp, stride, d_prep, nreads, w,
                           std_scale, (void *)d_psum, d_below, d_above, (int64_t)nreads);
    else
        hipLaunchKernelGGL(k_roll_stats<int64_t>, dim3(nreads), dim3(TPB), 0, c->stream, d_comp, stride, d_prep, nreads, w,
                           std_scale, (void *)d_psum, d_below, d_above, (int64_t)nreads);
    SK_HIP(hipGetLastError());
    return SK_OK;
}

int sk_launch_prep_i16(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len,
                       int32_t nreads, int32_t lo, int32_t hi, int mode, double std_scale,
                       int16_t *d_comp, sk_prep *d_prep, uint64_t *d_mask, int64_t mask_stride,
                       int32_t t0, int32_t t1, const int32_t *d_list, const int32_t *d_count, void *d_mask2, int row16)
{
    if (nreads <= 0) return SK_OK;
    const bool listed = d_list != nullptr;
    ListedArgs la;
    la.list = d_list; la.count = d_count; la.mask2 = (unsigned char *)d_mask2; la.row16 = row16;
    if (mode == SK_PREP_MEDMAD && t0 <= 0 && t1 == 0x7fffffff) {      // medmad: one wavefront per read
        const int rc = sk_launch_prepw_medmad(c, d_sig, stride, d_len, nreads, lo, hi, d_comp, d_prep);
        if (rc != 1) return rc;
    }
    const int64_t nbins = (int64_t)hi - (int64_t)lo - 1 > 0 ? (int64_t)hi - lo - 1 : 0;
    const int64_t nb4 = (nbins + 3) & ~(int64_t)3;
    if (mode == SK_PREP_DRNA && !listed && d_mask != nullptr && nbins >= 1 && nbins <= 2048 && t0 >= 0 && t1 > t0 &&
        t1 <= 16384 && stride < (1 << 19) && sk_tune("SK_DRNA_STEP") == nullptr && sk_tune("SK_PREP_BLOCK") == nullptr) {
        // one look at the read: the statistics window collected in LDS, every later tile classified from registers
        const int lbits_words = (int)(((stride + 63) / 64) * 2 + 2);
        const size_t dl = sizeof(Scratch) + (size_t)nb4 * 4 + (size_t)lbits_words * 4 + (size_t)(t1 + 8 * TPB + 8) * sizeof(int16_t);
        if (dl <= 60 * 1024) {
            const int vec_ok1 = (((uintptr_t)d_sig & 15) == 0 && (stride % 8) == 0) ? 1 : 0;
            int per_cu = (int)((160 * 1024) / (dl + 512));
            if (per_cu > 5) per_cu = 5;
            long long g = (long long)c->num_cu * per_cu * 4;
            const int grid = g > nreads ? nreads : (int)g;
            hipLaunchKernelGGL(k_drna_stats, dim3(grid), dim3(TPB), dl, c->stream, d_sig, stride, d_len, nreads, lo, hi,
                               std_scale, vec_ok1, t0, t1, lbits_words, d_prep, d_mask, mask_stride);
            SK_HIP(hipGetLastError());
            return SK_OK;
        }
    }
    size_t lds = sizeof(Scratch) + (size_t)nb4 * 4;
    if (lds > 160 * 1024)
        return sk_fail(SK_ERR_UNSUPPORTED,
                       "outlier limits (%d, %d) span %lld integer values: the LDS histogram holds %d (%s)",
                       lo, hi, (long long)nbins, 38900,
                       "narrow -scale_low/-scale_hi / -lim_low/-lim_hi");
    const int vec_ok = ((((uintptr_t)d_sig & 15) == 0 && (stride % 8) == 0) ? 1 : 0) |
                       ((((uintptr_t)d_comp & 15) == 0 && (stride % 8) == 0) ? 2 : 0);
    // keep the compacted samples in LDS when the whole read fits next to the histograms without
    // dropping below ~4 workgroups per CU (mean/std modes only; medmad never re-reads samples)
    const size_t lds_comp = (size_t)stride * sizeof(int16_t);
    const bool ldscomp = mode != SK_PREP_MEDMAD && lds + lds_comp <= 40 * 1024;
    if (ldscomp) lds += lds_comp;
    const bool windowed = t0 > 0 || t1 < 0x7fffffff;
    if (listed && (windowed || mode != SK_PREP_SEGMENT || (!ldscomp && d_comp == nullptr)))
        return sk_fail(SK_ERR_INVALID, "internal: listed prep is the segmenter variant (scratch rows for long reads)");
    auto fn = listed ? (ldscomp ? k_prep_i16<true, false, false, true> : k_prep_i16<false, false, false, true>)
              : (mode == SK_PREP_MEDMAD) ? k_prep_i16<false, false, true>
              : ldscomp ? (windowed ? k_prep_i16<true, true, false> : k_prep_i16<true, false, false>)
                        : (windowed ? k_prep_i16<false, true, false> : k_prep_i16<false, false, false>);
    if (lds > 64 * 1024)
        SK_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // persistent grid: as many workgroups as the chip holds (8 x 256 threads per CU, LDS permitting)
    int per_cu = (int)((160 * 1024) / (lds + 512));
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    // several grid "rounds" so that the resident count (fewer than per_cu when SGPRs bind) does
    // not have to divide the grid; SK_PREP_ROUNDS is a tuning override
    int rounds = 6;
    if (const char *e = sk_tune("SK_PREP_ROUNDS")) { int v = atoi(e); if (v > 0) rounds = v; }
    if (const char *e = sk_tune("SK_PREP_PERCU")) { int v = atoi(e); if (v > 0 && v < per_cu) per_cu = v; }
    long long g = (long long)c->num_cu * per_cu * rounds;
    int grid = g > nreads ? nreads : (int)g;
    if (listed && grid > c->num_cu) grid = c->num_cu;      // (the list is almost always empty)
    hipLaunchKernelGGL(fn, dim3(grid), dim3(TPB), lds, c->stream, d_sig, stride, d_len, nreads,
                       lo, hi, mode, std_scale, vec_ok, t0, t1, d_comp, d_prep, d_mask, mask_stride, la);
    SK_HIP(hipGetLastError());
    return SK_OK;
}

int sk_launch_prep_f64(sk_ctx *c, const double *d_sig, const int64_t *d_off, int32_t nreads,
                       double lo, double hi, int mode, double std_scale,
                       double *d_comp, sk_prep *d_prep, uint64_t *d_mask, int64_t mask_rows, const int32_t *d_rlen)
{
    if (nreads <= 0) return SK_OK;
    ListedF64 la;
    la.list = nullptr; la.count = nullptr; la.mask2 = nullptr; la.row16 = 0; la.scratch_stride = 0; la.len = d_rlen;
    hipLaunchKernelGGL(k_prep_f64<false>, dim3(nreads), dim3(TPB), 0, c->stream, d_sig, d_off, nreads, lo, hi, mode,
                       std_scale, d_comp, d_prep, d_mask, mask_rows, la);
    SK_HIP(hipGetLastError());
    return SK_OK;
}

// numpy-order redo of listed raw reads through the pA conversion (see k_prep_pa_listed): d_scratch holds `grid` rows of
// scratch_stride doubles
int sk_launch_prep_pa_listed(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len, const double *d_cal,
                             const int32_t *d_list, const int32_t *d_count, int grid, double lo, double hi, double std_scale,
                             double *d_scratch, int64_t scratch_stride, sk_prep *d_prep, void *d_mask2, int row16)
{
    if (grid <= 0) return SK_OK;
    ListedF64 la;
    la.list = d_list; la.count = d_count; la.mask2 = (unsigned char *)d_mask2; la.row16 = row16;
    la.scratch_stride = scratch_stride; la.len = nullptr;
    hipLaunchKernelGGL(k_prep_pa_listed, dim3(grid), dim3(TPB), 0, c->stream, d_sig, stride, d_len, d_cal, lo, hi, std_scale,
                       d_scratch, d_prep, la);
    SK_HIP(hipGetLastError());
    return SK_OK;
}

// The same statistics in numpy's order for the reads of a device-side list (the streaming float64 kernel's
// uncertified reads): segmenter mode rewrites those reads' {in band, kept} entries in place (d_scratch: `grid` rows of
// scratch_stride doubles), medmad mode their prep records and comp rows.
int sk_launch_prep_f64_listed(sk_ctx *c, const double *d_sig, const int64_t *d_off, const int32_t *d_list,
                              const int32_t *d_count, int grid, double lo, double hi, int mode, double std_scale,
                              double *d_comp_or_scratch, int64_t scratch_stride, sk_prep *d_prep, void *d_mask2, int row16,
                              const int32_t *d_rlen)
{
    if (grid <= 0) return SK_OK;
    ListedF64 la;
    la.list = d_list; la.count = d_count; la.mask2 = (unsigned char *)d_mask2; la.row16 = row16;
    la.scratch_stride = scratch_stride; la.len = d_rlen;
    hipLaunchKernelGGL(k_prep_f64<true>, dim3(grid), dim3(TPB), 0, c->stream, d_sig, d_off, 0, lo, hi, mode,
                       std_scale, d_comp_or_scratch, d_prep, (uint64_t *)nullptr, (int64_t)0, la);
    SK_HIP(hipGetLastError());
    return SK_OK;
}
