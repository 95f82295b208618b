// sk_sdtw.hip -- subsequence DTW (distance, start, end) for gfx950.
//
// Replaces mlpy.dtw_subsequence as MotifSeq uses it (/root/reference/MotifSeq.py:437-439):
// cdtw.c subsequence() fills D[i][j] = |x_i - y_j| + min3(D[i-1][j], D[i-1][j-1], D[i][j-1])
// with a free start along y (D[0][j] = |x_0 - y_j|), dtw.pyx takes the first argmin of the
// last row, subsequence_path() back-traces (diag first, then j-1, then i-1) and MotifSeq keeps
// only path_y[0], path_y[-1] and the distance.
//
// Design (wave-systolic, no cost matrix, no LDS, no MFMA -- it is a min-plus recurrence):
//   * a group of L lanes (L = 16: four reads per wave; L = 64: one read per wave) owns one
//     read; lane l owns R consecutive motif rows held in VGPRs together with their current
//     D (f64) and S (the column where the cell's back-trace would reach row 0; i32).
//   * at step t lane l processes column j = t - l: its R cells form a dependent chain
//     (up comes from the cell just computed), the values of lane l-1's bottom row arrive by
//     DPP row_shr:1 / wave_shr:1, and the read sample y_j marches lane to lane the same way.
//   * FP64 add/sub/compare/select only: every D[i][j] is one correctly rounded add of
//     correctly rounded operands, so the result is bit-identical to the CPU matrix whatever
//     the traversal order.  Start propagation reproduces the back-trace tie order exactly:
//         s1 = (left < diag) ? S_left : S_diag ;  s = (up < min(diag,left)) ? S_up : s1
//   * rows are distributed so that row 0 sits in slot 0 of lane 0 and row N-1 in the last
//     slot of the last lane: the first P = L*R - N lanes own R-1 rows ("short" lanes, their
//     last slot is a dead pad cell) -- so neither end needs a run-time slot select.
//   * boundaries come for free: D starts at +inf (column -1), lane 0's incoming "up" is the
//     virtual row -1 (D = 0, S = j + 1), samples past the end are +inf so they never win
//     the running argmin kept by the last lane.
//   * the normalisation (x - center) / scale of MotifSeq.py:192-200 / :186-191 is fused into
//     the sample feed: every L steps each lane normalises one sample of the next block.
#include "sk_common.h"
#include <math.h>
#include <string.h>
#include <vector>

namespace {

constexpr int DPP_ROW_SHR1  = 0x111;   // lane i <- lane i-1 inside a row of 16; lane 0 keeps `old`
constexpr int DPP_ROW_ROL1  = 0x12F;   // row_ror:15 == rotate left by one inside a row of 16
constexpr int DPP_WAVE_SHR1 = 0x138;   // lane i <- lane i-1 across the wave; lane 0 keeps `old`
constexpr int DPP_WAVE_ROL1 = 0x134;   // lane i <- lane i+1 across the wave (rotate)

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, 0xF, false);
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double old, double src)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xF, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

template <int L, int R, int FEED>
__global__ __launch_bounds__(256)
void k_sdtw(const void *__restrict__ samples, int64_t stride, const int64_t *__restrict__ off,
            const sk_prep *__restrict__ prep, int nreads, const double *__restrict__ xlay, int P,
            sk_hit *__restrict__ out, double *__restrict__ last_row)
{
    static_assert(L == 16 || L == 64, "lanes per read");
    constexpr int G = 64 / L;
    constexpr int SHR = (L == 16) ? DPP_ROW_SHR1 : DPP_WAVE_SHR1;
    constexpr int ROL = (L == 16) ? DPP_ROW_ROL1 : DPP_WAVE_ROL1;
    const double INF = __builtin_huge_val();

    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int g = lane / L, l = lane % L;
    int r = wave * G + g;
    const bool live = r < nreads;
    if (!live) r = nreads - 1;

    // ---- per-read parameters -------------------------------------------------
    int n, flags = 0;
    double center = 0.0, scale = 1.0;
    const int16_t *s16 = nullptr;
    const double  *s64 = nullptr;
    if constexpr (FEED == SK_FEED_I16) {
        const sk_prep pr = prep[r];
        n = pr.n; flags = pr.flags; center = pr.center; scale = pr.scale;
        s16 = (const int16_t *)samples + (int64_t)r * stride;
    } else if constexpr (FEED == SK_FEED_F64_NORM) {
        const sk_prep pr = prep[r];
        n = pr.n; flags = pr.flags; center = pr.center; scale = pr.scale;
        s64 = (const double *)samples + off[r];
    } else {
        n = (int)(off[r + 1] - off[r]);
        if (n == 0) flags = SK_FLAG_EMPTY;
        s64 = (const double *)samples + off[r];
    }
    if (!live) n = 0;

    int nmax = n;                                   // wave-uniform step count
#pragma unroll
    for (int d = L; d < 64; d <<= 1) nmax = max(nmax, __shfl_xor(nmax, d));
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    const int nblk = (nmax + L - 1 + L - 1) / L;    // steps 0 .. nmax-1 + L-1

    // ---- this lane's motif rows ------------------------------------------------
    double x[R];
#pragma unroll
    for (int k = 0; k < R; k++) x[k] = xlay[l * R + k];
    const bool shortlane = l < P;

    double D[R];
    int    S[R];
#pragma unroll
    for (int k = 0; k < R; k++) { D[k] = INF; S[k] = 0; }
    // my bottom row at my current column.  A lane 0 that owns no rows (R == 1, N < L) forwards
    // the virtual row -1, whose value at column -1 is (D = 0, S = 0).
    double botD = (R == 1 && l == 0 && shortlane) ? 0.0 : INF;  int botS = 0;
    double diagD = (l == 0) ? 0.0 : INF;  int diagS = 0;   // lane l-1's bottom one column back
    double y = INF;                                 // columns < 0: cost +inf keeps D at +inf
    double best = INF;  int bestS = -1, bestJ = -1;

    auto fetch = [&](int idx) -> double {           // normalised sample idx of my read, +inf past the end
        if constexpr (FEED == SK_FEED_I16) {
            int16_t raw = (idx < n) ? s16[idx] : (int16_t)0;
            double v = ((double)raw - center) / scale;
            return (idx < n) ? v : INF;
        } else if constexpr (FEED == SK_FEED_F64_NORM) {
            double raw = (idx < n) ? s64[idx] : 0.0;
            double v = (raw - center) / scale;
            return (idx < n) ? v : INF;
        } else {
            return (idx < n) ? s64[idx] : INF;
        }
    };

    double F = fetch(l);
    for (int blk = 0; blk < nblk; blk++) {
        const double Fnext = fetch((blk + 1) * L + l);     // in flight during the L steps below
#pragma unroll 2
        for (int q = 0; q < L; q++) {
            const int t = blk * L + q;
            // ---- systolic shift: sample and lane l-1's bottom row arrive -------
            y = dpp_f64<SHR>(F, y);                         // lane 0 takes sample t from the feed
            F = dpp_f64<ROL>(F, F);
            const double upD = dpp_f64<SHR>(0.0, botD);     // lane 0: virtual row -1 (D = 0)
            const int    upS = dpp_i32<SHR>(t + 1, botS);   //         whose S is column + 1
            // ---- R cells of column j = t - l ---------------------------------
            double dgD = diagD;  int dgS = diagS;           // (i-1, j-1)
            double uD = upD;     int uS = upS;              // (i-1, j)
#pragma unroll
            for (int k = 0; k < R; k++) {
                const double lfD = D[k];  const int lfS = S[k];     // (i, j-1)
                const double c = fabs(x[k] - y);
                const bool lt1 = lfD < dgD;                 // diag wins ties over left
                const double m1 = lt1 ? lfD : dgD;
                const int    s1 = lt1 ? lfS : dgS;
                const bool lt2 = uD < m1;                   // up only if strictly smaller
                const double m = lt2 ? uD : m1;
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
                botD = shortlane ? upD : D[0];              // a lane with no rows just forwards
                botS = shortlane ? upS : S[0];
            }
            // ---- running first-argmin of the last row (meaningful in lane L-1) --
            const int j = t - l;
            if (D[R - 1] < best) { best = D[R - 1]; bestS = S[R - 1]; bestJ = j; }
            if (last_row != nullptr) {
                if (l == L - 1 && r == 0 && j >= 0 && j < n) last_row[j] = D[R - 1];
            }
        }
        F = Fnext;
    }

    if (live && l == L - 1) {
        sk_hit h;
        if (n > 0) { h.dist = best; h.start = bestS; h.end = bestJ; }
        else       { h.dist = __builtin_nan(""); h.start = -1; h.end = -1; }
        h.n = n;
        h.flags = flags;
        out[r] = h;
    }
}

typedef void (*sdtw_fn)(const void *, int64_t, const int64_t *, const sk_prep *, int, const double *, int,
                        sk_hit *, double *);

template <int L, int FEED>
sdtw_fn pick_r(int R)
{
    switch (R) {
#define SK_CASE(RR) case RR: return k_sdtw<L, RR, FEED>;
        SK_CASE(1) SK_CASE(2) SK_CASE(3) SK_CASE(4) SK_CASE(5) SK_CASE(6) SK_CASE(7) SK_CASE(8)
        SK_CASE(9) SK_CASE(10) SK_CASE(11) SK_CASE(12) SK_CASE(13) SK_CASE(14) SK_CASE(15) SK_CASE(16)
#undef SK_CASE
    }
    return nullptr;
}

template <int FEED>
sdtw_fn pick(int L, int R)
{
    return (L == 16) ? pick_r<16, FEED>(R) : pick_r<64, FEED>(R);
}

} // namespace

// Host side: lay the motif out per lane, pick (L, R), launch.
int sk_launch_sdtw(sk_ctx *c, const sk_sdtw_args *a)
{
    const int N = a->nmotif;
    if (N <= 0) return sk_fail(SK_ERR_INVALID, "empty motif");
    if (a->nreads <= 0) return SK_OK;
    int L, R;
    if (N <= 16 * 16)      { L = 16; R = (N + 15) / 16; }
    else if (N <= 64 * 16) { L = 64; R = (N + 63) / 64; }
    else return sk_fail(SK_ERR_UNSUPPORTED, "motif of %d points exceeds the %d this build keeps in registers",
                        N, 64 * 16);
    const int P = L * R - N;                 // short lanes (own R-1 rows), always < L

    // The laid-out motif stays resident between calls; re-upload only when it changes.
    const bool same = c->motif.p && c->motif_src.size() == (size_t)N &&
                      memcmp(c->motif_src.data(), a->motif, (size_t)N * sizeof(double)) == 0;
    if (!same) {
        // the previous launch may still be reading the old layout
        SK_HIP(hipStreamSynchronize(c->stream));
        c->motif_host.assign((size_t)L * R, 0.0);
        int row = 0;
        for (int l = 0; l < L; l++) {
            int cnt = (l < P) ? R - 1 : R;
            for (int k = 0; k < cnt; k++) c->motif_host[(size_t)l * R + k] = a->motif[row++];
        }
        if (row != N) return sk_fail(SK_ERR_INVALID, "internal: motif layout mismatch");
        int rc = sk_reserve(c, &c->motif, c->motif_host.size() * sizeof(double));
        if (rc) return rc;
        SK_HIP(hipMemcpyAsync(c->motif.p, c->motif_host.data(), c->motif_host.size() * sizeof(double),
                              hipMemcpyHostToDevice, c->stream));
        c->motif_src.assign(a->motif, a->motif + N);
    }

    sdtw_fn fn = nullptr;
    switch (a->feed) {
        case SK_FEED_I16:      fn = pick<SK_FEED_I16>(L, R); break;
        case SK_FEED_F64_NORM: fn = pick<SK_FEED_F64_NORM>(L, R); break;
        case SK_FEED_F64_RAW:  fn = pick<SK_FEED_F64_RAW>(L, R); break;
    }
    if (!fn) return sk_fail(SK_ERR_UNSUPPORTED, "no kernel for L=%d R=%d", L, R);

    const int reads_per_block = 4 * (64 / L);
    const int grid = (a->nreads + reads_per_block - 1) / reads_per_block;
    SK_HIP(hipEventRecord(c->ev[2], c->stream));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, c->stream, a->samples, a->stride, a->off, a->prep,
                       a->nreads, (const double *)c->motif.p, P, a->out, a->last_row);
    SK_HIP(hipGetLastError());
    SK_HIP(hipEventRecord(c->ev[3], c->stream));
    return SK_OK;
}
