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
//     virtual row -1 (D = 0, S = j + 1), samples outside [0, n) are +inf so those columns stay
//     at +inf and never win the running argmin kept by the last lane.
//   * the normalisation (x - center) / scale of MotifSeq.py:192-200 / :186-191 is fused into
//     the sample feed: every L steps each lane normalises one sample of the next block.
//
// Three modes of the one kernel template:
//   FULL   one pass carrying D and S (12 of its 28 VALU cycles per cell are the S tracking).
//   DIST   pass A of the two-pass scheme: D only (16 cycles per cell) -> dist and end; every
//          CK steps each lane dumps its systolic state (R D's, y, bottom, diag) to HBM.
//   START  pass B: for each read restart from the last checkpoint at least `span` columns
//          before `end`, run with S tracking up to `end` only and read S there.  Cells of the
//          restart front carry S = -1; if the optimal path crosses the front the read's start
//          stays -1 and the read is appended to a retry list (FULL pass on those reads only).
//          The restart reproduces the state bit for bit, so the result is exact either way.
#include "sk_sdtw_dev.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

template <int L, int R, int FEED, int MODE>
__global__ __launch_bounds__(256)
void k_sdtw(const sdtw_kargs a)
{
    static_assert(L == 16 || L == 64, "lanes per read");
    constexpr int G = 64 / L;
    constexpr int SHR = (L == 16) ? DPP_ROW_SHR1 : DPP_WAVE_SHR1;
    constexpr int ROL = (L == 16) ? DPP_ROW_ROL1 : DPP_WAVE_ROL1;
    constexpr bool TRACK = (MODE != MODE_DIST);
    constexpr int CKW = R + 3;                      // doubles per lane per checkpoint
    const double INF = __builtin_huge_val();

    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int g = lane / L, l = lane % L;
    int slot = wave * G + g;
    int nreads = a.nreads;
    if (a.gate_ptr) {                               // whole-call fallback: runs only when the guard raised an alarm
        if (*a.gate_ptr == 0) return;               // (launch-uniform)
        if (a.guard && blockIdx.x == 0 && threadIdx.x == 0) a.guard[SK_GUARD_FELLBACK] = 1;
    }
    if (a.count_ptr) {                              // retry pass: the list length is only known on the device
        const int cnt = *a.count_ptr;
        if (a.total_ptr && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.total_ptr, cnt);
        nreads = min(cnt - a.list_off, a.nreads);
        if (nreads <= 0) return;                    // (block-uniform) nothing listed for this launch
    }
    const bool live = slot < nreads;
    if (!live) slot = nreads - 1;
    const int r = a.ridx ? a.ridx[a.list_off + slot] : a.read0 + slot;

    // ---- per-read parameters -------------------------------------------------
    int n, flags = 0;
    double center = 0.0, scale = 1.0, rc1 = 0.0, rc2 = 0.0;
    const int16_t *s16 = nullptr;
    const double  *s64 = nullptr;
    if constexpr (FEED == SK_FEED_I16) {
        const sk_prep pr = a.prep[r];
        n = pr.n; flags = pr.flags & SK_FLAG_PUBLIC; center = pr.center; scale = pr.scale;
        s16 = (const int16_t *)a.samples + (int64_t)r * a.stride;
    } else if constexpr (FEED == SK_FEED_F64_NORM) {
        const sk_prep pr = a.prep[r];
        n = pr.n; flags = pr.flags & SK_FLAG_PUBLIC; center = pr.center; scale = pr.scale;
        rc1 = pr.top; rc2 = pr.bot;                 // zscale re-centring terms (0.0 unless sklearn applied them)
        s64 = (const double *)(((pr.flags & SK_IFLAG_INPLACE) && a.samples_raw) ? a.samples_raw : a.samples) + a.off[r];
    } else {
        n = (int)(a.off[r + 1] - a.off[r]);
        if (n == 0) flags = SK_FLAG_EMPTY;
        s64 = (const double *)a.samples + a.off[r];
    }
    if (!live) n = 0;

    // ---- step range of this group: [tbase, tlast] ---------------------------------
    int tbase = 0, tlast = n - 1 + L - 1, end = -1, c0 = 0;
    if (n <= 0) tlast = -1;
    if constexpr (MODE == MODE_START) {
        end = a.out[r].end;
        if (n > 0 && end >= 0) {
            c0 = max(0, end - a.span) / a.ck;       // last checkpoint at least `span` columns back
            if (c0 > a.nck) c0 = a.nck;
            tbase = c0 * a.ck;
            tlast = end + L - 1;                    // lane L-1 reaches column `end`
        } else {
            tlast = -1;
        }
    }
    int nsteps = tlast - tbase + 1;                 // wave-uniform step count
    if (nsteps < 0) nsteps = 0;
#pragma unroll
    for (int d = L; d < 64; d <<= 1) nsteps = max(nsteps, __shfl_xor(nsteps, d));
    nsteps = __builtin_amdgcn_readfirstlane(nsteps);
    const int nblk = (nsteps + L - 1) / L;

    // ---- this lane's motif rows ------------------------------------------------
    double x[R];
#pragma unroll
    for (int k = 0; k < R; k++) x[k] = a.xlay[l * R + k];
    const bool shortlane = l < a.P;

    double D[R];
    int    S[R];
#pragma unroll
    for (int k = 0; k < R; k++) { D[k] = INF; S[k] = -1; }
    // my bottom row at my current column.  A lane 0 that owns no rows (R == 1, N < L) forwards
    // the virtual row -1, whose value at column -1 is (D = 0, S = 0).
    double botD = (R == 1 && l == 0 && shortlane) ? 0.0 : INF;
    int    botS = (R == 1 && l == 0 && shortlane) ? 0 : -1;
    double diagD = (l == 0) ? 0.0 : INF;            // lane l-1's bottom one column back
    int    diagS = (l == 0) ? tbase : -1;           // virtual row -1 at column tbase-1: S = column + 1
    // MODE_CHAIN, chunks below the first: the row above lane 0 is the last row of the previous chunk
    // (per column, from memory) instead of the virtual row; it does not exist at column -1
    bool chained = false;
    const double *pD = nullptr;  const int32_t *pS = nullptr;
    double PFD = 0.0;  int PFS = 0;
    if constexpr (MODE == MODE_CHAIN) {
        chained = a.prevD != nullptr;
        if (chained) {
            pD = a.prevD + (int64_t)(r - a.read0) * a.row_stride;
            pS = a.prevS + (int64_t)(r - a.read0) * a.row_stride;
            if (l == 0) { diagD = INF; diagS = -1; }
            if (R == 1 && l == 0 && shortlane) { botD = INF; botS = -1; }
            PFD = (l < n) ? pD[l] : INF;
            PFS = (l < n) ? pS[l] : -1;
        }
    }
    double y = INF;                                 // columns < 0: cost +inf keeps D at +inf
    double best = INF;  int bestS = -1, bestJ = -1;

    if constexpr (MODE == MODE_START) {
        if (c0 > 0) {                               // restart from the saved systolic state
            const double *cp = a.ckpt + (((int64_t)(r - a.read0) * a.nck + (c0 - 1)) * L + l) * CKW;
#pragma unroll
            for (int k = 0; k < R; k++) D[k] = cp[k];
            y = cp[R]; botD = cp[R + 1]; diagD = cp[R + 2];
        }
    }

    auto fetch = [&](int idx) -> double {           // normalised sample idx of my read, +inf outside
        if constexpr (FEED == SK_FEED_I16) {
            int16_t raw = (idx < n) ? s16[idx] : (int16_t)0;
            double v = ((double)raw - center) / scale;
            return (idx < n) ? v : INF;
        } else if constexpr (FEED == SK_FEED_F64_NORM) {
            double raw = (idx < n) ? s64[idx] : 0.0;
            double v = ((raw - center) - rc1) / scale - rc2;
            return (idx < n) ? v : INF;
        } else {
            return (idx < n) ? s64[idx] : INF;
        }
    };

    double F = fetch(tbase + l);
    for (int blk = 0; blk < nblk; blk++) {
        const double Fnext = fetch(tbase + (blk + 1) * L + l);   // in flight during the L steps below
        double PFDn = 0.0;  int PFSn = 0;
        if constexpr (MODE == MODE_CHAIN) {
            if (chained) {
                const int idx = tbase + (blk + 1) * L + l;
                PFDn = (idx < n) ? pD[idx] : INF;
                PFSn = (idx < n) ? pS[idx] : -1;
            }
        }
        if constexpr (MODE == MODE_DIST) {
            // checkpoint c (>= 1) = state at the beginning of step c*ck (ck is a multiple of L)
            const int t0 = blk * L;
            if (t0 > 0 && (t0 % a.ck) == 0 && t0 / a.ck <= a.nck && live) {
                double *cp = a.ckpt + (((int64_t)(r - a.read0) * a.nck + (t0 / a.ck - 1)) * L + l) * CKW;
#pragma unroll
                for (int k = 0; k < R; k++) cp[k] = D[k];
                cp[R] = y; cp[R + 1] = botD; cp[R + 2] = diagD;
            }
        }
#pragma unroll 2
        for (int q = 0; q < L; q++) {
            const int t = tbase + blk * L + q;
            // ---- systolic shift: sample and lane l-1's bottom row arrive -------
            y = dpp_f64<SHR>(F, y);                         // lane 0 takes sample t from the feed
            F = dpp_f64<ROL>(F, F);
            double topD = 0.0;  int topS = t + 1;           // lane 0: virtual row -1 (D = 0, S = column + 1)
            if constexpr (MODE == MODE_CHAIN) {
                if (chained) { topD = PFD; topS = PFS; }    //   or the previous chunk's last row at column t
                PFD = dpp_f64<ROL>(PFD, PFD);
                PFS = dpp_i32<ROL>(PFS, PFS);
            }
            const double upD = dpp_f64<SHR>(topD, botD);
            int upS = 0;
            if constexpr (TRACK) upS = dpp_i32<SHR>(topS, botS);
            // ---- R cells of column j = t - l ---------------------------------
            double dgD = diagD;  int dgS = diagS;           // (i-1, j-1)
            double uD = upD;     int uS = upS;              // (i-1, j)
#pragma unroll
            for (int k = 0; k < R; k++) {
                const double lfD = D[k];                    // (i, j-1)
                const double c = fabs(x[k] - y);
                double nd;
                if constexpr (TRACK) {
                    const int lfS = S[k];
                    const bool lt1 = lfD < dgD;             // diag wins ties over left
                    // int16 reads normalise to finite values (a degenerate read -- scale 0 -- is flagged and its
                    // result unspecified), so there v_min_f64 is the select; written as a select on doubles the
                    // compiler spends two v_cndmask each.  float64 input may carry NaN: keep the literal select.
                    double m1;
                    if constexpr (FEED == SK_FEED_I16) m1 = vmin(lfD, dgD); else m1 = lt1 ? lfD : dgD;
                    const int    s1 = lt1 ? lfS : dgS;
                    const bool lt2 = uD < m1;               // up only if strictly smaller
                    double m;
                    if constexpr (FEED == SK_FEED_I16) m = vmin(uD, m1); else m = lt2 ? uD : m1;
                    const int    s = lt2 ? uS : s1;
                    nd = c + m;
                    dgS = lfS;  S[k] = s;  uS = s;
                } else {
                    nd = c + vmin(vmin(dgD, lfD), uD);
                }
                dgD = lfD;
                D[k] = nd;
                uD = nd;
            }
            diagD = upD;  diagS = upS;
            if constexpr (R >= 2) {
                botD = shortlane ? D[R - 2] : D[R - 1];
                if constexpr (TRACK) botS = shortlane ? S[R - 2] : S[R - 1];
            } else {
                botD = shortlane ? upD : D[0];              // a lane with no rows just forwards
                if constexpr (TRACK) botS = shortlane ? upS : S[0];
            }
            const int j = t - l;
            if constexpr (MODE == MODE_START) {
                if (j == end) bestS = S[R - 1];             // S of cell (N-1, end), lane L-1
            } else {
                // ---- running first-argmin of the last row (meaningful in lane L-1) --
                if (D[R - 1] < best) {
                    best = D[R - 1]; bestJ = j;
                    if constexpr (TRACK) bestS = S[R - 1];
                }
                if constexpr (MODE == MODE_FULL || MODE == MODE_CHAIN) {
                    if (a.last_row != nullptr) {
                        if (l == L - 1 && slot == 0 && j >= 0 && j < n) a.last_row[j] = D[R - 1];
                    }
                }
                if constexpr (MODE == MODE_CHAIN) {
                    if (a.rowD != nullptr && live && l == L - 1 && j >= 0 && j < n) {
                        a.rowD[(int64_t)(r - a.read0) * a.row_stride + j] = D[R - 1];
                        a.rowS[(int64_t)(r - a.read0) * a.row_stride + j] = S[R - 1];
                    }
                }
            }
        }
        F = Fnext;
        if constexpr (MODE == MODE_CHAIN) { PFD = PFDn; PFS = PFSn; }
    }

    if (live && l == L - 1) {
        if constexpr (MODE == MODE_START) {
            if (n > 0 && end >= 0) {
                if (bestS >= 0) a.out[r].start = bestS;
                else a.retry[atomicAdd(a.retry_cnt, 1)] = r;
            }
        } else {
            sk_hit h;
            if (n > 0) { h.dist = best; h.start = bestS; h.end = bestJ; }
            else       { h.dist = __builtin_nan(""); h.start = -1; h.end = -1; }
            h.n = n;
            h.flags = flags;
            a.out[a.out_by_slot ? a.list_off + slot : r] = h;
        }
    }
}

// ---- the audit (round 5): which reads, and the comparison -----------------------------------------------------
// read of audit slot s: a hashed position inside [s * period, (s + 1) * period).  `salt` is the context's count of
// audited calls: a stream of equally shaped calls audits different positions call after call (lane group within a wave,
// chunk boundary, place in the sorted order), and a call's choice is still a function of (call number, shape) alone.
__global__ void k_audit_pick(int32_t *list, int naudit, int period, int nreads, unsigned salt)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s == 0) { list[0] = naudit; list[1] = 0; }
    if (s >= naudit) return;
    const int64_t base = (int64_t)s * period;
    const int span = (int)min((int64_t)period, (int64_t)nreads - base);
    unsigned h = (unsigned)s * 2654435761u + 0x9E3779B9u;
    h ^= salt * 0x9E3779B9u;
    h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
    list[2 + s] = (int)(base + (int64_t)(h % (unsigned)span));
}

// exact record vs the screening scheme's: distance bit for bit (two NaNs agree), start, end, n.  The exact one wins.
__global__ void k_audit_compare(const int32_t *list, const sk_hit *exact, sk_hit *out, int naudit, int32_t *guard)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= naudit) return;
    const int r = list[2 + s];
    const sk_hit e = exact[s], g = out[r];
    const bool same_d = (__double_as_longlong(e.dist) == __double_as_longlong(g.dist)) || (e.dist != e.dist && g.dist != g.dist);
    if (same_d && e.start == g.start && e.end == g.end && e.n == g.n) return;
    out[r] = e;
    atomicAdd(&guard[SK_GUARD_MISMATCH], 1);
    atomicAdd(&guard[SK_GUARD_ALARM], 1);
}

static void sk_audit_pick(int32_t *list, int naudit, int period, int nreads, unsigned salt, hipStream_t st)
{
    hipLaunchKernelGGL(k_audit_pick, dim3((naudit + 255) / 256), dim3(256), 0, st, list, naudit, period, nreads, salt);
}
static void sk_audit_compare(const int32_t *list, const sk_hit *exact, sk_hit *out, int naudit, int32_t *guard, hipStream_t st)
{
    hipLaunchKernelGGL(k_audit_compare, dim3((naudit + 255) / 256), dim3(256), 0, st, list, exact, out, naudit, guard);
}

typedef void (*sdtw_fn)(const sdtw_kargs);

template <int L, int FEED, int MODE>
sdtw_fn pick_r(int R)
{
    switch (R) {
#define SK_CASE(RR) case RR: return k_sdtw<L, RR, FEED, MODE>;
        SK_CASE(1) SK_CASE(2) SK_CASE(3) SK_CASE(4) SK_CASE(5) SK_CASE(6) SK_CASE(7) SK_CASE(8)
        SK_CASE(9) SK_CASE(10) SK_CASE(11) SK_CASE(12) SK_CASE(13) SK_CASE(14) SK_CASE(15) SK_CASE(16)
#undef SK_CASE
    }
    return nullptr;
}

template <int FEED>
sdtw_fn pick(int L, int R, int mode)
{
    if (L == 16) {
        if (mode == MODE_FULL) return pick_r<16, FEED, MODE_FULL>(R);
        if (mode == MODE_DIST) return pick_r<16, FEED, MODE_DIST>(R);
        return pick_r<16, FEED, MODE_START>(R);
    }
    if (mode == MODE_FULL) return pick_r<64, FEED, MODE_FULL>(R);
    if (mode == MODE_DIST) return pick_r<64, FEED, MODE_DIST>(R);
    if (mode == MODE_CHAIN) return pick_r<64, FEED, MODE_CHAIN>(R);
    return pick_r<64, FEED, MODE_START>(R);
}

sdtw_fn pick_any(int feed, int L, int R, int mode)
{
    switch (feed) {
        case SK_FEED_I16:      return pick<SK_FEED_I16>(L, R, mode);
        case SK_FEED_F64_NORM: return pick<SK_FEED_F64_NORM>(L, R, mode);
        case SK_FEED_F64_RAW:  return pick<SK_FEED_F64_RAW>(L, R, mode);
    }
    return nullptr;
}

int launch(sk_ctx *c, sdtw_fn fn, const sdtw_kargs &k, int L, hipStream_t stream = nullptr)
{
    if (k.nreads <= 0) return SK_OK;
    const int reads_per_block = 4 * (64 / L);
    const int grid = (k.nreads + reads_per_block - 1) / reads_per_block;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, stream ? stream : c->stream, k);
    SK_HIP(hipGetLastError());
    return SK_OK;
}

} // namespace

// Motifs of more than 1024 points: chunks of 1024 rows (64 lanes x 16) are swept one after the other
// with the exact single pass; the last row of a chunk (cost and start column, per read column) goes
// through memory and enters the next chunk where the virtual row -1 enters the first.  Reads are
// processed in batches so that the two row buffers stay within a fixed budget.
static int launch_chained(sk_ctx *c, const sk_sdtw_args *a)
{
    const int N = a->nmotif;
    const int CH = 64 * 16;
    const int nchunks = (N + CH - 1) / CH;
    // layouts of all chunks back to back; chunk i at lay_off[i]
    std::vector<size_t> lay_off(nchunks);
    std::vector<int> Rc(nchunks), Pc(nchunks);
    size_t total = 0;
    for (int i = 0; i < nchunks; i++) {
        const int rows = (i + 1 < nchunks) ? CH : N - i * CH;
        Rc[i] = (rows + 63) / 64;
        Pc[i] = 64 * Rc[i] - rows;
        lay_off[i] = total;
        total += (size_t)64 * Rc[i];
    }
    const bool same = c->motif.p && c->motif_L == -1 && c->motif_src.size() == (size_t)N &&
                      memcmp(c->motif_src.data(), a->motif, (size_t)N * sizeof(double)) == 0;
    if (!same) {
        SK_HIP(hipStreamSynchronize(c->stream));
        c->motif_host.assign(total, 0.0);
        int row = 0;
        for (int i = 0; i < nchunks; i++)
            for (int l = 0; l < 64; l++) {
                const int cnt = (l < Pc[i]) ? Rc[i] - 1 : Rc[i];
                for (int k = 0; k < cnt; k++) c->motif_host[lay_off[i] + (size_t)l * Rc[i] + k] = a->motif[row++];
            }
        if (row != N) return sk_fail(SK_ERR_INVALID, "internal: motif layout mismatch");
        int rc = sk_reserve(c, &c->motif, total * sizeof(double));
        if (rc) return rc;
        SK_HIP(hipMemcpyAsync(c->motif.p, c->motif_host.data(), total * sizeof(double), hipMemcpyHostToDevice,
                              c->stream));
        c->motif_src.assign(a->motif, a->motif + N);
        c->motif_L = -1;                            // (chunked layout: never equal to a plain one)
        c->motif64_valid = false;
        c->motifq_valid = false;
    }
    // row buffers: two (ping-pong) of [batch][row_stride] doubles + ints
    const int64_t row_stride = (a->max_len > 0 ? a->max_len : 1);
    const size_t per_read = (size_t)row_stride * 12 * 2;
    int64_t batch = (int64_t)(((size_t)6 << 30) / per_read);
    if (batch > a->nreads) batch = a->nreads;
    if (batch < 1) batch = 1;
    int rc = sk_reserve(c, &c->ckpt, (size_t)batch * per_read);
    if (rc) return rc;
    double  *bufD[2] = {(double *)c->ckpt.p, (double *)c->ckpt.p + (size_t)batch * row_stride};
    int32_t *bufS[2] = {(int32_t *)(bufD[1] + (size_t)batch * row_stride),
                        (int32_t *)(bufD[1] + (size_t)batch * row_stride) + (size_t)batch * row_stride};
    sdtw_kargs k;
    memset(&k, 0, sizeof k);
    k.samples = a->samples; k.samples_raw = a->samples_raw; k.stride = a->stride; k.off = a->off; k.prep = a->prep;
    k.out = a->out; k.last_row = nullptr; k.row_stride = row_stride;
    c->last_retry = 0;
    c->retry_dev = false;
    c->prof_chunks = 0;
    SK_HIP(hipEventRecord(c->ev[2], c->stream));
    for (int64_t r0 = 0; r0 < a->nreads; r0 += batch) {
        k.read0 = (int)r0;
        k.nreads = (int)((a->nreads - r0 < batch) ? a->nreads - r0 : batch);
        for (int i = 0; i < nchunks; i++) {
            sdtw_fn fn = pick_any(a->feed, 64, Rc[i], MODE_CHAIN);
            if (!fn) return sk_fail(SK_ERR_UNSUPPORTED, "no chained kernel for R=%d", Rc[i]);
            k.xlay = (const double *)c->motif.p + lay_off[i];
            k.P = Pc[i];
            k.prevD = (i > 0) ? bufD[(i - 1) & 1] : nullptr;
            k.prevS = (i > 0) ? bufS[(i - 1) & 1] : nullptr;
            k.rowD = (i + 1 < nchunks) ? bufD[i & 1] : nullptr;
            k.rowS = (i + 1 < nchunks) ? bufS[i & 1] : nullptr;
            k.last_row = (i + 1 == nchunks && r0 == 0) ? a->last_row : nullptr;
            if ((rc = launch(c, fn, k, 64))) return rc;
        }
    }
    SK_HIP(hipEventRecord(c->ev[3], c->stream));
    return SK_OK;
}

// Host side: lay the motif out per lane, pick (L, R), choose one or two passes, launch.
// (the fused prologue is instantiated for histograms of 1 025 .. 1 280 bins: the default limits 0 / 1 200)
bool sk_sdtw_fuse_ok(int32_t lo, int32_t hi, int mode, int64_t stride)
{
    if (sk_tune("SK_PREP_BLOCK") || sk_tune("SK_DTW_NOFUSE")) return false;
    if (mode == SK_PREP_ZSCALE) return stride > 0 && stride <= 4096 && hi > lo;
    const int64_t nbins = (int64_t)hi - lo - 1;
    return nbins > 1024 && nbins <= 1280;
}

// will this call take the screening scheme (sk_sdtwq.hip)?  (the look-back / checkpoint numbers as in sk_launch_sdtw)
static bool screens(const sk_sdtw_args *a, int span, int ck)
{
    const int N = a->nmotif;
    if (N > 64 * 16 || a->last_row || a->force_single || a->nreads < 256) return false;
    if (a->max_len < 4 * (int64_t)(span + ck)) return false;
    if (const char *e = sk_tune("SK_DTW_SCHEME")) if (strcmp(e, "full") == 0 || strcmp(e, "exact2") == 0) return false;
    for (int i = 0; i < N; i++) if (!(fabs(a->motif[i]) < QLIM)) return false;
    return true;
}

int sk_launch_sdtw(sk_ctx *c, const sk_sdtw_args *a_in)
{
    const sk_sdtw_args *a = a_in;
    sk_sdtw_args a_unfused;
    const int N = a->nmotif;
    if (N <= 0) return sk_fail(SK_ERR_INVALID, "empty motif");
    if (a->nreads <= 0) return SK_OK;
    if (a->fuse) {
        int ck0 = 128, span0 = N + N / 8 + 8;
        if (const char *e = sk_tune("SK_DTW_CK")) { int v = atoi(e); if (v >= 64 && v % 64 == 0) ck0 = v; }
        if (const char *e = sk_tune("SK_DTW_SPAN")) { int v = atoi(e); if (v > 0) span0 = v; }
        if (!screens(a, span0, ck0) || !sk_sdtw_fuse_ok(a->fuse->lo, a->fuse->hi, a->fuse->mode, a->stride)) {
            // no screening pass to carry the prologue: filter + statistics as their own kernel, now
            SK_HIP(hipEventRecord(c->ev[0], c->stream));
            int rc0 = sk_launch_prep_i16(c, a->fuse->raw, a->stride, a->fuse->len, a->nreads, a->fuse->lo, a->fuse->hi,
                                         a->fuse->mode, 0.0, (int16_t *)a->samples, (sk_prep *)a->prep, nullptr, 0);
            if (rc0) return rc0;
            SK_HIP(hipEventRecord(c->ev[1], c->stream));
            a_unfused = *a;
            a_unfused.fuse = nullptr;
            a = &a_unfused;
        }
    }
    int L, R;
    if (N <= 16 * 16)      { L = 16; R = (N + 15) / 16; }
    else if (N <= 64 * 16) { L = 64; R = (N + 63) / 64; }
    else return launch_chained(c, a);               // more rows than a wavefront keeps in registers
    // A couple of thousand reads cannot fill the chip four to a wavefront: what counts then is the
    // latency of one sweep, which is shorter with the read spread over 64 lanes (fewer rows per lane).
    // Measured at 163 points x 4 000 samples: 64 reads 1.12 -> 0.54 ms, 1 024 reads 0.52 -> 0.29 ms,
    // break-even near 4 096 reads.
    int small_max = 2048;
    if (const char *e = sk_tune("SK_DTW_SMALL_MAX")) { int v = atoi(e); if (v >= 0) small_max = v; }
    if (L == 16 && N >= 32 && a->nreads <= small_max && !sk_tune("SK_DTW_NO_SMALL")) { L = 64; R = (N + 63) / 64; }
    const int P = L * R - N;                 // short lanes (own R-1 rows), always < L

    // The laid-out motif stays resident between calls; re-upload only when it changes.
    const bool same = c->motif.p && c->motif_L == L && c->motif_src.size() == (size_t)N &&
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
        c->motif_L = L;
        c->motif64_valid = false;
        c->motifq_valid = false;
    }

    sdtw_kargs k;
    memset(&k, 0, sizeof k);
    k.samples = a->samples; k.samples_raw = a->samples_raw; k.stride = a->stride; k.off = a->off; k.prep = a->prep;
    k.nreads = a->nreads; k.read0 = 0; k.ridx = nullptr; k.xlay = (const double *)c->motif.p; k.P = P;
    k.out = a->out; k.last_row = a->last_row;

    // ---- one pass or two? ---------------------------------------------------------------
    // Two passes pay when reads are much longer than the look-back window; the caller's
    // max_len bounds every read's filtered length.
    // steps between checkpoints (multiple of both L) and look-back in columns; env overrides are
    // for tuning runs only.  Typical optimal paths span about N columns (140 for N=163).
    // Measured on the C4 workload: optimal paths are 0.4-1.06 N columns wide (median 0.46 N for
    // reads without the motif, 0.98 N with it); a path wider than the look-back only costs that
    // read an exact retry.
    // Look-back in two tiers (screening scheme): every read first gets N + 8 columns (optimal paths on the C4
    // workload are 0.4-1.0 N columns wide) and the reads whose path crossed that front are redone with 2 N + N/4 + 16 columns before anything falls back to the
    // exact single pass (measured, C4: window pass 19.3 ms per 1 M reads with one tier of N + N/8 + 8 columns,
    // 17.4 ms at N columns).  The exact two-pass scheme keeps one tier of N + N/8 + 8.
    int ck = 128;
    int span = N + N / 8 + 8, span_q = N + 8, span2 = 2 * N + N / 4 + 16;
    if (const char *e = sk_tune("SK_DTW_CK")) { int v = atoi(e); if (v >= 64 && v % 64 == 0) ck = v; }
    if (const char *e = sk_tune("SK_DTW_SPAN")) { int v = atoi(e); if (v > 0) { span = span_q = v; span2 = 0; } }
    if (const char *e = sk_tune("SK_DTW_SPAN2")) { int v = atoi(e); if (v >= 0) span2 = v; }
    const int64_t maxlen = a->max_len;
    const char *scheme = sk_tune("SK_DTW_SCHEME");          // A/B switch: "full" = the exact single pass
    const bool two_pass = !a->last_row && !a->force_single && maxlen >= 4 * (int64_t)(span + ck) &&
                          a->nreads >= 256 && !(scheme && strcmp(scheme, "full") == 0);
    SK_HIP(hipEventRecord(c->ev[2], c->stream));
    if (a->fuse && !two_pass) return sk_fail(SK_ERR_INVALID, "internal: fused prologue without a screening pass");
    if (!two_pass) {
        c->last_retry = 0;
        c->retry_dev = false;
        c->prof_chunks = 0;
        sdtw_fn fn = pick_any(a->feed, L, R, MODE_FULL);
        if (!fn) return sk_fail(SK_ERR_UNSUPPORTED, "no kernel for L=%d R=%d", L, R);
        int rc = launch(c, fn, k, L);
        if (rc) return rc;
        SK_HIP(hipEventRecord(c->ev[3], c->stream));
        return SK_OK;
    }

    sdtw_fn fa = pick_any(a->feed, L, R, MODE_DIST);
    sdtw_fn fb = pick_any(a->feed, L, R, MODE_START);
    sdtw_fn ff = pick_any(a->feed, L, R, MODE_FULL);
    if (!fa || !fb || !ff) return sk_fail(SK_ERR_UNSUPPORTED, "no kernel for L=%d R=%d", L, R);

    // ---- fixed-point screening + certified exact window (sk_sdtwq.hip) ------------------------
    // needs every motif value inside the fixed-point range; SK_DTW_SCHEME=exact2 keeps the
    // two exact FP64 passes below (A/B runs).
    bool qok = true;
    for (int i = 0; i < N; i++) qok = qok && (fabs(a->motif[i]) < QLIM);
    if (const char *e = sk_tune("SK_DTW_SCHEME")) qok = qok && strcmp(e, "exact2") != 0;
    // Reads the windowed pass cannot certify are appended to a device-side list and redone by the exact single
    // pass.  The host never learns the count (no sync inside the call): the retry launches are sized for the
    // worst case and return at once where the list ends.  A short list is latency-bound, so its first 8 192
    // entries are swept with each read spread over 64 lanes; whatever lies beyond uses the batch layout.
    int rc;
    if ((rc = sk_reserve(c, &c->retry, 2 * ((size_t)a->nreads + 2) * sizeof(int32_t)))) return rc;
    if ((rc = sk_reserve(c, &c->dtwcnt, 128))) return rc;
    int32_t *cnt = (int32_t *)c->retry.p;                  // [0] = counter, [2..] = read indices
    int32_t *ecnt = cnt + a->nreads + 2;                   // a second list of the same shape: the early retry's
    SK_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t), c->stream));
    SK_HIP(hipMemsetAsync(ecnt, 0, sizeof(int32_t), c->stream));
    if (!a->accumulate) SK_HIP(hipMemsetAsync(c->dtwcnt.p, 0, 128, c->stream));  // [0] retried, [1] second tier, +16: clock, +32: guard counters, +64: window-pass steps
    c->retry_dev = true;
    // the motif laid out for 64 lanes (the short retry list is swept with a read per wavefront): uploaded here, ahead
    // of everything the call enqueues, because the early retry runs on another stream
    if (L == 16 && !c->motif64_valid && pick_any(a->feed, 64, (N + 63) / 64, MODE_FULL)) {
        const int R64 = (N + 63) / 64, P64 = 64 * R64 - N;
        SK_HIP(hipStreamSynchronize(c->stream));          // an earlier launch may still read it
        if (c->stream3) SK_HIP(hipStreamSynchronize(c->stream3));
        if (c->stream4) SK_HIP(hipStreamSynchronize(c->stream4));
        c->motif64_host.assign((size_t)64 * R64, 0.0);
        int row = 0;
        for (int l = 0; l < 64; l++) {
            const int rows = (l < P64) ? R64 - 1 : R64;
            for (int kk = 0; kk < rows; kk++) c->motif64_host[(size_t)l * R64 + kk] = a->motif[row++];
        }
        if ((rc = sk_reserve(c, &c->motif64, c->motif64_host.size() * sizeof(double)))) return rc;
        SK_HIP(hipMemcpyAsync(c->motif64.p, c->motif64_host.data(),
                              c->motif64_host.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        c->motif64_valid = true;
    }
    auto launch_retry = [&](int32_t *list = nullptr, hipStream_t stream = nullptr) -> int {
        if (!list) list = cnt;
        sdtw_kargs kr = k;
        kr.read0 = 0; kr.ridx = list + 2; kr.count_ptr = list; kr.ckpt = nullptr;
        kr.total_ptr = (int32_t *)c->dtwcnt.p; kr.list_off = 0;
        const int R64 = (N + 63) / 64, P64 = 64 * R64 - N;
        sdtw_fn f64 = (L == 16) ? pick_any(a->feed, 64, R64, MODE_FULL) : nullptr;
        if (f64) {
            sdtw_kargs k64 = kr;
            k64.xlay = (const double *)c->motif64.p; k64.P = P64;
            k64.nreads = a->nreads < 8192 ? a->nreads : 8192;
            int rc2;
            if ((rc2 = launch(c, f64, k64, 64, stream))) return rc2;
            if (a->nreads <= 8192) return SK_OK;
            kr.list_off = 8192; kr.total_ptr = nullptr; kr.nreads = a->nreads - 8192;
            return launch(c, ff, kr, L, stream);
        }
        kr.nreads = a->nreads;
        return launch(c, ff, kr, L, stream);
    };
    if (a->fuse && !qok) return sk_fail(SK_ERR_INVALID, "internal: fused prologue without a screening pass");
    // ---- the guard around the screening scheme (round 5) -----------------------------------------
    // (a) pass W tests the certificate's premise on every result it accepts (k_sdtw_w: premise_holds);
    // (b) AUDIT: one read in every `period` (hashed position inside each run of `period` reads; 4 096) is also swept
    //     by the exact single pass, on the third stream beside the window passes, into records of its own; a compare
    //     kernel behind everything counts the records that differ, and the exact one wins;
    // (c) FALLBACK: if (a) or (b) counted anything the premise is broken for reasons unknown, so no record of this LAUNCH
    //     SET (one motif over one sub-batch: what this function enqueues) is trusted: the exact single pass over all of
    //     its reads is enqueued behind a gate word and returns at once while that word is zero (no host synchronisation
    //     anywhere).  sk_last_dtw_guard() reports the counters.  Scope: an API call that makes several launch sets
    //     (multi-motif, the host entry points' ingest sub-batches) shares ONE set of counters (`accumulate`), so after
    //     an alarm every LATER launch set of the call runs exactly as well (the gate stays open); launch sets that
    //     finished BEFORE the alarm without one of their own are not redone -- each of them passed its own premise
    //     test and its own audit.  The message on stderr says "launch set(s)" for that reason.
    int32_t *guard = (int32_t *)c->dtwcnt.p + 8;
    const bool guarded = qok && sk_tune("SK_DTW_NOGUARD") == nullptr;
    int audit_period = 4096;
    if (const char *e = sk_tune("SK_DTW_AUDIT_PERIOD")) { const int v = atoi(e); if (v >= 0) audit_period = v; }
    int naudit = (guarded && audit_period > 0) ? (int)((a->nreads + audit_period - 1) / audit_period) : 0;
    // The audit costs one exact sweep's LATENCY whatever the number of reads it takes (one wavefront each, side by side):
    // (maxlen + 64) steps of 8 ceil(N/64) + 20 instructions.  Beside the window passes of a large call that is free;
    // a call of few, long reads (25 000 of 37 000 samples: 3.4 ms of sweep beside 0.9 ms of windows, in a 20 ms call)
    // would wait for it.  So where the sweep is expected to outlast the window passes by more than 2 % of the call,
    // only every K-th such call of this context is audited, K chosen to keep the AVERAGE cost at those 2 %.  The
    // premise test (a) and the image bound stay on every call; an explicit SK_DTW_AUDIT_PERIOD audits every call.
    if (naudit && sk_tune("SK_DTW_AUDIT_PERIOD") == nullptr) {
        const double audit_ms = (double)(maxlen + 64) * (8.0 * ((N + 63) / 64) + 20.0) * 5.0 / 2.4e6;
        const double window_ms = (double)a->nreads * N * 5.85e-8;
        const double call_ms = (double)a->nreads * (double)maxlen * N * 7.4e-11;
        const double exposed = audit_ms - window_ms;
        if (exposed > 0.02 * call_ms) {
            double kd = exposed / (0.02 * call_ms + 1e-9);
            const uint32_t K = kd > 64.0 ? 64u : (uint32_t)kd + 1u;
            if (c->dtw_sparse_calls++ % K) naudit = 0;
        }
    }
    int32_t *alist = nullptr;  sk_hit *aout = nullptr;
    if (naudit) {
        if ((rc = sk_reserve(c, &c->audit, ((size_t)naudit + 2) * sizeof(int32_t) + 16 + (size_t)naudit * sizeof(sk_hit)))) return rc;
        alist = (int32_t *)c->audit.p;
        aout = (sk_hit *)((char *)c->audit.p + ((((size_t)naudit + 2) * sizeof(int32_t) + 15) & ~(size_t)15));
    }
    auto launch_audit = [&](hipStream_t stream) -> int {    // the exact pass over the audit reads (records by slot)
        if (!naudit) return SK_OK;
        sk_audit_pick(alist, naudit, audit_period, a->nreads, c->dtw_audit_calls++, stream ? stream : c->stream);
        SK_HIP(hipGetLastError());
        sdtw_kargs kr = k;
        kr.read0 = 0; kr.ridx = alist + 2; kr.count_ptr = alist; kr.ckpt = nullptr; kr.out = aout; kr.out_by_slot = 1;
        kr.total_ptr = guard + SK_GUARD_AUDITED; kr.list_off = 0;
        const int R64 = (N + 63) / 64, P64 = 64 * R64 - N;
        sdtw_fn f64 = (L == 16) ? pick_any(a->feed, 64, R64, MODE_FULL) : nullptr;
        if (f64) {                                          // a short list: one read per wavefront, as the retry
            sdtw_kargs k64 = kr;
            k64.xlay = (const double *)c->motif64.p; k64.P = P64;
            k64.nreads = naudit < 8192 ? naudit : 8192;
            int rc2;
            if ((rc2 = launch(c, f64, k64, 64, stream))) return rc2;
            if (naudit <= 8192) return SK_OK;
            kr.list_off = 8192; kr.total_ptr = nullptr; kr.nreads = naudit - 8192;
            return launch(c, ff, kr, L, stream);
        }
        kr.nreads = naudit;
        return launch(c, ff, kr, L, stream);
    };
    auto finish_guard = [&]() -> int {                      // on the main stream, behind every writer of out[]
        if (!guarded) return SK_OK;
        if (naudit) {
            sk_audit_compare(alist, aout, a->out, naudit, guard, c->stream);
            SK_HIP(hipGetLastError());
        }
        sdtw_kargs kf = k;                                  // the gated exact pass over the whole call
        kf.read0 = 0; kf.nreads = a->nreads; kf.ridx = nullptr; kf.count_ptr = nullptr; kf.ckpt = nullptr;
        kf.gate_ptr = guard + SK_GUARD_ALARM; kf.guard = guard;
        return launch(c, ff, kf, L);
    };
    if (qok) {
        // The reads pass Q itself finds unscreenable (on the C4 batch: all of the ~190 that retry) get their exact
        // pass on a third stream as soon as pass Q is done -- one sweep's latency (0.5 ms) that then runs beside the
        // window passes instead of behind them; what the window passes give up on follows as before.
        const bool early = sk_tune("SK_DTW_NO_EARLY") == nullptr;
        if (early && !c->stream3) {
            SK_HIP(hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
            for (int i = 0; i < 2; i++) SK_HIP(hipEventCreateWithFlags(&c->ev_r[i], hipEventDisableTiming));
        }
        if (early && naudit && !c->stream4) {
            SK_HIP(hipStreamCreateWithFlags(&c->stream4, hipStreamNonBlocking));
            SK_HIP(hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming));
        }
        if ((rc = sk_launch_sdtw_screen(c, a, ck, span_q, span2, cnt, cnt + 2, early ? ecnt : nullptr,
                                        early ? ecnt + 2 : nullptr))) return rc;
        if (early) {
            SK_HIP(hipStreamWaitEvent(c->stream3, c->ev_r[0], 0));
            if ((rc = launch_retry(ecnt, c->stream3))) return rc;
            SK_HIP(hipEventRecord(c->ev_r[1], c->stream3));
            // the audit's sweep beside both of them, on a stream of its own: it is one sweep's latency (0.6 ms) like the
            // early retry, and behind it on the same stream it outlasted the window passes of calls below 100 000 reads
            // (31 250 reads: 3.2 instead of 2.7 ms)
            if (naudit) {
                SK_HIP(hipStreamWaitEvent(c->stream4, c->ev_r[0], 0));
                if ((rc = launch_audit(c->stream4))) return rc;
                SK_HIP(hipEventRecord(c->ev_a, c->stream4));
            }
        }
        if ((rc = launch_retry())) return rc;
        if (early) {
            SK_HIP(hipStreamWaitEvent(c->stream, c->ev_r[1], 0));
            if (naudit) SK_HIP(hipStreamWaitEvent(c->stream, c->ev_a, 0));
        } else if ((rc = launch_audit(nullptr))) return rc;
        if ((rc = finish_guard())) return rc;
        SK_HIP(hipEventRecord(c->ev[3], c->stream));
        return SK_OK;
    }
    const int nck = (int)((maxlen + L - 1) / ck);          // checkpoints at steps ck, 2ck, ... <= last step
    const size_t per_read = (size_t)(nck > 0 ? nck : 1) * L * (R + 3) * sizeof(double);
    int64_t chunk;                                         // checkpoint scratch per chunk
    while (true) {
        chunk = sk_dtw_chunk_reads(per_read, a->nreads);
        rc = sk_reserve(c, &c->ckpt, (size_t)chunk * per_read);
        if (rc != SK_ERR_NOMEM || chunk <= 1024) break;
        sk_dtw_scratch_shrink(0);
    }
    if (rc) return rc;
    k.ckpt = (double *)c->ckpt.p; k.nck = nck; k.ck = ck; k.span = span;
    k.retry = cnt + 2; k.retry_cnt = cnt;
    // per-launch HIP events (pool grows on demand) so a profile can name each pass's duration
    const size_t nchunks = (size_t)((a->nreads + chunk - 1) / chunk);
    while (c->evpool.size() < 3 * nchunks) {
        hipEvent_t e;
        SK_HIP(hipEventCreate(&e));
        c->evpool.push_back(e);
    }
    c->prof_chunks = 0;
    for (int64_t r0 = 0; r0 < a->nreads; r0 += chunk) {
        k.read0 = (int)r0;
        k.nreads = (int)((a->nreads - r0 < chunk) ? a->nreads - r0 : chunk);
        hipEvent_t *ev = &c->evpool[3 * (size_t)c->prof_chunks];
        SK_HIP(hipEventRecord(ev[0], c->stream));
        if ((rc = launch(c, fa, k, L))) return rc;         // pass A: dist, end, checkpoints
        SK_HIP(hipEventRecord(ev[1], c->stream));
        if ((rc = launch(c, fb, k, L))) return rc;         // pass B: start from the nearest checkpoint
        SK_HIP(hipEventRecord(ev[2], c->stream));
        c->prof_reads[c->prof_chunks < 64 ? c->prof_chunks : 63] = k.nreads;
        c->prof_chunks++;
    }
    // reads whose path crossed the restart front: exact single pass on just those
    if ((rc = launch_retry())) return rc;
    SK_HIP(hipEventRecord(c->ev[3], c->stream));
    return SK_OK;
}
