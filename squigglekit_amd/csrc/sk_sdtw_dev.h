// sk_sdtw_dev.h -- device helpers and the kernel argument block shared by the DTW kernels
// (sk_sdtw.hip: exact FP64 passes; sk_sdtwq.hip: fixed-point screening pass + certified window).
#pragma once
#include "sk_common.h"

namespace {

constexpr int DPP_ROW_SHR1  = 0x111;   // lane i <- lane i-1 inside a row of 16; lane 0 keeps `old`
constexpr int DPP_ROW_ROL1  = 0x12F;   // row_ror:15 == rotate left by one inside a row of 16
constexpr int DPP_WAVE_SHR1 = 0x138;   // lane i <- lane i-1 across the wave; lane 0 keeps `old`
constexpr int DPP_ROW_SHL1  = 0x101;   // lane i <- lane i+1 inside a row of 16; lane 15 keeps `old`
constexpr int DPP_WAVE_SHL1 = 0x130;   // lane i <- lane i+1 across the wave; lane 63 keeps `old`
constexpr int DPP_WAVE_ROL1 = 0x134;   // lane i <- lane i+1 across the wave (rotate)

enum { MODE_FULL = 0, MODE_DIST = 1, MODE_START = 2, MODE_CHAIN = 3 };   // CHAIN: FULL on one row chunk of a long motif

// Fixed-point screening (sk_sdtwq.hip): one unit = 2^-22 of a normalised signal unit.
constexpr int      QS     = 22;
constexpr double   QSCALE = 4194304.0;          // 2^22
constexpr double   QUNIT  = 1.0 / 4194304.0;
constexpr double   QLIM   = 400.0;              // |x|, |y| must stay below this (400 * 2^22 < 2^31)
constexpr unsigned QINF   = 0xFFFFFFFFu;        // "+inf" cost, and the sample fed outside [0, n)
constexpr unsigned QSAFE  = 0xF0000000u;        // a minimum at or above this may have saturated

// Guard counters of one DTW call (device, ints 8.. of sk_ctx::dtwcnt; sk_last_dtw_guard()):
enum { SK_GUARD_VIOL = 0,       // window pass: an accepted result contradicted the screening values it rests on
       SK_GUARD_AUDITED = 1,    // reads the audit re-ran with the exact single pass
       SK_GUARD_MISMATCH = 2,   // ... whose record differed from the screening scheme's (the exact record then wins)
       SK_GUARD_IMGREJ = 3,     // reads whose sample image cannot be bounded tightly enough: exact pass, by design
       SK_GUARD_ALARM = 4,      // VIOL + MISMATCH: non-zero opens the gate of the whole-call exact fallback
       SK_GUARD_FELLBACK = 5,   // the whole call was redone by the exact single pass
       SK_GUARD_WORDS = 8 };
enum { SK_HOLE_NONE = 0, SK_HOLE_QERR1 = 1, SK_HOLE_FMA64 = 2, SK_HOLE_FMA64_UNGUARDED = 3 };

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

// min of two non-NaN doubles as ONE v_min_f64 (fmin() would add canonicalising ops in IEEE mode)
__device__ __forceinline__ double vmin(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

struct sdtw_kargs {
    const void    *samples;     // int16 or double samples (filtered)
    const void    *samples_raw; // float64 feeds: unfiltered input, read for reads flagged SK_IFLAG_INPLACE (or nullptr)
    int64_t        stride;      // row stride for FEED_I16
    const int64_t *off;         // ragged offsets for the f64 feeds
    const sk_prep *prep;        // n / center / scale per read (not for F64_RAW)
    int            nreads;      // reads (or entries of ridx) covered by this launch
    int            read0;       // first read of this launch (chunking); checkpoint slot = r - read0
    const int32_t *ridx;        // optional indirection: launch slot -> read (retry pass)
    const int32_t *count_ptr;   // with ridx: the list length lives on the device (nreads = capacity of the launch)
    int            list_off;    //            first list entry this launch covers
    int32_t       *total_ptr;   //            optional: += list length (diagnostic, read by sk_last_dtw_retries)
    const double  *xlay;        // motif laid out per lane [L][R]
    int            P;           // number of short lanes
    sk_hit        *out;
    double        *last_row;    // FULL only: cost[-1, :] of read 0
    double        *ckpt;        // exact scheme: [slot][nck][L][R+3] doubles
    int            nck;         // checkpoints per read
    int            ck;          // steps between checkpoints (multiple of L)
    int            span;        // START: look-back in columns
    int32_t       *retry;       // START: reads whose result could not be certified
    int32_t       *retry_cnt;
    // fixed-point screening scheme
    const unsigned *xlayq;      // motif, quantised + biased, laid out per lane [L][R]
    unsigned      *ckq;         // [slot][nck][L][(R+3)/2]: the high halves of the R + 2 state words, two to a dword
    unsigned      *lastq;       // [slot][lq_stride]: screening cost of the last row per column
    int64_t        lq_stride;
    int32_t       *qflag;       // [slot]: screening minimum of the read; QINF = a sample left the fixed-point range
    unsigned       qerr;        // E: bound (in units) on |screening cost - exact cost| of any cell
    int            wmax;        // widest candidate-column range the window pass accepts
    unsigned      *lsum;        // [slot][nck+1][L]: minimum of the last-row columns a lane stored per checkpoint interval
    unsigned      *wstate;      // [slot][L][R+2]: restart state of the window pass (pass P -> pass W)
    void          *wrec;        // [slot] {tbase, jlo, jhi, flags}: pass P -> pass W
    void          *wrec_q;      // [read - read0] {-, jlo, jhi, -}: pass Q's epilogue -> the first tier of pass P
    int32_t        tier2;       // pass P: 1 = look for the candidate columns again (second tier) instead of taking pass Q's;
                                // 2 = the launch covers the sibling list (a read's second cluster of candidate columns)
    void          *sib;         // [chunk] {read, jlo, jhi, -}: second clusters, appended by pass Q's epilogue (nullptr: none)
    int32_t       *sib_cnt;     //         their number (device; may run past sib_cap: entries beyond it were not stored)
    int            sib_cap;     //         capacity of the list
    sk_hit        *sib_out;     // [chunk] pass W's result per sibling slot (n = 1: certified)
    // pass Q with the filter + medmad statistics fused in as a prologue (int16 reads): the wave preps its own reads
    const int16_t *fz_raw;      // raw rows (same stride), or nullptr: prep / samples were filled by an earlier kernel
    const int32_t *fz_len;
    int            fz_lo, fz_hi, fz_vec;
    int            fz_mode;     // 0: medmad (histogram median + MAD), 1: zscale (numpy-order mean / std; rows <= 4 096 samples)
    int            lds_wave_words; // pass Q: words of dynamic LDS per wavefront (the prologue's histogram, then the interval's last-row values)
    int32_t       *early;       // reads pass Q already knows cannot be screened (candidate range too wide, samples out of
    int32_t       *early_cnt;   // range): their exact retry starts right behind pass Q, beside the window passes
    unsigned long long *wsteps; // pass W: [0] += steps this wavefront ran (all its read groups in lockstep), [1] += steps its
                                // reads asked for (sum over the groups); or nullptr
    unsigned long long *clk;    // pass Q: {shader cycles, 100 MHz reference ticks} of the first wave's sweep, or nullptr
    int            force_retry; // sensitivity runs: reads whose hash (10 bits) is below this take the exact retry
    // run-time guard of the screening certificate: counters in device memory, see SK_GUARD_*
    int32_t       *guard;       // [SK_GUARD_WORDS] or nullptr
    int            hole;        // tests only (SK_DTW_HOLE): SK_HOLE_* -- a known precision hole re-introduced on purpose
    int            out_by_slot; // exact kernel: the record of launch slot s goes to out[s], not out[read] (audit pass)
    const int32_t *gate_ptr;    // exact kernel: the whole launch returns at once unless *gate_ptr != 0 (whole-call fallback)
    // window pass, second tier: the reads of one chunk whose path crossed the first (short) look-back
    const int32_t *wl_list;     // reads to process (nullptr: all of the chunk); wl_count: their number (device)
    const int32_t *wl_count;
    int32_t       *soft;        // where a read goes whose path crossed THIS look-back (nullptr: the exact retry list)
    int32_t       *soft_cnt;
    // row-chunked motifs (MODE_CHAIN): the last row of the chunk above / of this chunk, per column
    const double  *prevD;       // [slot][row_stride] or nullptr (first chunk: virtual row -1)
    const int32_t *prevS;
    double        *rowD;        // [slot][row_stride] or nullptr (last chunk)
    int32_t       *rowS;
    int64_t        row_stride;
};

} // namespace
