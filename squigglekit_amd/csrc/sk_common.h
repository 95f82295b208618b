// sk_common.h -- internal declarations shared by the HIP translation units.
// gfx950 only.  Not part of the public ABI (that is include/squigglekit_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdarg.h>
#include <vector>
#include "squigglekit_hip.h"

#define SK_MAX_DEVICES 16
#define SK_WAVE 64

// Per-read result of the prep kernel; consumed by the DTW / segment kernels.
// 48 bytes, one per read, lives in HBM.
struct sk_prep {
    int32_t n;        // samples surviving scale_outliers
    int32_t flags;    // SK_FLAG_*
    double  center;   // medmad: median            zscale: mean       segmenter: median
    double  scale;    // medmad: MAD*1.4826        zscale: std (0->1) segmenter: std
    double  top;      // segmenter: median + std*std_scale   (segmenter.py:413)
    double  bot;      // segmenter: median - std*std_scale   (segmenter.py:414)
};

// Internal flag bits of sk_prep::flags (never reach sk_hit::flags: the DTW kernels keep the low byte only).
// INPLACE: nothing of this float64 read was dropped by the filter, so its filtered samples were not copied -- the DTW
// feed reads them from the caller's buffer (sk_sdtw_args::samples_raw) at the same offsets.
#define SK_IFLAG_INPLACE 0x100
#define SK_FLAG_PUBLIC   0xff

enum sk_prep_mode { SK_PREP_MEDMAD = 0, SK_PREP_ZSCALE = 1, SK_PREP_SEGMENT = 2, SK_PREP_DRNA = 3 };

// Growable device scratch buffer.
struct sk_buf {
    void  *p = nullptr;
    size_t cap = 0;
};

struct sk_ctx {
    int         device = -1;
    bool        ready = false;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;    // second stream (created on first use): overlapped walks / copies
    hipStream_t stream3 = nullptr;    // third stream (created on first use): the early exact retry beside the window passes
    hipEvent_t  ev_r[2] = {nullptr, nullptr};   // its ordering events: pass Q done / early retry done
    hipStream_t stream4 = nullptr;    // fourth stream (created on first use): the audit's exact sweep beside the window passes
    hipEvent_t  ev_a = nullptr;       // audit sweep done
    hipEvent_t  ev_s[2] = {nullptr, nullptr};   // second clusters (siblings) on the third stream: pass Q done / their round done
    uint32_t    dtw_sparse_calls = 0; // calls whose audit would be exposed (few, long reads): every K-th is audited
    uint32_t    dtw_audit_calls = 0;  // audited launches so far: salt of the audit's choice of reads (k_audit_pick)
    hipEvent_t  ev_chunk[9] = {};     // ordering events between the two streams (no timing)
    hipEvent_t  ev[4] = {nullptr, nullptr, nullptr, nullptr};   // prep start/stop, main start/stop
    bool        ev_valid = false;
    int         num_cu = 0;
    // scratch (grown on demand, reused across calls)
    sk_buf sig;       // staged input samples
    sk_buf len;       // int32 per read
    sk_buf rlen;      // int32 per read: the caller's per-read cut of a ragged float64 batch (sk_segment_batch_f64_len)
    sk_buf off;       // int64 per read (+1) for ragged f64
    sk_buf comp;      // compacted samples
    sk_buf prep;      // sk_prep per read
    sk_buf mask;      // in-band bit masks (segmenter)
    sk_buf motif;     // laid-out motif rows
    sk_buf out;       // sk_hit / segs staging
    sk_buf out2;      // nsegs staging
    sk_buf misc;
    sk_buf seghints;  // per read: the stretches of quiet mask entries and their anchors, k_seg_stats -> k_seg_walk4
    sk_buf pacal;     // per read {offset, range / digitisation}: the pA conversion of raw rows (sk_segment_batch_i16_pa)
    sk_buf ckpt;      // DTW checkpoints (systolic state dumps: doubles or fixed-point units)
    sk_buf motifq;    // fixed-point motif layout
    sk_buf motif64;   // the motif laid out for 64 lanes (retry pass of a short motif)
    bool   motif64_valid = false;
    std::vector<double> motif64_host;
    sk_buf lastq;     // screening pass: last-row costs per column
    sk_buf qflag;     // screening pass: per-read "left the fixed-point range" flags
    sk_buf wsoft;     // window pass: reads of the chunk that go to its second tier ([0] = count)
    sk_buf lsum;      // screening pass: last-row minima per checkpoint interval and lane
    sk_buf wstate;    // window pass: restart state per read (pass P -> pass W)
    sk_buf wrec;      // window pass: restart point per read {tbase, jlo, jhi, flags}
    sk_buf wrecq;     // screening pass epilogue: candidate columns per read (first tier of pass P)
    sk_buf order;     // [1024] sort cursors, then the chunk's reads in the order the window passes take them
    sk_buf motifw;    // the motif (doubles) in the screening scheme's own per-lane layout
    std::vector<double> motifw_host;
    int    motifq_L = 0;   // lanes per read of the layouts in motifq / motifw
    sk_buf retry;     // DTW retry list: [0] = count, [1] = pad, [2 ..] = reads (segmenter: [0] = count, [1 ..] = reads)
    sk_buf dtwcnt;    // [0] = reads retried by the exact pass, summed over the launches of one API call (device); [1] second tier; +16 clock; +32 guard counters
    sk_buf audit;     // the audit's read list ([0] = count, [2 ..] = reads) and its exact records
    sk_buf sib;       // window passes: [0..3] count, then {read, jlo, jhi, -} per second cluster of candidate columns
    sk_buf sibout;    // ... and pass W's record per sibling
    sk_buf sibstate;  // ... their own pass-P state and window records (the round runs beside the main window passes)
    bool   retry_dev = false;   // the last DTW call left its retry count on the device (read lazily)
    std::vector<unsigned> motifq_host;
    bool   motifq_valid = false;
    int    f64_stream = 0;   // the last float64 call took the streaming statistics kernel (its retry count: retry[0])
    int    pa_raw = 0;       // the last pA call of raw rows stayed in the raw domain (k_seg_stats<.., PA>)
    std::vector<const int32_t *> pa_retry_ptrs;   // ... the device counters of its numpy-order redo lists (one per chunk)
    std::vector<int64_t> pa_off_host;             // slot offsets of the float64 fallback of that route
    int    last_retry = 0;   // reads that needed the exact single-pass retry in the last DTW call
    std::vector<hipEvent_t> evpool;   // per-launch events of the two-pass DTW (3 per chunk)
    int    prof_chunks = 0;  // chunks of the last two-pass DTW call (0: single pass)
    int    prof_reads[64] = {0};
    std::vector<double> motif_host;   // last laid-out motif (kept alive for async H2D)
    std::vector<double> motif_src;    // the motif it was built from (upload cache key, with motif_L)
    int    motif_L = 0;               // lanes per read of the layout in `motif`
    void  *comm = nullptr;            // ncclComm_t of this device (sk_comm.hip), or nullptr
    sk_buf commbuf;                   // staging for the small host-side exchanges
};

// ---- tuning switches (sk_runtime.hip) ----
// Every environment switch the library reads is listed in ONE table (SK_TUNABLES in sk_runtime.hip, exported as text by
// sk_tunables()); none changes results.  They are read only when SK_TUNING=1 is set as well (tests, tools/, bench.py
// set it): a stray SK_* variable in a user's shell cannot move a production job onto a slow or rarely used path.
// sk_tune() returns the variable's value, or nullptr when it is unset, tuning is off, or the name is not in the table.
const char *sk_tune(const char *name);

// ---- runtime (sk_runtime.hip) ----
// Every entry point that touches a context holds that context's lock from sk_cur() to its return (SURVEY 8(b):
// "per-device context guarded by a mutex"): two host threads bound to the same slot take turns call by call -- the
// scratch buffers sk_reserve may free, the streams' event slots and the "last call" counters belong to one call at a
// time.  Recursive: entry points call each other.  Contexts of different slots never wait for each other.
void sk_ctx_lock(sk_ctx *c);
void sk_ctx_unlock(sk_ctx *c);
struct sk_ctx_guard {
    sk_ctx *c;
    explicit sk_ctx_guard(sk_ctx *ctx) : c(ctx) { if (c) sk_ctx_lock(c); }
    ~sk_ctx_guard() { if (c) sk_ctx_unlock(c); }
    sk_ctx_guard(const sk_ctx_guard &) = delete;
    sk_ctx_guard &operator=(const sk_ctx_guard &) = delete;
};
sk_ctx *sk_cur(void);                       // bound context or nullptr (error set)
sk_ctx *sk_ctx_of(int device);              // context slot of a device (ready or not)
int  sk_bound_device(void);                 // device the calling thread is bound to, or -1
int  sk_fail(int code, const char *fmt, ...);
int  sk_reserve(sk_ctx *c, sk_buf *b, size_t bytes);
// Reads per chunk of a checkpointing DTW call: scratch budget (64 GB, or SK_DTW_SCRATCH_MB) / per_read, at
// least 1024 reads (64 when the budget was set by hand, so that tests can force many small chunks).
int64_t sk_dtw_chunk_reads(size_t per_read, int64_t nreads);
void    sk_dtw_scratch_shrink(int reset);   // halve the calling thread's scratch budget (reset != 0: back to the default)
#define SK_HIP(call)                                                                    \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess)                                                           \
            return sk_fail(SK_ERR_HIP, "%s failed: %s (%s:%d)", #call,                 \
                           hipGetErrorString(e_), __FILE__, __LINE__);                  \
    } while (0)

// ---- prep (sk_prep.hip) ----
// i16: rows of `stride` samples; comp gets the filtered samples of read r at
// comp + r*stride.  mask (segmenter only) gets ceil(stride/64) words per read.
int sk_launch_prep_i16(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len,
                       int32_t nreads, int32_t lo, int32_t hi, int mode, double std_scale,
                       int16_t *d_comp, sk_prep *d_prep, uint64_t *d_mask, int64_t mask_stride,
                       int32_t t0 = 0, int32_t t1 = 0x7fffffff,    // statistics window (filtered index)
                       // SEGMENT mode over a device-side list of reads (d_list[0 .. *d_count)): statistics in numpy's
                       // order, masks written as {in band, kept} entries in raw coordinates (sk_segstat.hip)
                       const int32_t *d_list = nullptr, const int32_t *d_count = nullptr,
                       void *d_mask2 = nullptr, int row16 = 0);
// wavefront-per-read medmad variant (sk_prepw.hip): same results; returns 1 (nothing launched) when
// the configuration is outside its range and the caller has to use the workgroup-per-read kernel
int sk_launch_prepw_medmad(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len,
                           int32_t nreads, int32_t lo, int32_t hi, int16_t *d_comp, sk_prep *d_prep);
// f64 ragged: read r is sig[off[r]..off[r+1]); comp uses the same offsets.
int sk_launch_prep_f64(sk_ctx *c, const double *d_sig, const int64_t *d_off, int32_t nreads,
                       double lo, double hi, int mode, double std_scale,
                       double *d_comp, sk_prep *d_prep, uint64_t *d_mask, int64_t mask_rows,
                       const int32_t *d_rlen = nullptr);   // optional: read r is its first d_rlen[r] samples

// ---- DTW (sk_sdtw.hip) ----
// Input kinds for the sample feed.
enum sk_feed { SK_FEED_I16 = 0,      // int16 filtered samples + (center, scale) from sk_prep
               SK_FEED_F64_NORM = 1, // float64 filtered samples + (center, scale) from sk_prep
               SK_FEED_F64_RAW = 2 };// float64 already normalised (mlpy boundary), ragged
// int16 MotifSeq calls with medmad: filter + statistics can run as the prologue of the screening pass
// (sk_sdtwq.hip) instead of as a kernel of their own; sk_launch_sdtw falls back to the kernel when its
// screening scheme does not apply to the call.
struct sk_prep_fuse {
    const int16_t *raw = nullptr;   // device, rows of `stride` samples (sk_sdtw_args::stride)
    const int32_t *len = nullptr;   // device
    int32_t lo = 0, hi = 0;         // scale_outliers limits
    int     mode = SK_PREP_MEDMAD;  // SK_PREP_MEDMAD or SK_PREP_ZSCALE (round 5: reads of up to 4 096 samples)
};
// can filter + statistics of this call ride in the screening pass?  medmad: limits inside the prologue's histogram
// range; zscale: rows of at most 4 096 samples (the wave keeps the compacted read in LDS)
bool sk_sdtw_fuse_ok(int32_t lo, int32_t hi, int mode = SK_PREP_MEDMAD, int64_t stride = 0);

struct sk_sdtw_args {
    int           feed;
    const void   *samples;     // int16* or double*
    int64_t       stride;      // row stride (feed I16), ignored when off != nullptr
    const int64_t *off;        // ragged offsets (nreads+1) or nullptr
    const sk_prep *prep;       // per-read n/center/scale (nullptr for F64_RAW: n from off)
    int32_t       nreads;
    const double *motif;       // HOST pointer, nmotif points
    int32_t       nmotif;
    sk_hit       *out;         // device, nreads records
    double       *last_row;    // device, optional: cost[-1,:] of read 0 (single-pair call)
    int64_t       max_len;     // upper bound of any read's filtered length (chooses 1 vs 2 passes)
    int           force_single;// 1: always the single FULL pass
    int           accumulate = 0;  // 1: a further launch set of the same API call (keep the retry total)
    const void   *samples_raw = nullptr;  // float64 feeds: the unfiltered input (same offsets), read instead of `samples`
                                          // for reads flagged SK_IFLAG_INPLACE
    const sk_prep_fuse *fuse = nullptr;   // non-null: samples / prep are NOT filled yet (see sk_prep_fuse); they are
                                          // the writable c->comp / c->prep buffers
};
int sk_launch_sdtw(sk_ctx *c, const sk_sdtw_args *a);
// fixed-point screening + certified window over all reads (sk_sdtwq.hip); leaves the retry list on the device
int sk_launch_sdtw_screen(sk_ctx *c, const sk_sdtw_args *a, int ck, int span, int span2,
                          int32_t *d_retry_cnt, int32_t *d_retry, int32_t *d_early_cnt, int32_t *d_early);

// ---- dRNA --signal branch (rolling mean): statistics + masks (sk_prep.hip), scan (sk_segment.hip) ----
struct sk_roll_params;
// comp / prep as left by sk_launch_prep_i16; psum: nreads * (stride + 1) int64 scratch; masks: two
// transposed bit masks (t < bot, t > bot), `words` words per read each, word wi of read r at [wi * nreads + r]
bool sk_roll_stream_ok(int64_t stride, int32_t w, int32_t lo, int32_t hi);   // can the streaming kernel take these rows?
int sk_launch_roll_stream(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads, int32_t lo,
                          int32_t hi, int32_t w, double std_scale, sk_prep *d_prep, uint64_t *d_below, uint64_t *d_above,
                          int32_t *d_redo /* [nreads + 2] */);
size_t sk_roll_one_lds(int64_t stride, int32_t w);      // LDS of the one-look kernel for these rows (0: they do not fit)
int sk_launch_roll_one(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads, int32_t lo,
                       int32_t hi, int32_t w, double std_scale, sk_prep *d_prep, uint64_t *d_below, uint64_t *d_above);
int sk_launch_roll_stats(sk_ctx *c, const int16_t *d_comp, int64_t stride, sk_prep *d_prep, int32_t nreads,
                         int32_t w, double std_scale, int64_t *d_psum, uint64_t *d_below, uint64_t *d_above);
int sk_launch_roll_walk(sk_ctx *c, const uint64_t *d_below, const uint64_t *d_above, const sk_prep *d_prep,
                        int32_t nreads, const sk_roll_params *p, int32_t *d_xy, int32_t *d_found, int64_t row_words = 0);

// ---- segmenter, streaming path (sk_segstat.hip) ----
int  sk_segment_fast_row16(int64_t stride);
bool sk_segment_fast_applies(const void *d_sig, int64_t stride, int32_t lo, int32_t hi, double std_scale);
// d_cal != nullptr: the pA route in the raw domain (round 6) -- [nreads][2] = {offset, range / digitisation}; lo / hi are
// then the limits in pA and d_scratch holds num_cu rows of `stride` doubles for the numpy-order redo
int  sk_launch_segment_fast(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads,
                            const sk_seg_params *p, int32_t lo, int32_t hi, sk_prep *d_prep, void *d_mask2,
                            int32_t *d_retry, int32_t *d_segs, int32_t *d_nsegs, int32_t max_segs,
                            const double *d_cal = nullptr, double *d_scratch = nullptr);
bool sk_segment_pa_applies(const void *d_sig, int64_t stride, double std_scale);
int  sk_launch_prep_pa_listed(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len, const double *d_cal,
                              const int32_t *d_list, const int32_t *d_count, int grid, double lo, double hi, double std_scale,
                              double *d_scratch, int64_t scratch_stride, sk_prep *d_prep, void *d_mask2, int row16);

// ---- segment walk (sk_segment.hip) ----
struct sk_drna_params;
int sk_launch_drna_walk(sk_ctx *c, const uint64_t *d_mask, int64_t mask_rows, const sk_prep *d_prep,
                        int32_t nreads, const sk_drna_params *p, int32_t *d_segs, int32_t *d_nsegs,
                        int32_t max_segs);
int sk_launch_segment_walk(sk_ctx *c, const uint64_t *d_mask, int64_t mask_stride,
                           const int64_t *d_mask_off, const sk_prep *d_prep, int32_t nreads,
                           const sk_seg_params *p, int32_t *d_segs, int32_t *d_nsegs,
                           int32_t max_segs);

// ---- synth (sk_synth.hip) ----
int sk_launch_synth(sk_ctx *c, int16_t *d_sig, int64_t stride, int32_t nreads, int32_t nsamples,
                    uint64_t seed, const int16_t *d_motif_i16, int32_t nmotif, int64_t row0 = 0,
                    int hit_pm = 500, int stretch_pm = 0, int stretch = 1);
int sk_launch_synth_windows(sk_ctx *c, int16_t *d_sig, int64_t stride, int32_t nreads, int32_t nsamples,
                            uint64_t seed, int64_t row0, const int16_t *d_tmpl, int32_t ntmpl, float sigma);
int sk_launch_raw_to_pa(sk_ctx *c, const int16_t *d_sig, int64_t stride, int32_t nreads, int32_t nsamples,
                        double offset, double raw_unit, double *d_out, int64_t *d_off);
int sk_launch_centi_to_f64(sk_ctx *c, const int32_t *d_centi, int64_t total, double *d_out);
// ragged form with per-read constants: read r's len = off[r+1] - off[r] samples, cal[2r] = offset, cal[2r+1] = raw unit
int sk_launch_rows_to_pa(sk_ctx *c, const int16_t *d_sig, int64_t stride, int32_t nreads, const int64_t *d_off,
                         const double *d_cal, double *d_out);

// ---- float64 reads, streaming statistics (sk_f64stat.hip) + the numpy-order redo of its uncertified reads ----
bool sk_f64_fast_applies(int64_t maxlen, double std_scale);
int  sk_f64_row16(int64_t maxlen);
int  sk_launch_f64_stats(sk_ctx *c, const double *d_sig, const int64_t *d_off, const int32_t *d_rlen, int32_t nreads, int64_t maxlen,
                         double lo, double hi, int mode, double std_scale, sk_prep *d_prep, void *d_mask2, int row16,
                         int32_t *d_len, int32_t *d_retry, double *d_comp);
int  sk_launch_prep_f64_listed(sk_ctx *c, const double *d_sig, const int64_t *d_off, const int32_t *d_list,
                               const int32_t *d_count, int grid, double lo, double hi, int mode, double std_scale,
                               double *d_comp_or_scratch, int64_t scratch_stride, sk_prep *d_prep, void *d_mask2,
                               int row16, const int32_t *d_rlen = nullptr);
int  sk_launch_seg_walk_masks(sk_ctx *c, const void *d_mask2, int row16, const int32_t *d_len, int32_t nreads,
                              const sk_seg_params *p, int32_t *d_segs, int32_t *d_nsegs, int32_t max_segs);
