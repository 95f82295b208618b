/*
 * squigglekit_hip.h -- C ABI of libsquigglekit_hip.so (MI355X / gfx950).
 *
 * The reference (Psy-Fer/SquiggleKit) has no FFI: its hot path sits behind two
 * in-process Python call boundaries,
 *     segmenter.get_segs(sig, args)          /root/reference/segmenter.py:399
 *     mlpy.dtw_subsequence(model, sig)       /root/reference/MotifSeq.py:437
 * wrapped by per-read loops (segmenter.py:189-230, MotifSeq.py:261-298) that
 * filter (scale_outliers) and normalise each read first.  This header is what
 * a ctypes binding at those two call sites binds instead (INTEGRATION.md shows
 * the stub).  Batch entry points take many reads per call because one read is
 * far too little work for a GPU; the single-pair entry points keep the
 * reference's one-call-per-read shape.
 *
 * Conventions
 *   - plain pointers and sizes only; caller owns every buffer; the library
 *     keeps no caller pointer after a call returns.
 *   - every function returns SK_OK (0) or a negative sk_status; the message is
 *     available from sk_last_error() (thread local).
 *   - "host" entry points take host pointers and do H2D / kernels / D2H;
 *     "_dev" entry points take device pointers obtained from sk_dev_alloc()
 *     and leave results in HBM (what bench.py times).
 *   - no CPU fallback exists: without a usable HIP device every compute entry
 *     point fails with SK_ERR_NO_DEVICE.
 */
#ifndef SQUIGGLEKIT_HIP_H
#define SQUIGGLEKIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sk_status {
    SK_OK              =  0,
    SK_ERR_INVALID     = -1,   /* bad argument (NULL, negative size, bad params)        */
    SK_ERR_NO_DEVICE   = -2,   /* no HIP device / sk_init not called / device lost      */
    SK_ERR_HIP         = -3,   /* a HIP runtime call failed (message has the detail)    */
    SK_ERR_NOMEM       = -4,   /* device or host allocation failed                      */
    SK_ERR_UNSUPPORTED = -5,   /* shape outside what the kernels cover (message says)   */
    SK_ERR_OVERFLOW    = -6    /* max_segs too small for at least one read              */
} sk_status;

/* scale modes of MotifSeq.py -l/--scale (MotifSeq.py:96) */
enum { SK_SCALE_MEDMAD = 0, SK_SCALE_ZSCALE = 1 };

/* flags in sk_hit.flags */
enum {
    SK_FLAG_EMPTY      = 1,    /* no sample survived scale_outliers: dist = NaN, start=end=-1 */
    SK_FLAG_DEGENERATE = 2,    /* medmad with MAD == 0 (reference divides by zero)            */
    SK_FLAG_RECENTRE   = 4     /* float64 zscale: sklearn.preprocessing.scale subtracted the residual mean a
                                  second time ("mean not close to zero", near-constant or huge-valued
                                  data, MotifSeq.py:187-191); applied here too -- informational        */
};

/* get_segs parameters: the argparse flags of segmenter.py:65-96 that reach
 * scale_outliers (segmenter.py:311-318) and get_segs (segmenter.py:399-470). */
typedef struct sk_seg_params {
    int32_t error;       /* -e/--error      default 5    */
    int32_t corrector;   /* -c/--corrector  default 50   */
    int32_t window;      /* -w/--window     default 150  */
    int32_t seg_dist;    /* -d/--seg_dist   default 50   */
    double  std_scale;   /* -t/--std_scale  default 0.75 */
    double  stall_len;   /* -l/--stall_len  default 0.25 */
    int32_t lim_low;     /* -lim_low        default 0    */
    int32_t lim_hi;      /* -lim_hi         default 900  */
} sk_seg_params;

/* One MotifSeq hit: what get_region_multi (MotifSeq.py:431-449) derives from
 * dist, path[1][0], path[1][-1].  24 bytes. */
typedef struct sk_hit {
    double  dist;        /* cost[-1, argmin]                                   */
    int32_t start;       /* path[1][0]   (index into the FILTERED signal)      */
    int32_t end;         /* path[1][-1]  (argmin of the last row)              */
    int32_t n;           /* samples that survived scale_outliers               */
    int32_t flags;       /* SK_FLAG_*                                          */
} sk_hit;

/* ---- runtime --------------------------------------------------------- */
const char *sk_version(void);
const char *sk_last_error(void);
int  sk_device_count(void);                 /* >= 0, or negative sk_status                */
int  sk_init(int device);                   /* bind the calling thread to `device`        */
/* The same with an explicit context slot (0..15; sk_init uses slot == device).  Every slot has its own stream and
 * scratch, so several host threads (or ranks) can share one GPU: how the sharded multi-GPU paths are exercised on
 * a one-GPU box.  The RCCL entry points still need one device per rank. */
int  sk_init_slot(int slot, int device);
int  sk_shutdown(void);                     /* free every per-device context              */
int  sk_sync(void);                         /* wait for the bound device's stream         */
int  sk_device_name(char *buf, int cap);    /* marketing/gcn name of the bound device     */
/* "0000:c1:00.0" of the bound device (cap >= 16): the key of /sys/bus/pci/devices/<id>/local_cpulist -- a multi-GPU
 * host binds each rank's feeder thread to the CPUs next to its GPU (squigglekit_amd/multigpu.py) */
int  sk_device_pci_bus_id(char *buf, int cap);

/* ---- device memory (for *_dev entry points) --------------------------- */
void *sk_dev_alloc(size_t bytes);           /* NULL on failure                            */
int   sk_dev_free(void *dptr);
int   sk_dev_upload(void *dst_dev, const void *src_host, size_t bytes);
int   sk_dev_download(void *dst_host, const void *src_dev, size_t bytes);

/* ---- pinned host memory (optional, for the host entry points) ---------- */
/* The host entry points (sk_*_batch_*) move a large batch in sub-batches, the H2D copy of one under the kernels
 * of the previous one.  With ordinary (pageable) caller memory the copies are staged by the HIP runtime; buffers
 * from sk_host_alloc() are page-locked and go by DMA at PCIe speed.  Either kind may be passed anywhere a host
 * pointer is expected. */
void *sk_host_alloc(size_t bytes);          /* NULL on failure                            */
int   sk_host_free(void *hptr);

/* ---- segmenter path --------------------------------------------------- */
/* Replaces, per read r: scale_outliers(sig) (segmenter.py:209,311-318) then
 * get_segs(sig, args) (segmenter.py:211,399-470).
 *   sig[r*stride .. r*stride+len[r])  raw samples of read r (caller applies
 *                                      the [:Num] cut of segmenter.py:207)
 *   segs[r][k][0..1]                   k-th [start,end] in FILTERED coordinates
 *   nsegs[r]                           number found; 0 == the reference's False
 * Returns SK_ERR_OVERFLOW if some nsegs[r] > max_segs (nsegs is still exact,
 * segs truncated). */
int sk_segment_batch_i16(const int16_t *sig, int64_t stride, const int32_t *len, int32_t nreads,
                         const sk_seg_params *p, int32_t *segs, int32_t *nsegs, int32_t max_segs);
/* float64 samples (pA TSVs; segmenter.py:198-199), ragged: read r is
 * sig[off[r] .. off[r+1]). */
int sk_segment_batch_f64(const double *sig, const int64_t *off, int32_t nreads,
                         const sk_seg_params *p, int32_t *segs, int32_t *nsegs, int32_t max_segs);
/* the same with the caller's sig[:Num] cut (segmenter.py:207) applied per read: read r is the first len[r] samples of
 * sig[off[r] .. off[r+1]) (len may be NULL: whole reads) -- a parsed TSV chunk goes in as it is, no repacking. */
int sk_segment_batch_f64_len(const double *sig, const int64_t *off, const int32_t *len, int32_t nreads,
                             const sk_seg_params *p, int32_t *segs, int32_t *nsegs, int32_t max_segs);
/* the same for a ragged batch of int32 CENTI-UNITS (sk_tsv_parse_centi: decimal tokens with at most two decimals --
 * SquigglePull's default pA output): sample = centi / 100.0 = float("ddd.dd") bit for bit (segmenter.py:198-199), made
 * on the device; half the bytes over PCIe. */
int sk_segment_batch_centi_len(const int32_t *centi, const int64_t *off, const int32_t *len, int32_t nreads,
                               const sk_seg_params *p, int32_t *segs, int32_t *nsegs, int32_t max_segs);
/* raw reads through the pA route -- what segmenter.py does with fast5 / slow5 input unless --raw_signal is given
 * (segmenter.py:345-349, 366-370, 385): np.round((raw + offset) * (float("%.2f" % range) / digitisation), 2), made on
 * the device from the int16 rows, then scale_outliers + get_segs on the float64 values.  calib[3 r ..] = digitisation,
 * offset, range of read r (what a fast5 / BLOW5 record carries).  The caller applies [:Num] via len[]. */
int sk_segment_batch_i16_pa(const int16_t *sig, int64_t stride, const int32_t *len, int32_t nreads, const double *calib,
                            const sk_seg_params *p, int32_t *segs, int32_t *nsegs, int32_t max_segs);
/* Since round 6 the samples of this route stay int16 on the device: v(x) = rint((x + offset) * unit * 100) / 100 is a
 * non-decreasing function of the sample, so scale_outliers' limits, np.median, np.std (from exact integer centi-pA sums)
 * and the two thresholds of get_segs (segmenter.py:311-318, 410-414, 431) are found in the raw domain -- 2 bytes a sample
 * instead of 8 -- and certified against numpy's rounding; reads that cannot be certified are redone from their float64
 * values in numpy's order.  sk_pa_calib turns the records' {digitisation, offset, range} into the {offset, raw_unit}
 * pairs of the device-resident form (range cut to two decimals first, segmenter.py:385); sk_last_pa_retries: reads of
 * the last such call that took the redo (-1: the call expanded every row to float64 -- rows whose stride is not a
 * multiple of 8). */
int sk_pa_calib(const double *calib, int32_t nreads, double *cal2);
int sk_segment_dev_i16_pa(const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads, const double *d_cal2,
                          const sk_seg_params *p, int32_t *d_segs, int32_t *d_nsegs, int32_t max_segs);
int sk_last_pa_retries(void);
/* device-resident form of sk_segment_batch_i16 (all pointers device). */
int sk_segment_dev_i16(const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads,
                       const sk_seg_params *p, int32_t *d_segs, int32_t *d_nsegs, int32_t max_segs);
/* device-resident form of sk_segment_batch_f64: d_off holds nreads + 1 ZERO-BASED offsets, total = d_off[nreads]
 * (the library cannot look), max_len >= the longest read. */
int sk_segment_dev_f64(const double *d_sig, const int64_t *d_off, int32_t nreads, int64_t total, int64_t max_len,
                       const sk_seg_params *p, int32_t *d_segs, int32_t *d_nsegs, int32_t max_segs);

/* ---- dRNA adapter segmenter (dRNA_segmenter.py, slow5 branch :85-176) ---- */
/* The script hard-codes these (dRNA_segmenter.py:80-104); they are parameters here. */
typedef struct sk_drna_params {
    int32_t error;          /* 5     tolerated out-of-band samples                         */
    int32_t no_err_thresh;  /* 2500  errors only count from this sample index on           */
    int32_t w;              /* 1200  constant corrector period                             */
    int32_t window;         /* 100   shortest segment kept                                 */
    int32_t seg_dist;       /* 1200  merge distance, and the "adapter found" stop distance */
    int32_t t_start;        /* 1000  statistics come from filtered samples [t_start, t_end) */
    int32_t t_end;          /* 5000                                                        */
    double  std_scale;      /* 0.8   top = median + std * std_scale (one sided: a < top)   */
    int32_t lim_low;        /* 0     scale_outliers limits (dRNA_segmenter.py:329-332)     */
    int32_t lim_hi;         /* 1200                                                        */
} sk_drna_params;
/* Per read: scale_outliers, window statistics, the scan.  segs holds every segment collected
 * before the scan stopped (the script prints the first one only). */
int sk_drna_segment_batch_i16(const int16_t *sig, int64_t stride, const int32_t *len, int32_t nreads,
                              const sk_drna_params *p, int32_t *segs, int32_t *nsegs, int32_t max_segs);

/* ---- dRNA adapter segmenter, --signal branch (dRNA_segmenter.py:272-326): rolling mean -------- */
/* t = pandas.Series(filtered).rolling(window=w).mean(); mn = t.mean(); std = t.std();
 * bot = mn - std * std_scale; runs of t < bot, closed by t > bot, merged when closer than seg_dist;
 * the first segment with lo_thresh <= length <= hi_thresh is reported as (start - shift, end - shift).
 * The script reads `w` before assigning it (:282) -- its commented-out default (:81) is 2000. */
typedef struct sk_roll_params {
    int32_t w;              /* 2000    rolling window (min_periods = w)            */
    int32_t seg_dist;       /* 1500                                                */
    int32_t hi_thresh;      /* 200000                                              */
    int32_t lo_thresh;      /* 2000                                                */
    int32_t shift;          /* 1000    subtracted from both ends when reporting    */
    double  std_scale;      /* 0.5                                                 */
    int32_t lim_low;        /* 0       scale_outliers limits (:329-332)            */
    int32_t lim_hi;         /* 1200                                                */
} sk_roll_params;
/* xy[2r], xy[2r+1] = the reported pair of read r when found[r] != 0. */
int sk_drna_roll_batch_i16(const int16_t *sig, int64_t stride, const int32_t *len, int32_t nreads,
                           const sk_roll_params *p, int32_t *xy, int32_t *found);
/* Device-resident forms of the two dRNA_segmenter.py branches (same kernels; d_sig, d_len and the outputs are device
 * pointers, every len[r] is clamped into [0, stride] by the kernels; nothing is synchronised -- sk_sync()).  What
 * bench.py times at 250 000 reads per call.  Reference: dRNA_segmenter.py:85-176 (slow5 branch), :272-326 (--signal). */
int sk_drna_segment_dev_i16(const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads,
                            const sk_drna_params *p, int32_t *d_segs, int32_t *d_nsegs, int32_t max_segs);
int sk_drna_roll_dev_i16(const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads,
                         const sk_roll_params *p, int32_t *d_xy, int32_t *d_found);

/* ---- MotifSeq path ---------------------------------------------------- */
/* Replaces, per read r: scale_outliers (MotifSeq.py:274,317-324), medmad
 * (:192-200) or zscale (:186-191), then mlpy.dtw_subsequence(motif, sig)
 * (:437) reduced to what the caller uses (:438-439): dist, start, end.
 * motif: nmotif float64 points (model[name], MotifSeq.py:354-428). */
int sk_motifseq_batch_i16(const int16_t *sig, int64_t stride, const int32_t *len, int32_t nreads,
                          const double *motif, int32_t nmotif, int32_t scale_mode,
                          int32_t scale_low, int32_t scale_hi, sk_hit *out);
/* Several motifs against the same reads -- the `for name in m_order` loop of MotifSeq.py:436:
 * motif k is motifs[motif_off[k] .. motif_off[k+1]); out is [nmotifs][nreads].  Filter and
 * statistics run once. */
int sk_motifseq_multi_batch_i16(const int16_t *sig, int64_t stride, const int32_t *len, int32_t nreads,
                                const double *motifs, const int32_t *motif_off, int32_t nmotifs,
                                int32_t scale_mode, int32_t scale_low, int32_t scale_hi, sk_hit *out);
int sk_motifseq_batch_f64(const double *sig, const int64_t *off, int32_t nreads,
                          const double *motif, int32_t nmotif, int32_t scale_mode,
                          int32_t scale_low, int32_t scale_hi, sk_hit *out);
/* Several motifs against the same ragged float64 batch -- the `for name in m_order` loop of MotifSeq.py:436 on pA
 * input: the batch is staged and filtered once, one DTW launch set per motif.  motif k = motifs[motif_off[k] ..
 * motif_off[k+1]); out is [nmotifs][nreads]. */
int sk_motifseq_multi_batch_f64(const double *sig, const int64_t *off, int32_t nreads,
                                const double *motifs, const int32_t *motif_off, int32_t nmotifs,
                                int32_t scale_mode, int32_t scale_low, int32_t scale_hi, sk_hit *out);
/* sk_motifseq_multi_batch_f64 for int32 centi-units (see sk_segment_batch_centi_len; MotifSeq.py:270) */
int sk_motifseq_multi_batch_centi(const int32_t *centi, const int64_t *off, int32_t nreads,
                                  const double *motifs, const int32_t *motif_off, int32_t nmotifs,
                                  int32_t scale_mode, int32_t scale_low, int32_t scale_hi, sk_hit *out);
/* device-resident form (d_sig, d_len, d_out device; motif host). */
int sk_motifseq_dev_i16(const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads,
                        const double *motif, int32_t nmotif, int32_t scale_mode,
                        int32_t scale_low, int32_t scale_hi, sk_hit *d_out);
/* device-resident forms of the multi-motif and float64 entry points (motifs / motif_off host; d_out is
 * [nmotifs][nreads]; d_off zero based, total = d_off[nreads], max_len >= the longest read). */
int sk_motifseq_multi_dev_i16(const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads,
                              const double *motifs, const int32_t *motif_off, int32_t nmotifs,
                              int32_t scale_mode, int32_t scale_low, int32_t scale_hi, sk_hit *d_out);
int sk_motifseq_dev_f64(const double *d_sig, const int64_t *d_off, int32_t nreads, int64_t total, int64_t max_len,
                        const double *motif, int32_t nmotif, int32_t scale_mode,
                        int32_t scale_low, int32_t scale_hi, sk_hit *d_out);

/* The mlpy boundary itself: dtw_subsequence(x, y) on already-normalised
 * float64 signals (MotifSeq.py:437).  Batch form: read r is y[off[r]..off[r+1]). */
int sk_dtw_subsequence_batch(const double *x, int32_t nx, const double *y, const int64_t *off,
                             int32_t nreads, sk_hit *out);
/* Single pair, the reference's call shape.  cost_last_row (may be NULL, else
 * ny doubles) receives cost[-1, :] (what view_region plots, MotifSeq.py:507). */
int sk_dtw_subsequence(const double *x, int32_t nx, const double *y, int32_t ny,
                       double *dist, int32_t *start, int32_t *end, double *cost_last_row);

/* The same call in mlpy's own C arithmetic for inputs that hold inf / nan -- `min3` as "a; if (b < m) m = b; if (c < m)
 * m = c", fabs, np.argmin's first-NaN rule, the back-trace of subsequence_path, all evaluated literally by one GPU lane
 * over the full cost matrix.  medmad of a read whose MAD is 0 divides by zero (MotifSeq.py:196-199) and the reference
 * prints whatever mlpy makes of the result; `MotifSeq.py --strict-compat` prints the same row through this entry. */
int sk_dtw_subsequence_cref(const double *x, int32_t nx, const double *y, int32_t ny,
                            double *dist, int32_t *start, int32_t *end);

/* Normalised signal of one read exactly as the reference hands it to
 * dtw_subsequence (used by MotifSeq -x/--sig_extract, MotifSeq.py:446-447).
 * out must hold len doubles; *n_out receives the filtered length. */
int sk_normalise_i16(const int16_t *sig, int32_t len, int32_t scale_mode,
                     int32_t scale_low, int32_t scale_hi, double *out, int32_t *n_out);
int sk_normalise_f64(const double *sig, int32_t len, int32_t scale_mode,
                     int32_t scale_low, int32_t scale_hi, double *out, int32_t *n_out);

/* ---- TSV ingest (host code; no GPU needed) ------------------------------ */
/* Native replacement of the per-line split + int()/float() loops that feed the hot path
 * (segmenter.py:192-201 reads columns 4.., MotifSeq.py:265-270 columns 8..; layout written by
 * SquigglePull.py:243-253).  Three calls: count lines, count data tokens per line (caller turns
 * them into offsets), parse into one flat float64 array.  Conversion is exactly float()'s for
 * plain decimal tokens; lines with any other token get SK_TSV_SLOW and should be parsed by the
 * caller the slow way.  Call sk_tsv_count_lines first on every new content of a buffer: it builds the
 * per-thread line index the other calls reuse (they re-check a cached index against the bytes -- every line
 * start must follow a newline, the line count must match -- and rebuild it otherwise). */
enum {
    SK_TSV_ALLINT   = 1,   /* every data token is [+-]digits                                  */
    SK_TSV_ANY      = 2,   /* some value is non-zero (the reference skips reads where none is) */
    SK_TSV_FIRSTDOT = 4,   /* the first data token contains '.' (segmenter.py:198 picks float) */
    SK_TSV_SLOW     = 8,   /* a token is outside the plain grammar: use the fallback parser    */
    SK_TSV_SHORT    = 16,  /* the line has no data column at all                               */
    SK_TSV_CENTI    = 32   /* sk_tsv_parse_centi: every data token has at most two decimals     */
};
int64_t sk_tsv_count_lines(const char *buf, size_t len);
int sk_tsv_count_tokens(const char *buf, size_t len, int32_t start_col, int64_t nlines, int64_t *ntok,
                        int32_t nthreads);
int sk_tsv_parse(const char *buf, size_t len, int32_t start_col, int64_t nlines, const int64_t *off,
                 double *values, int64_t *name_off, int32_t *name_len, int64_t *id_off, int32_t *id_len,
                 int32_t *flags, int32_t nthreads);

/* Decimal lines as int32 centi-units (round 6): SquigglePull's default output is np.round(pA, 2)
 * (SquigglePull.py:183-189,222), and float("ddd.dd") == (double)ddddd / 100.0 bit for bit, so a line whose data tokens
 * all have at most two decimals travels as int32 (half the bytes, one pass) and becomes float64 on the device.  Same
 * layout as sk_tsv_parse (values[off[i] .. off[i+1]) = line i); flags[i] & SK_TSV_CENTI says line i's values are valid --
 * a chunk with any line without it goes through sk_tsv_parse instead.  Replaces the float() loops of
 * segmenter.py:198-199 / MotifSeq.py:270 for such lines. */
int sk_tsv_parse_centi(const char *buf, size_t len, int32_t start_col, int64_t nlines, const int64_t *off,
                       int32_t *values, int64_t *name_off, int32_t *name_len, int64_t *id_off, int32_t *id_len,
                       int32_t *flags, int32_t nthreads);

/* Integer lines straight into int16 rows (the raw-signal TSVs SquigglePull writes): rows[i * stride ..] = line i's
 * data tokens, nsamp[i] their number.  flags[i] has SK_TSV_ALLINT only if every data token is [+-]digits, fits
 * int16 and the line has at most `stride` of them (the row is valid only then); other lines get SK_TSV_SLOW /
 * SK_TSV_SHORT and go the general way.  line_off[i] (nlines + 1 entries) = byte offset of line i. */
int sk_tsv_parse_i16(const char *buf, size_t len, int32_t start_col, int64_t nlines, int64_t stride, int16_t *rows,
                     int32_t *nsamp, int64_t *name_off, int32_t *name_len, int64_t *id_off, int32_t *id_len,
                     int32_t *flags, int64_t *line_off, int32_t nthreads);

/* ---- result tables as text / BLOW5 ingest (host code; no GPU needed) ------ */
/* The rows the command-line tools print (MotifSeq.py:446-449: 12 tab-separated columns per hit; segmenter.py:222-227:
 * name <TAB> s0,e0,s1,e1...) formatted on all cores: columns of strings, int32, float64 -- floats exactly as Python's
 * "{}".format(float) / repr() writes them -- or comma-joined int32 lists.  Rows with skip[i] != 0 are left out.
 * Returns a malloc'ed buffer of *out_len bytes (free with sk_fmt_free), NULL on failure. */
enum { SK_FMT_STR = 0, SK_FMT_I32 = 1, SK_FMT_F64 = 2, SK_FMT_CONST = 3, SK_FMT_I32LIST = 4, SK_FMT_STRSPAN = 5 };
typedef struct sk_fmt_col {
    int32_t        kind;     /* SK_FMT_*                                                                    */
    const void    *data;     /* STR / CONST: bytes; I32 / I32LIST: int32[]; F64: double[]                   */
    const int64_t *off;      /* STR / I32LIST: nrows + 1 offsets into data; CONST: off[0..1]; STRSPAN: [nrows][2]
                                = first / one-past-last byte of row i's string in data; else unused         */
} sk_fmt_col;
void *sk_fmt_rows(int64_t nrows, int32_t ncols, const sk_fmt_col *cols, const uint8_t *skip, int32_t nthreads,
                  int64_t *out_len);
void  sk_fmt_free(void *p);
/* out[i] = the standard normal CDF of z[i], the very double scipy.special.ndtr returns (MotifSeq.py:444 prints
 * scipy.stats.norm.cdf(z) digit for digit): the Cephes ndtr, restated in csrc/sk_io.cpp. */
void  sk_ndtr(const double *z, double *out, int64_t n);

/* BLOW5 (binary SLOW5: what the reference reads through pyslow5, segmenter.py:321-396, dRNA_segmenter.py:85-100).
 * sk_blow5_index walks the records of a file image from byte `first` (just behind the ASCII header): payload offset
 * and size of up to `cap` records; returns the number of records in the file (call with cap = 0 to count).
 * sk_blow5_index_some is the same walk for at most max_rec records from byte `pos`; *next_pos = where the next call
 * continues, fewer than max_rec records returned = end of the file (a reader indexes chunk by chunk).
 * sk_blow5_rows_i16 decodes records (comp: 0 = stored, 1 = zlib) into int16 rows of `stride` samples on all cores:
 * nsamp[i] samples, ids[i * id_width ..] the read id (NUL padded; NULL to skip), calib[3 i ..] = digitisation, offset,
 * range (NULL to skip); flags[i]: 1 = longer than a row (truncated to stride), 2 = unreadable record (also: offset /
 * size outside the `len` bytes of buf, a zlib record that inflates past 256x its size), 4 = id cut to id_width.
 * Both index calls return SK_ERR_INVALID for a file cut inside a record or a size field, or with anything but the
 * end marker behind the last record. */
int64_t sk_blow5_index(const void *buf, int64_t len, int64_t first, int64_t *rec_off, int64_t *rec_size, int64_t cap);
int64_t sk_blow5_index_some(const void *buf, int64_t len, int64_t pos, int64_t max_rec, int64_t *rec_off,
                            int64_t *rec_size, int64_t *next_pos);
int sk_blow5_rows_i16(const void *buf, int64_t len, const int64_t *rec_off, const int64_t *rec_size, int64_t nrec,
                      int32_t comp, int64_t stride, int16_t *rows, int32_t *nsamp, char *ids, int32_t id_width, double *calib,
                      int32_t *flags, int32_t nthreads);

/* ---- multi-GPU: the final gather of result records (RCCL over xGMI) ------ */
/* The reference's per-read loops (segmenter.py:189-230, MotifSeq.py:261-298) carry no state from one read
 * to the next, so N GPUs take contiguous blocks of the reads with no data-path collective; the one exchange
 * is an all-gather of the fixed-size records at the end.  librccl.so is loaded on first use; when it (or a
 * communicator) is not available these return SK_ERR_UNSUPPORTED and the caller concatenates the shards on
 * the host.  One communicator per device; the calls act on the calling thread's bound device.
 *   one process, one thread per GPU:   sk_comm_init_all(devices, n)  once, from any thread
 *   one process per GPU:               rank 0: sk_comm_unique_id(id) -> hand the 128 bytes to every rank;
 *                                      every rank: sk_init(dev); sk_comm_init_rank(id, nranks, rank)        */
int sk_comm_unique_id(void *id128);
int sk_comm_init_rank(const void *id128, int nranks, int rank);
int sk_comm_init_all(const int *devices, int ndev);
int sk_comm_info(int *nranks, int *rank);      /* what RCCL reports for this device's communicator */
/* d_recv[nranks * bytes] <- every rank's d_send[bytes], rank order; enqueued on the device's stream */
int sk_comm_allgather_dev(const void *d_send, void *d_recv, size_t bytes);
/* the same for small host buffers (timings, a barrier); synchronous */
int sk_comm_allgather_host(const void *send, void *recv, size_t bytes);
int sk_comm_destroy(void);

/* ---- tuning switches --------------------------------------------------- */
/* Every environment switch the library reads, as text: one line per switch, "name<TAB>values the parity test flips it
 * to<TAB>description".  None changes results; all are ignored unless SK_TUNING=1 is set too.  Returns the bytes needed
 * (including the terminating NUL); buf may be NULL. */
int sk_tunables(char *buf, int cap);

/* ---- instrumentation -------------------------------------------------- */
/* HIP-event durations (ms) of the kernels of the most recent *_dev / batch
 * call on this thread's device: prep (filter+stats), main (DTW or segment walk). */
int sk_last_kernel_ms(float *prep_ms, float *main_ms);
/* Reads of the most recent DTW call whose optimal path was longer than the two-pass
 * look-back window and were therefore recomputed by the exact single pass (diagnostic). */
int sk_last_dtw_retries(void);
/* Reads of the most recent DTW call whose path crossed the window pass's first (short) look-back and were redone
 * by its second tier (diagnostic; 0 when the call did not use the screening scheme). */
int sk_last_dtw_tier2(void);
/* Run-time guard of the screening scheme (the default DTW path: fixed-point screening + certified exact window).
 * Its exactness rests on a premise -- every screening cost lies within E = N + n + 2 units of the exact one -- that
 * is derived, not observed; the guard observes it.  out[0] results the window pass refused because the exact distance
 * contradicted the screening values (premise violations), out[1] reads the audit re-ran with the exact single pass
 * (one in 4 096, hashed; a call of few, long reads -- whose window passes are shorter than that one sweep -- is
 * audited one call in K <= 64, so that waiting for the sweep costs about 2 % on average: out[1] = 0 on the others;
 * SK_TUNING=1 SK_DTW_AUDIT_PERIOD=n audits every call), out[2] audited reads whose record differed (the exact record wins), out[3] reads kept
 * away from the screening because their sample image cannot be bounded tightly enough (exact pass, by design),
 * out[4] = out[0] + out[2] (the alarm), out[5] = 1 when the alarm made the library redo reads with the exact single
 * pass -- every read of the launch set that raised it (one motif over one ingest sub-batch) and of every later launch
 * set of the same call; launch sets of a multi-motif / sub-batched call that finished earlier passed their own premise
 * test and audit and are kept, out[6] reads whose candidate columns fell into two clusters and took a second window instead of the
 * exact pass (diagnostic), out[7] reserved.  In a healthy build out[0] = out[2] = out[4] = out[5] = 0, always.  The
 * two short forms return out[0] / out[2] (or a negative status).  The reference has no counterpart: mlpy's one
 * exact pass (/root/reference/MotifSeq.py:437-439) is what every record must equal. */
int sk_last_dtw_guard(int32_t *out /* [8] */);
int sk_last_dtw_premise_violations(void);
int sk_last_dtw_audit_mismatches(void);
/* Steps of the exact window pass in the most recent DTW call: out[0] = steps its wavefronts ran (the read groups of a
 * wavefront step together), out[1] = steps the reads asked for, summed over the groups.  bench.py prices the pass's own
 * issue roof with these (8 instructions per cell and step; mlpy's exact recurrence with start tracking,
 * /root/reference/MotifSeq.py:437-439).  Zeros when the call did not take the screening scheme. */
int sk_last_dtw_window_steps(uint64_t *out /* [2] */);
/* Reads of the most recent float64 call (sk_segment_*_f64, sk_motifseq_*_f64 with medmad) whose comparisons /
 * selection the streaming statistics kernel could not certify and that were redone in numpy's order (diagnostic);
 * -1 when the call did not use the streaming kernel (reads longer than 4 096 samples, zscale). */
int sk_last_f64_retries(void);
/* Shader clock (GHz) the screening pass of the most recent DTW call ran at: its first wavefront counts shader cycles
 * (s_memtime) against the constant 100 MHz reference (s_memrealtime) over its whole sweep.  0 when the call did not
 * use the screening scheme.  bench.py prices the VALU-issue roofline at this clock instead of a nominal one. */
int sk_last_dtw_clock(double *ghz);
/* Per-launch view of the most recent two-pass DTW call: summed HIP-event time and launch count
 * of the distance pass (k_sdtw<..,DIST>) and of the start pass (k_sdtw<..,START>), and the reads
 * covered by the largest launch.  *dist_launches == 0 means the call used the single pass. */
int sk_last_dtw_profile(float *dist_ms, int *dist_launches, float *start_ms, int *start_launches,
                        int *reads_per_launch);
/* Synthetic squiggle generator on the device (bench input; not a reference
 * function): fills d_sig[nreads][nsamples] int16 deterministically from seed. */
int sk_synth_squiggles_dev(int16_t *d_sig, int64_t stride, int32_t nreads, int32_t nsamples,
                           uint64_t seed, const double *motif, int32_t nmotif);
/* Variants of the generator for bench.py's sensitivity runs and its multi-rank parity check (not reference
 * functions either).  Row r of a call is row row0 + r of the seed's batch, so any slice of any rank's batch can be
 * regenerated anywhere.  Defaults {0, 500, 0, 1, NULL, 0, 0} reproduce sk_synth_squiggles_dev. */
typedef struct sk_synth_opts {
    int64_t        row0;             /* first row of the seed's batch this call generates                      */
    int32_t        hit_permille;     /* reads carrying an exact copy of the motif (default 500)                */
    int32_t        stretch_permille; /* reads carrying the motif with every point repeated `stretch` times     */
    int32_t        stretch;          /* (their optimal path is that much wider: they take the exact retry)     */
    const int16_t *tmpl;             /* HOST pointer or NULL.  Non-NULL: every read is a window of this        */
    int32_t        ntmpl;            /* measured squiggle (ntmpl samples) at a random offset, plus rounded     */
    double         tmpl_noise;       /* N(0, tmpl_noise) noise; no plateaus / implants / spikes               */
} sk_synth_opts;
int sk_synth_variant_dev(int16_t *d_sig, int64_t stride, int32_t nreads, int32_t nsamples, uint64_t seed,
                         const double *motif, int32_t nmotif, const sk_synth_opts *o);
/* The float64 pA image of a device-resident int16 batch, value for value what SquigglePull writes for it
 * (SquigglePull.py:183-189,238-240: np.round((raw + offset) * range / digitisation, 2)) -- input for the float64
 * entry points (bench / tests; not a reference function of the hot path).  d_out: nreads * nsamples doubles,
 * d_off: nreads + 1 zero-based offsets (read r at r * nsamples). */
int sk_synth_pa_dev(const int16_t *d_raw, int64_t stride, int32_t nreads, int32_t nsamples,
                    double offset, double range, double digitisation, double *d_out, int64_t *d_off);

#ifdef __cplusplus
}
#endif
#endif /* SQUIGGLEKIT_HIP_H */
