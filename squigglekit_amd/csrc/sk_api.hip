// sk_api.hip -- the C-ABI entry points of include/squigglekit_hip.h.
// Host logic only: argument checks, scratch sizing, H2D / launches / D2H on the bound
// device's stream.  No arithmetic on sample data happens on the host.
#include "sk_common.h"
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace {

int check_i16(const void *sig, int64_t stride, const int32_t *len, int32_t nreads)
{
    if (nreads < 0) return sk_fail(SK_ERR_INVALID, "nreads < 0");
    if (nreads && (!sig || !len)) return sk_fail(SK_ERR_INVALID, "NULL sig/len");
    if (stride <= 0) return sk_fail(SK_ERR_INVALID, "stride must be positive");
    return SK_OK;
}

// host-buffer entry points: every len[r] must lie in [0, stride] -- the kernels use it as a trip count
// over the read's row (the *_dev_* entry points cannot look at device memory; the kernels clamp there)
int check_len_host(const int32_t *len, int32_t nreads, int64_t stride)
{
    for (int32_t r = 0; r < nreads; r++)
        if (len[r] < 0 || (int64_t)len[r] > stride)
            return sk_fail(SK_ERR_INVALID, "len[%d] = %d is outside [0, stride = %lld]", r, len[r], (long long)stride);
    return SK_OK;
}

int check_seg_params(const sk_seg_params *p)
{
    if (!p) return sk_fail(SK_ERR_INVALID, "NULL sk_seg_params");
    if (p->corrector < 0)
        return sk_fail(SK_ERR_INVALID, "corrector must be >= 0 (the reference divides by zero otherwise)");
    return SK_OK;
}

// clamp the outlier limits to what an int16 can hold (no sample can lie outside)
void clamp_limits(int32_t *lo, int32_t *hi)
{
    if (*lo < -32769) *lo = -32769;
    if (*hi > 32768) *hi = 32768;
}

__global__ void k_normalise_i16(const int16_t *__restrict__ comp, const sk_prep *__restrict__ prep,
                                double *__restrict__ out)
{
    const sk_prep pr = prep[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < pr.n; i += gridDim.x * blockDim.x)
        out[i] = ((double)comp[i] - pr.center) / pr.scale;     // MotifSeq.py:199 / sklearn.scale
}

__global__ void k_normalise_f64(const double *__restrict__ comp, const sk_prep *__restrict__ prep,
                                double *__restrict__ out)
{
    const sk_prep pr = prep[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < pr.n; i += gridDim.x * blockDim.x)
        out[i] = ((comp[i] - pr.center) - pr.top) / pr.scale - pr.bot;   // top / bot: sklearn's re-centring, 0 unless applied
}

// mlpy 3.5.0's subsequence() / subsequence_path() arithmetic restated LITERALLY, for inputs that hold inf / nan (medmad
// of a read whose MAD is 0, MotifSeq.py:196-199): `min3` is "m = a; if (b < m) m = b; if (c < m) m = c" and every
// comparison with a NaN is false, exactly as the C code behaves; np.argmin returns the first NaN.  The systolic kernels
// use v_min_f64, which drops NaNs -- they are never given such input (SK_FLAG_DEGENERATE).  One lane walks the whole
// matrix: this runs for the handful of degenerate reads `MotifSeq.py --strict-compat` prints.
__device__ __forceinline__ double cref_min3(double a, double b, double c)
{
    double m = a;
    if (b < m) m = b;
    if (c < m) m = c;
    return m;
}

__global__ void k_dtw_cref(const double *__restrict__ x, int n, const double *__restrict__ y, int m,
                           double *__restrict__ cost, sk_hit *__restrict__ out)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    cost[0] = fabs(x[0] - y[0]);
    for (int i = 1; i < n; i++) cost[(size_t)i * m] = fabs(x[i] - y[0]) + cost[(size_t)(i - 1) * m];
    for (int j = 1; j < m; j++) cost[j] = fabs(x[0] - y[j]);                       // free start
    for (int i = 1; i < n; i++) {
        const double *up = cost + (size_t)(i - 1) * m;
        double *row = cost + (size_t)i * m;
        for (int j = 1; j < m; j++) row[j] = fabs(x[i] - y[j]) + cref_min3(up[j], up[j - 1], row[j - 1]);
    }
    const double *last = cost + (size_t)(n - 1) * m;
    int idx = 0;                                                                    // np.argmin: first NaN, else first minimum
    double best = last[0];
    if (!(best != best))
        for (int j = 1; j < m; j++) {
            if (last[j] != last[j]) { idx = j; break; }
            if (last[j] < best) { best = last[j]; idx = j; }
        }
    int i = n - 1, j = idx;                                                         // subsequence_path: while i > 0
    while (i > 0) {
        if (j == 0) { i--; continue; }
        const double up = cost[(size_t)(i - 1) * m + j], dg = cost[(size_t)(i - 1) * m + j - 1], lf = cost[(size_t)i * m + j - 1];
        const double mc = cref_min3(up, dg, lf);
        if (dg == mc)      { i--; j--; }
        else if (lf == mc) { j--; }
        else               { i--; }
    }
    sk_hit h;
    h.dist = last[idx]; h.start = j; h.end = idx; h.n = m; h.flags = 0;
    out[0] = h;
}

// The host entry points move a large batch in sub-batches: the H2D copy of sub-batch k + 1 (second stream) runs
// under the kernels of sub-batch k.  Pageable caller memory: hipMemcpyAsync stages it and returns, the kernels
// launched before keep running meanwhile.  Memory from sk_host_alloc() (pinned): plain DMA at PCIe speed.
struct SubBatches {
    int32_t per = 0, n = 1;
};
SubBatches sub_batches(int32_t nreads, int64_t stride)
{
    SubBatches sb;
    int64_t target = (int64_t)256 << 20;                    // bytes of samples per sub-batch
    if (const char *e = sk_tune("SK_INGEST_MB")) { const long v = atol(e); if (v > 0) target = (int64_t)v << 20; }
    int64_t per = target / (stride * (int64_t)sizeof(int16_t));
    if (per < 4096) per = 4096;
    if (per * 2 > nreads) { sb.per = nreads; sb.n = 1; return sb; }
    sb.n = (int32_t)((nreads + per - 1) / per);
    sb.per = (int32_t)(((int64_t)nreads + sb.n - 1) / sb.n);
    return sb;
}

int second_stream(sk_ctx *c)
{
    if (!c->stream2) {
        SK_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        for (int i = 0; i < 9; i++) SK_HIP(hipEventCreateWithFlags(&c->ev_chunk[i], hipEventDisableTiming));
    }
    return SK_OK;
}

int motifseq_dev(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads,
                 const double *motif, int32_t nmotif, int32_t scale_mode, int32_t scale_low, int32_t scale_hi,
                 sk_hit *d_out, int accumulate);

} // namespace

extern "C" {

// ------------------------------------------------------------------ pinned host memory for callers
void *sk_host_alloc(size_t bytes)
{
    sk_ctx *c = sk_cur();
    if (!c) return nullptr;
    sk_ctx_guard c_lock(c);
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        sk_fail(SK_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

int sk_host_free(void *p)
{
    if (p) SK_HIP(hipHostFree(p));
    return SK_OK;
}

// ------------------------------------------------------------------ MotifSeq, device resident
// Last step of the host-facing DTW entry points: the guard counters of the screening scheme (sk_last_dtw_guard) ride
// with the final synchronisation, and an alarm -- something that cannot happen in a healthy build -- is said out loud
// once per call (the records are right either way: the library redid the call with the exact pass).
static int finish_dtw_host(sk_ctx *c)
{
    int32_t g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c->retry_dev && c->dtwcnt.p)
        SK_HIP(hipMemcpyAsync(g, (const int32_t *)c->dtwcnt.p + 8, sizeof g, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));
    if (g[4])
        fprintf(stderr, "squigglekit: DTW screening guard: %d premise violation(s), %d audit mismatch(es) of %d audited reads -- "
                        "the launch set(s) concerned and every later one of this call were %s by the exact pass; please report this\n",
                g[0], g[2], g[1], g[5] ? "redone" : "NOT redone");
    return SK_OK;
}

int sk_motifseq_dev_i16(const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads,
                        const double *motif, int32_t nmotif, int32_t scale_mode,
                        int32_t scale_low, int32_t scale_hi, sk_hit *d_out)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    return motifseq_dev(c, d_sig, stride, d_len, nreads, motif, nmotif, scale_mode, scale_low, scale_hi, d_out, 0);
}

} // extern "C"

namespace {

int motifseq_dev(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads,
                 const double *motif, int32_t nmotif, int32_t scale_mode, int32_t scale_low, int32_t scale_hi,
                 sk_hit *d_out, int accumulate)
{
    int rc = check_i16(d_sig, stride, d_len, nreads);
    if (rc) return rc;
    if (!motif || nmotif <= 0) return sk_fail(SK_ERR_INVALID, "empty motif");
    if (scale_mode != SK_SCALE_MEDMAD && scale_mode != SK_SCALE_ZSCALE)
        return sk_fail(SK_ERR_INVALID, "unknown scale mode %d", scale_mode);
    if (nreads == 0) return SK_OK;
    if (!d_out) return sk_fail(SK_ERR_INVALID, "NULL out");
    clamp_limits(&scale_low, &scale_hi);
    if ((rc = sk_reserve(c, &c->comp, (size_t)nreads * (size_t)stride * sizeof(int16_t)))) return rc;
    if ((rc = sk_reserve(c, &c->prep, (size_t)nreads * sizeof(sk_prep)))) return rc;

    // medmad with the usual limits: filter + statistics ride in the screening pass as its prologue (sk_sdtwq.hip);
    // sk_launch_sdtw runs them as a kernel of their own when it does not take the screening scheme
    sk_prep_fuse fz;
    fz.raw = d_sig; fz.len = d_len; fz.lo = scale_low; fz.hi = scale_hi;
    fz.mode = scale_mode == SK_SCALE_MEDMAD ? SK_PREP_MEDMAD : SK_PREP_ZSCALE;
    const bool fuse = sk_sdtw_fuse_ok(scale_low, scale_hi, fz.mode, stride);
    SK_HIP(hipEventRecord(c->ev[0], c->stream));
    if (!fuse) {
        rc = sk_launch_prep_i16(c, d_sig, stride, d_len, nreads, scale_low, scale_hi,
                                scale_mode == SK_SCALE_MEDMAD ? SK_PREP_MEDMAD : SK_PREP_ZSCALE, 0.0,
                                (int16_t *)c->comp.p, (sk_prep *)c->prep.p, nullptr, 0);
        if (rc) return rc;
    }
    SK_HIP(hipEventRecord(c->ev[1], c->stream));

    sk_sdtw_args a;
    a.feed = SK_FEED_I16; a.samples = c->comp.p; a.stride = stride; a.off = nullptr;
    a.prep = (const sk_prep *)c->prep.p; a.nreads = nreads; a.motif = motif; a.nmotif = nmotif;
    a.out = d_out; a.last_row = nullptr; a.max_len = stride; a.force_single = 0; a.accumulate = accumulate;
    a.fuse = fuse ? &fz : nullptr;
    rc = sk_launch_sdtw(c, &a);
    if (rc) return rc;
    c->ev_valid = true;
    return SK_OK;
}

} // namespace

extern "C" {

// ------------------------------------------------------------------ MotifSeq, host buffers
int sk_motifseq_batch_i16(const int16_t *sig, int64_t stride, const int32_t *len, int32_t nreads,
                          const double *motif, int32_t nmotif, int32_t scale_mode,
                          int32_t scale_low, int32_t scale_hi, sk_hit *out)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    int rc = check_i16(sig, stride, len, nreads);
    if (rc) return rc;
    if ((rc = check_len_host(len, nreads, stride))) return rc;
    if (nreads == 0) return SK_OK;
    if (!out) return sk_fail(SK_ERR_INVALID, "NULL out");
    const size_t sb = (size_t)nreads * (size_t)stride * sizeof(int16_t);
    if ((rc = sk_reserve(c, &c->sig, sb))) return rc;
    if ((rc = sk_reserve(c, &c->len, (size_t)nreads * sizeof(int32_t)))) return rc;
    if ((rc = sk_reserve(c, &c->out, (size_t)nreads * sizeof(sk_hit)))) return rc;
    const SubBatches B = sub_batches(nreads, stride);
    if (B.n > 1 && (rc = second_stream(c))) return rc;
    for (int32_t bi = 0; bi < B.n; bi++) {
        const int32_t r0 = bi * B.per;
        const int32_t nr = (nreads - r0 < B.per) ? nreads - r0 : B.per;
        if (nr <= 0) break;
        int16_t *d_sig = (int16_t *)c->sig.p + (size_t)r0 * (size_t)stride;
        int32_t *d_len = (int32_t *)c->len.p + r0;
        hipStream_t cs = (B.n > 1) ? c->stream2 : c->stream;
        SK_HIP(hipMemcpyAsync(d_sig, sig + (size_t)r0 * (size_t)stride, (size_t)nr * (size_t)stride * sizeof(int16_t),
                              hipMemcpyHostToDevice, cs));
        SK_HIP(hipMemcpyAsync(d_len, len + r0, (size_t)nr * sizeof(int32_t), hipMemcpyHostToDevice, cs));
        if (B.n > 1) {                                      // the kernels of this sub-batch wait for its copy only
            SK_HIP(hipEventRecord(c->ev_chunk[bi & 7], cs));
            SK_HIP(hipStreamWaitEvent(c->stream, c->ev_chunk[bi & 7], 0));
        }
        rc = motifseq_dev(c, d_sig, stride, d_len, nr, motif, nmotif, scale_mode, scale_low, scale_hi,
                          (sk_hit *)c->out.p + r0, bi > 0);
        if (rc) return rc;
    }
    SK_HIP(hipMemcpyAsync(out, c->out.p, (size_t)nreads * sizeof(sk_hit), hipMemcpyDeviceToHost, c->stream));
    if ((rc = finish_dtw_host(c))) return rc;
    return SK_OK;
}

// One (sub-)batch of the multi-motif path, everything device resident: filter + statistics once (as the prologue of
// the first motif's screening pass when that applies), then one DTW launch set per motif.  Motif k's records go to
// d_out + k * out_stride.
static int motifseq_multi_dev(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nr,
                              int16_t *d_comp, sk_prep *d_prep, const double *motifs, const int32_t *motif_off,
                              int32_t nmotifs, int32_t scale_mode, int32_t scale_low, int32_t scale_hi,
                              sk_hit *d_out, int64_t out_stride, int later_batch)
{
    int rc;
    sk_prep_fuse fz;
    fz.raw = d_sig; fz.len = d_len; fz.lo = scale_low; fz.hi = scale_hi;
    fz.mode = scale_mode == SK_SCALE_MEDMAD ? SK_PREP_MEDMAD : SK_PREP_ZSCALE;
    const bool fuse = sk_sdtw_fuse_ok(scale_low, scale_hi, fz.mode, stride);
    SK_HIP(hipEventRecord(c->ev[0], c->stream));
    if (!fuse) {
        rc = sk_launch_prep_i16(c, d_sig, stride, d_len, nr, scale_low, scale_hi,
                                scale_mode == SK_SCALE_MEDMAD ? SK_PREP_MEDMAD : SK_PREP_ZSCALE, 0.0, d_comp, d_prep,
                                nullptr, 0);
        if (rc) return rc;
    }
    SK_HIP(hipEventRecord(c->ev[1], c->stream));
    for (int32_t k = 0; k < nmotifs; k++) {
        sk_sdtw_args a;
        a.feed = SK_FEED_I16; a.samples = d_comp; a.stride = stride; a.off = nullptr;
        a.prep = d_prep; a.nreads = nr; a.motif = motifs + motif_off[k];
        a.nmotif = motif_off[k + 1] - motif_off[k]; a.out = d_out + (size_t)k * (size_t)out_stride;
        a.last_row = nullptr; a.max_len = stride; a.force_single = 0;
        a.accumulate = (later_batch || k > 0) ? 1 : 0;
        a.fuse = (fuse && k == 0) ? &fz : nullptr;      // (the later motifs find the samples / statistics in place)
        if ((rc = sk_launch_sdtw(c, &a))) return rc;
    }
    return SK_OK;
}

static int check_multi(const double *motifs, const int32_t *motif_off, int32_t nmotifs, int32_t scale_mode)
{
    if (!motifs || !motif_off || nmotifs <= 0) return sk_fail(SK_ERR_INVALID, "no motifs");
    for (int32_t k = 0; k < nmotifs; k++)
        if (motif_off[k + 1] <= motif_off[k]) return sk_fail(SK_ERR_INVALID, "motif %d is empty", k);
    if (scale_mode != SK_SCALE_MEDMAD && scale_mode != SK_SCALE_ZSCALE)
        return sk_fail(SK_ERR_INVALID, "unknown scale mode %d", scale_mode);
    return SK_OK;
}

// device-resident form: d_out is [nmotifs][nreads]
int sk_motifseq_multi_dev_i16(const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads,
                              const double *motifs, const int32_t *motif_off, int32_t nmotifs,
                              int32_t scale_mode, int32_t scale_low, int32_t scale_hi, sk_hit *d_out)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    int rc = check_i16(d_sig, stride, d_len, nreads);
    if (rc) return rc;
    if ((rc = check_multi(motifs, motif_off, nmotifs, scale_mode))) return rc;
    if (nreads == 0) return SK_OK;
    if (!d_out) return sk_fail(SK_ERR_INVALID, "NULL out");
    clamp_limits(&scale_low, &scale_hi);
    if ((rc = sk_reserve(c, &c->comp, (size_t)nreads * (size_t)stride * sizeof(int16_t)))) return rc;
    if ((rc = sk_reserve(c, &c->prep, (size_t)nreads * sizeof(sk_prep)))) return rc;
    rc = motifseq_multi_dev(c, d_sig, stride, d_len, nreads, (int16_t *)c->comp.p, (sk_prep *)c->prep.p, motifs,
                            motif_off, nmotifs, scale_mode, scale_low, scale_hi, d_out, nreads, 0);
    if (rc) return rc;
    c->ev_valid = true;
    return SK_OK;
}

// Several motifs against the same reads (the `for name in m_order` loop of MotifSeq.py:436):
// filter + statistics once, one DTW launch set per motif.  out is [nmotifs][nreads].
int sk_motifseq_multi_batch_i16(const int16_t *sig, int64_t stride, const int32_t *len, int32_t nreads,
                                const double *motifs, const int32_t *motif_off, int32_t nmotifs,
                                int32_t scale_mode, int32_t scale_low, int32_t scale_hi, sk_hit *out)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    int rc = check_i16(sig, stride, len, nreads);
    if (rc) return rc;
    if ((rc = check_len_host(len, nreads, stride))) return rc;
    if ((rc = check_multi(motifs, motif_off, nmotifs, scale_mode))) return rc;
    if (nreads == 0) return SK_OK;
    if (!out) return sk_fail(SK_ERR_INVALID, "NULL out");
    clamp_limits(&scale_low, &scale_hi);
    const size_t sb = (size_t)nreads * (size_t)stride * sizeof(int16_t);
    const size_t ob = (size_t)nreads * (size_t)nmotifs * sizeof(sk_hit);
    if ((rc = sk_reserve(c, &c->sig, sb))) return rc;
    if ((rc = sk_reserve(c, &c->len, (size_t)nreads * sizeof(int32_t)))) return rc;
    if ((rc = sk_reserve(c, &c->comp, sb))) return rc;
    if ((rc = sk_reserve(c, &c->prep, (size_t)nreads * sizeof(sk_prep)))) return rc;
    if ((rc = sk_reserve(c, &c->out, ob))) return rc;
    // sub-batches as in sk_motifseq_batch_i16: the H2D copy of one runs on the second stream under the kernels of the
    // previous one; filter + statistics once per sub-batch (as the prologue of the first motif's screening pass when
    // that applies), then one DTW launch set per motif
    const SubBatches B = sub_batches(nreads, stride);
    if (B.n > 1 && (rc = second_stream(c))) return rc;
    for (int32_t bi = 0; bi < B.n; bi++) {
        const int32_t r0 = bi * B.per;
        const int32_t nr = (nreads - r0 < B.per) ? nreads - r0 : B.per;
        if (nr <= 0) break;
        int16_t *d_sig = (int16_t *)c->sig.p + (size_t)r0 * (size_t)stride;
        int16_t *d_comp = (int16_t *)c->comp.p + (size_t)r0 * (size_t)stride;
        int32_t *d_len = (int32_t *)c->len.p + r0;
        sk_prep *d_prep = (sk_prep *)c->prep.p + r0;
        hipStream_t cs = (B.n > 1) ? c->stream2 : c->stream;
        SK_HIP(hipMemcpyAsync(d_sig, sig + (size_t)r0 * (size_t)stride, (size_t)nr * (size_t)stride * sizeof(int16_t),
                              hipMemcpyHostToDevice, cs));
        SK_HIP(hipMemcpyAsync(d_len, len + r0, (size_t)nr * sizeof(int32_t), hipMemcpyHostToDevice, cs));
        if (B.n > 1) {                                      // the kernels of this sub-batch wait for its copy only
            SK_HIP(hipEventRecord(c->ev_chunk[bi & 7], cs));
            SK_HIP(hipStreamWaitEvent(c->stream, c->ev_chunk[bi & 7], 0));
        }
        rc = motifseq_multi_dev(c, d_sig, stride, d_len, nr, d_comp, d_prep, motifs, motif_off, nmotifs, scale_mode,
                                scale_low, scale_hi, (sk_hit *)c->out.p + r0, nreads, bi > 0);
        if (rc) return rc;
    }
    c->ev_valid = true;
    SK_HIP(hipMemcpyAsync(out, c->out.p, ob, hipMemcpyDeviceToHost, c->stream));
    if ((rc = finish_dtw_host(c))) return rc;
    return SK_OK;
}

// Stage a ragged float64 batch: samples -> c->sig, zero-based offsets -> c->off.
// Returns the total sample count in *total and the longest read in *maxlen.
// centi: `sig` holds int32 centi-units (sk_tsv_parse_centi) -- half the bytes over PCIe; the float64 image
// (c / 100.0 = float("ddd.dd"), sk_synth.hip k_centi_to_f64) is made on the device, and what follows is the same.
static int stage_ragged_f64(sk_ctx *c, const void *sig_any, const int64_t *off, int32_t nreads,
                            int64_t *total, int64_t *maxlen, bool centi = false)
{
    const double *sig = (const double *)sig_any;
    if (!sig || !off) return sk_fail(SK_ERR_INVALID, "NULL sig/off");
    std::vector<int64_t> rel((size_t)nreads + 1);
    int64_t mx = 0;
    for (int32_t r = 0; r <= nreads; r++) rel[r] = off[r] - off[0];
    for (int32_t r = 0; r < nreads; r++) {
        const int64_t n = rel[r + 1] - rel[r];
        if (n < 0 || n > 0x7fffff00) return sk_fail(SK_ERR_INVALID, "bad length for read %d", r);
        if (n > mx) mx = n;
    }
    *total = rel[nreads];
    *maxlen = mx;
    int rc;
    if ((rc = sk_reserve(c, &c->sig, (size_t)(*total > 0 ? *total : 1) * sizeof(double)))) return rc;
    if ((rc = sk_reserve(c, &c->off, rel.size() * sizeof(int64_t)))) return rc;
    if (centi) {
        if ((rc = sk_reserve(c, &c->misc, (size_t)(*total > 0 ? *total : 1) * sizeof(int32_t) + 16))) return rc;
        SK_HIP(hipMemcpyAsync(c->misc.p, (const int32_t *)sig_any + off[0], (size_t)*total * sizeof(int32_t),
                              hipMemcpyHostToDevice, c->stream));
        if ((rc = sk_launch_centi_to_f64(c, (const int32_t *)c->misc.p, *total, (double *)c->sig.p))) return rc;
    } else
        SK_HIP(hipMemcpyAsync(c->sig.p, sig + off[0], (size_t)*total * sizeof(double), hipMemcpyHostToDevice, c->stream));
    SK_HIP(hipMemcpyAsync(c->off.p, rel.data(), rel.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));        // rel goes out of scope
    return SK_OK;
}

// device-resident core of the float64 MotifSeq path: d_sig / d_off (zero based, nreads + 1) are device pointers.
// Filter + statistics once, then one DTW launch set per motif (nmotifs >= 1; motif k = motifs + motif_off[k], its records
// go to d_out + k * out_stride).
static int motifseq_multi_dev_f64(sk_ctx *c, const double *d_sig, const int64_t *d_off, int32_t nreads, int64_t total,
                                  int64_t maxlen, const double *motifs, const int32_t *motif_off, int32_t nmotifs,
                                  int32_t scale_mode, int32_t scale_low, int32_t scale_hi, sk_hit *d_out, int64_t out_stride)
{
    int rc;
    if ((rc = sk_reserve(c, &c->comp, (size_t)(total > 0 ? total : 1) * sizeof(double)))) return rc;
    if ((rc = sk_reserve(c, &c->prep, (size_t)nreads * sizeof(sk_prep)))) return rc;
    SK_HIP(hipEventRecord(c->ev[0], c->stream));
    c->f64_stream = 0;
    if (scale_mode == SK_SCALE_MEDMAD && sk_f64_fast_applies(maxlen, 1.0)) {
        c->f64_stream = 1;
        // streaming statistics (sk_f64stat.hip), then the general kernel over the (almost always empty) list of reads
        // whose median / MAD bin it could not resolve
        if ((rc = sk_reserve(c, &c->retry, ((size_t)nreads + 16) * sizeof(int32_t)))) return rc;
        int32_t *retry = (int32_t *)c->retry.p;
        rc = sk_launch_f64_stats(c, d_sig, d_off, nullptr, nreads, maxlen, (double)scale_low, (double)scale_hi, SK_PREP_MEDMAD,
                                 0.0, (sk_prep *)c->prep.p, nullptr, 0, nullptr, retry, (double *)c->comp.p);
        if (rc) return rc;
        rc = sk_launch_prep_f64_listed(c, d_sig, d_off, retry + 1, retry, nreads < 2 * c->num_cu ? nreads : 2 * c->num_cu,
                                       (double)scale_low, (double)scale_hi, SK_PREP_MEDMAD, 0.0, (double *)c->comp.p, 0,
                                       (sk_prep *)c->prep.p, nullptr, 0);
    } else {
        rc = sk_launch_prep_f64(c, d_sig, d_off, nreads, (double)scale_low,
                                (double)scale_hi, scale_mode == SK_SCALE_MEDMAD ? SK_PREP_MEDMAD : SK_PREP_ZSCALE,
                                0.0, (double *)c->comp.p, (sk_prep *)c->prep.p, nullptr, 0);
    }
    if (rc) return rc;
    SK_HIP(hipEventRecord(c->ev[1], c->stream));
    for (int32_t k = 0; k < nmotifs; k++) {
        sk_sdtw_args a;
        a.feed = SK_FEED_F64_NORM; a.samples = c->comp.p; a.stride = 0; a.off = d_off;
        a.samples_raw = d_sig;                          // (reads the filter left whole are not copied: SK_IFLAG_INPLACE)
        a.prep = (const sk_prep *)c->prep.p; a.nreads = nreads; a.motif = motifs + motif_off[k];
        a.nmotif = motif_off[k + 1] - motif_off[k];
        a.out = d_out + (size_t)k * (size_t)out_stride; a.last_row = nullptr; a.max_len = maxlen; a.force_single = 0;
        a.accumulate = k > 0 ? 1 : 0;
        if ((rc = sk_launch_sdtw(c, &a))) return rc;
    }
    c->ev_valid = true;
    return SK_OK;
}

static int motifseq_dev_f64(sk_ctx *c, const double *d_sig, const int64_t *d_off, int32_t nreads, int64_t total,
                            int64_t maxlen, const double *motif, int32_t nmotif, int32_t scale_mode,
                            int32_t scale_low, int32_t scale_hi, sk_hit *d_out)
{
    const int32_t moff[2] = {0, nmotif};
    return motifseq_multi_dev_f64(c, d_sig, d_off, nreads, total, maxlen, motif, moff, 1, scale_mode, scale_low, scale_hi,
                                  d_out, nreads);
}

int sk_motifseq_batch_f64(const double *sig, const int64_t *off, int32_t nreads,
                          const double *motif, int32_t nmotif, int32_t scale_mode,
                          int32_t scale_low, int32_t scale_hi, sk_hit *out)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (nreads < 0) return sk_fail(SK_ERR_INVALID, "nreads < 0");
    if (!motif || nmotif <= 0) return sk_fail(SK_ERR_INVALID, "empty motif");
    if (scale_mode != SK_SCALE_MEDMAD && scale_mode != SK_SCALE_ZSCALE)
        return sk_fail(SK_ERR_INVALID, "unknown scale mode %d", scale_mode);
    if (nreads == 0) return SK_OK;
    if (!out) return sk_fail(SK_ERR_INVALID, "NULL out");
    int64_t total, maxlen;
    int rc = stage_ragged_f64(c, sig, off, nreads, &total, &maxlen);
    if (rc) return rc;
    if ((rc = sk_reserve(c, &c->out, (size_t)nreads * sizeof(sk_hit)))) return rc;
    rc = motifseq_dev_f64(c, (const double *)c->sig.p, (const int64_t *)c->off.p, nreads, total, maxlen, motif, nmotif,
                          scale_mode, scale_low, scale_hi, (sk_hit *)c->out.p);
    if (rc) return rc;
    SK_HIP(hipMemcpyAsync(out, c->out.p, (size_t)nreads * sizeof(sk_hit), hipMemcpyDeviceToHost, c->stream));
    if ((rc = finish_dtw_host(c))) return rc;
    return SK_OK;
}

// Several motifs against the same ragged float64 batch (the `for name in m_order` loop of MotifSeq.py:436 on pA input):
// the batch is staged once, filter + statistics run once, one DTW launch set per motif.  out is [nmotifs][nreads].
static int motifseq_multi_batch_ragged(const void *sig, bool centi, const int64_t *off, int32_t nreads,
                                       const double *motifs, const int32_t *motif_off, int32_t nmotifs,
                                       int32_t scale_mode, int32_t scale_low, int32_t scale_hi, sk_hit *out);
int sk_motifseq_multi_batch_f64(const double *sig, const int64_t *off, int32_t nreads,
                                const double *motifs, const int32_t *motif_off, int32_t nmotifs,
                                int32_t scale_mode, int32_t scale_low, int32_t scale_hi, sk_hit *out)
{
    return motifseq_multi_batch_ragged(sig, false, off, nreads, motifs, motif_off, nmotifs, scale_mode, scale_low, scale_hi, out);
}
int sk_motifseq_multi_batch_centi(const int32_t *centi, const int64_t *off, int32_t nreads,
                                  const double *motifs, const int32_t *motif_off, int32_t nmotifs,
                                  int32_t scale_mode, int32_t scale_low, int32_t scale_hi, sk_hit *out)
{
    return motifseq_multi_batch_ragged(centi, true, off, nreads, motifs, motif_off, nmotifs, scale_mode, scale_low, scale_hi, out);
}
static int motifseq_multi_batch_ragged(const void *sig, bool centi, const int64_t *off, int32_t nreads,
                                       const double *motifs, const int32_t *motif_off, int32_t nmotifs,
                                       int32_t scale_mode, int32_t scale_low, int32_t scale_hi, sk_hit *out)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (nreads < 0) return sk_fail(SK_ERR_INVALID, "nreads < 0");
    int rc = check_multi(motifs, motif_off, nmotifs, scale_mode);
    if (rc) return rc;
    if (nreads == 0) return SK_OK;
    if (!out) return sk_fail(SK_ERR_INVALID, "NULL out");
    int64_t total, maxlen;
    if ((rc = stage_ragged_f64(c, sig, off, nreads, &total, &maxlen, centi))) return rc;
    const size_t ob = (size_t)nreads * (size_t)nmotifs * sizeof(sk_hit);
    if ((rc = sk_reserve(c, &c->out, ob))) return rc;
    rc = motifseq_multi_dev_f64(c, (const double *)c->sig.p, (const int64_t *)c->off.p, nreads, total, maxlen, motifs,
                                motif_off, nmotifs, scale_mode, scale_low, scale_hi, (sk_hit *)c->out.p, nreads);
    if (rc) return rc;
    SK_HIP(hipMemcpyAsync(out, c->out.p, ob, hipMemcpyDeviceToHost, c->stream));
    if ((rc = finish_dtw_host(c))) return rc;
    return SK_OK;
}

int sk_motifseq_dev_f64(const double *d_sig, const int64_t *d_off, int32_t nreads, int64_t total, int64_t max_len,
                        const double *motif, int32_t nmotif, int32_t scale_mode,
                        int32_t scale_low, int32_t scale_hi, sk_hit *d_out)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (nreads < 0 || total < 0 || max_len < 0 || max_len > 0x7fffff00) return sk_fail(SK_ERR_INVALID, "bad sizes");
    if (!motif || nmotif <= 0) return sk_fail(SK_ERR_INVALID, "empty motif");
    if (scale_mode != SK_SCALE_MEDMAD && scale_mode != SK_SCALE_ZSCALE)
        return sk_fail(SK_ERR_INVALID, "unknown scale mode %d", scale_mode);
    if (nreads == 0) return SK_OK;
    if (!d_sig || !d_off || !d_out) return sk_fail(SK_ERR_INVALID, "NULL sig/off/out");
    return motifseq_dev_f64(c, d_sig, d_off, nreads, total, max_len, motif, nmotif, scale_mode, scale_low, scale_hi,
                            d_out);
}

// ------------------------------------------------------------------ mlpy boundary (pre-normalised f64)
int sk_dtw_subsequence_batch(const double *x, int32_t nx, const double *y, const int64_t *off,
                             int32_t nreads, sk_hit *out)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (nreads < 0 || nx <= 0 || !x) return sk_fail(SK_ERR_INVALID, "bad query");
    if (nreads == 0) return SK_OK;
    if (!y || !off || !out) return sk_fail(SK_ERR_INVALID, "NULL y/off/out");
    const int64_t total = off[nreads] - off[0];
    if (total < 0) return sk_fail(SK_ERR_INVALID, "offsets not increasing");
    int64_t maxlen = 0;
    for (int32_t r = 0; r < nreads; r++) {
        const int64_t n = off[r + 1] - off[r];
        if (n < 0 || n > 0x7fffff00) return sk_fail(SK_ERR_INVALID, "bad length for read %d", r);
        if (n > maxlen) maxlen = n;
    }
    int rc;
    if ((rc = sk_reserve(c, &c->sig, (size_t)(total > 0 ? total : 1) * sizeof(double)))) return rc;
    if ((rc = sk_reserve(c, &c->off, (size_t)(nreads + 1) * sizeof(int64_t)))) return rc;
    if ((rc = sk_reserve(c, &c->out, (size_t)nreads * sizeof(sk_hit)))) return rc;
    std::vector<int64_t> rel((size_t)nreads + 1);
    for (int32_t r = 0; r <= nreads; r++) rel[r] = off[r] - off[0];
    SK_HIP(hipMemcpyAsync(c->sig.p, y + off[0], (size_t)total * sizeof(double), hipMemcpyHostToDevice, c->stream));
    SK_HIP(hipMemcpyAsync(c->off.p, rel.data(), rel.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));        // rel is about to go out of scope
    sk_sdtw_args a;
    a.feed = SK_FEED_F64_RAW; a.samples = c->sig.p; a.stride = 0; a.off = (const int64_t *)c->off.p;
    a.prep = nullptr; a.nreads = nreads; a.motif = x; a.nmotif = nx; a.out = (sk_hit *)c->out.p;
    a.last_row = nullptr; a.max_len = maxlen; a.force_single = 0;
    SK_HIP(hipEventRecord(c->ev[0], c->stream));
    SK_HIP(hipEventRecord(c->ev[1], c->stream));
    if ((rc = sk_launch_sdtw(c, &a))) return rc;
    c->ev_valid = true;
    SK_HIP(hipMemcpyAsync(out, c->out.p, (size_t)nreads * sizeof(sk_hit), hipMemcpyDeviceToHost, c->stream));
    if ((rc = finish_dtw_host(c))) return rc;
    return SK_OK;
}

int sk_dtw_subsequence(const double *x, int32_t nx, const double *y, int32_t ny,
                       double *dist, int32_t *start, int32_t *end, double *cost_last_row)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!x || !y || nx <= 0 || ny <= 0) return sk_fail(SK_ERR_INVALID, "empty x or y");
    int rc;
    if ((rc = sk_reserve(c, &c->sig, (size_t)ny * sizeof(double)))) return rc;
    if ((rc = sk_reserve(c, &c->off, 2 * sizeof(int64_t)))) return rc;
    if ((rc = sk_reserve(c, &c->out, sizeof(sk_hit)))) return rc;
    if (cost_last_row && (rc = sk_reserve(c, &c->misc, (size_t)ny * sizeof(double)))) return rc;
    const int64_t rel[2] = {0, ny};
    SK_HIP(hipMemcpyAsync(c->sig.p, y, (size_t)ny * sizeof(double), hipMemcpyHostToDevice, c->stream));
    SK_HIP(hipMemcpyAsync(c->off.p, rel, sizeof rel, hipMemcpyHostToDevice, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));
    sk_sdtw_args a;
    a.feed = SK_FEED_F64_RAW; a.samples = c->sig.p; a.stride = 0; a.off = (const int64_t *)c->off.p;
    a.prep = nullptr; a.nreads = 1; a.motif = x; a.nmotif = nx; a.out = (sk_hit *)c->out.p;
    a.last_row = cost_last_row ? (double *)c->misc.p : nullptr;
    a.max_len = ny; a.force_single = 1;
    if ((rc = sk_launch_sdtw(c, &a))) return rc;
    sk_hit h;
    SK_HIP(hipMemcpyAsync(&h, c->out.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    if (cost_last_row)
        SK_HIP(hipMemcpyAsync(cost_last_row, c->misc.p, (size_t)ny * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));
    if (dist) *dist = h.dist;
    if (start) *start = h.start;
    if (end) *end = h.end;
    return SK_OK;
}

int sk_normalise_i16(const int16_t *sig, int32_t len, int32_t scale_mode,
                     int32_t scale_low, int32_t scale_hi, double *out, int32_t *n_out)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (len < 0 || (len && (!sig || !out))) return sk_fail(SK_ERR_INVALID, "bad arguments");
    if (scale_mode != SK_SCALE_MEDMAD && scale_mode != SK_SCALE_ZSCALE)
        return sk_fail(SK_ERR_INVALID, "unknown scale mode %d", scale_mode);
    if (len == 0) { if (n_out) *n_out = 0; return SK_OK; }
    clamp_limits(&scale_low, &scale_hi);
    const int64_t stride = ((int64_t)len + 7) & ~7ll;
    int rc;
    if ((rc = sk_reserve(c, &c->sig, (size_t)stride * 2))) return rc;
    if ((rc = sk_reserve(c, &c->len, sizeof(int32_t)))) return rc;
    if ((rc = sk_reserve(c, &c->comp, (size_t)stride * 2))) return rc;
    if ((rc = sk_reserve(c, &c->prep, sizeof(sk_prep)))) return rc;
    if ((rc = sk_reserve(c, &c->misc, (size_t)len * sizeof(double)))) return rc;
    SK_HIP(hipMemcpyAsync(c->sig.p, sig, (size_t)len * 2, hipMemcpyHostToDevice, c->stream));
    SK_HIP(hipMemcpyAsync(c->len.p, &len, sizeof len, hipMemcpyHostToDevice, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));
    rc = sk_launch_prep_i16(c, (const int16_t *)c->sig.p, stride, (const int32_t *)c->len.p, 1, scale_low,
                            scale_hi, scale_mode == SK_SCALE_MEDMAD ? SK_PREP_MEDMAD : SK_PREP_ZSCALE, 0.0,
                            (int16_t *)c->comp.p, (sk_prep *)c->prep.p, nullptr, 0);
    if (rc) return rc;
    hipLaunchKernelGGL(k_normalise_i16, dim3(64), dim3(256), 0, c->stream, (const int16_t *)c->comp.p,
                       (const sk_prep *)c->prep.p, (double *)c->misc.p);
    SK_HIP(hipGetLastError());
    sk_prep pr;
    SK_HIP(hipMemcpyAsync(&pr, c->prep.p, sizeof pr, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));
    if (pr.n > 0)
        SK_HIP(hipMemcpy(out, c->misc.p, (size_t)pr.n * sizeof(double), hipMemcpyDeviceToHost));
    if (n_out) *n_out = pr.n;
    return SK_OK;
}

int sk_normalise_f64(const double *sig, int32_t len, int32_t scale_mode,
                     int32_t scale_low, int32_t scale_hi, double *out, int32_t *n_out)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (len < 0 || (len && (!sig || !out))) return sk_fail(SK_ERR_INVALID, "bad arguments");
    if (scale_mode != SK_SCALE_MEDMAD && scale_mode != SK_SCALE_ZSCALE)
        return sk_fail(SK_ERR_INVALID, "unknown scale mode %d", scale_mode);
    if (len == 0) { if (n_out) *n_out = 0; return SK_OK; }
    const int64_t off[2] = {0, len};
    int64_t total, maxlen;
    int rc = stage_ragged_f64(c, sig, off, 1, &total, &maxlen);
    if (rc) return rc;
    if ((rc = sk_reserve(c, &c->comp, (size_t)len * sizeof(double)))) return rc;
    if ((rc = sk_reserve(c, &c->prep, sizeof(sk_prep)))) return rc;
    if ((rc = sk_reserve(c, &c->misc, (size_t)len * sizeof(double)))) return rc;
    rc = sk_launch_prep_f64(c, (const double *)c->sig.p, (const int64_t *)c->off.p, 1, (double)scale_low,
                            (double)scale_hi, scale_mode == SK_SCALE_MEDMAD ? SK_PREP_MEDMAD : SK_PREP_ZSCALE,
                            0.0, (double *)c->comp.p, (sk_prep *)c->prep.p, nullptr, 0);
    if (rc) return rc;
    hipLaunchKernelGGL(k_normalise_f64, dim3(64), dim3(256), 0, c->stream, (const double *)c->comp.p,
                       (const sk_prep *)c->prep.p, (double *)c->misc.p);
    SK_HIP(hipGetLastError());
    sk_prep pr;
    SK_HIP(hipMemcpyAsync(&pr, c->prep.p, sizeof pr, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));
    if (pr.n > 0)
        SK_HIP(hipMemcpy(out, c->misc.p, (size_t)pr.n * sizeof(double), hipMemcpyDeviceToHost));
    if (n_out) *n_out = pr.n;
    return SK_OK;
}

// ------------------------------------------------------------------ segmenter
int sk_segment_dev_i16(const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads,
                       const sk_seg_params *p, int32_t *d_segs, int32_t *d_nsegs, int32_t max_segs)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    int rc = check_i16(d_sig, stride, d_len, nreads);
    if (rc) return rc;
    if ((rc = check_seg_params(p))) return rc;
    if (max_segs <= 0) return sk_fail(SK_ERR_INVALID, "max_segs must be positive");
    if (nreads == 0) return SK_OK;
    if (!d_segs || !d_nsegs) return sk_fail(SK_ERR_INVALID, "NULL segs/nsegs");
    int32_t lo = p->lim_low, hi = p->lim_hi;
    clamp_limits(&lo, &hi);
    const int64_t words = (stride + 63) / 64;
    if ((rc = sk_reserve(c, &c->prep, (size_t)nreads * sizeof(sk_prep)))) return rc;
    // slots past nsegs[r] read as zero, whatever the buffer held before
    SK_HIP(hipMemsetAsync(d_segs, 0, (size_t)nreads * 2 * (size_t)max_segs * sizeof(int32_t), c->stream));
    if (sk_segment_fast_applies(d_sig, stride, lo, hi, p->std_scale)) {
        // streaming statistics (sk_segstat.hip): reads of up to 4 096 samples, exact integer sums, certified
        // integer thresholds; the numpy-order kernel redoes the (almost always empty) list of uncertified reads
        const size_t mb = (size_t)nreads * (size_t)sk_segment_fast_row16(stride) * 16;
        if ((rc = sk_reserve(c, &c->mask, mb))) return rc;
        if ((rc = sk_reserve(c, &c->retry, ((size_t)nreads + 16) * sizeof(int32_t)))) return rc;
        rc = sk_launch_segment_fast(c, d_sig, stride, d_len, nreads, p, lo, hi, (sk_prep *)c->prep.p, c->mask.p,
                                    (int32_t *)c->retry.p, d_segs, d_nsegs, max_segs);
        if (rc) return rc;
        c->ev_valid = true;
        return SK_OK;
    }
    if ((rc = sk_reserve(c, &c->comp, (size_t)nreads * (size_t)stride * sizeof(int16_t)))) return rc;
    if ((rc = sk_reserve(c, &c->mask, (size_t)nreads * (size_t)words * sizeof(uint64_t)))) return rc;
    SK_HIP(hipEventRecord(c->ev[0], c->stream));
    rc = sk_launch_prep_i16(c, d_sig, stride, d_len, nreads, lo, hi, SK_PREP_SEGMENT, p->std_scale,
                            (int16_t *)c->comp.p, (sk_prep *)c->prep.p, (uint64_t *)c->mask.p, nreads);
    if (rc) return rc;
    SK_HIP(hipEventRecord(c->ev[1], c->stream));
    rc = sk_launch_segment_walk(c, (const uint64_t *)c->mask.p, nreads, nullptr, (const sk_prep *)c->prep.p,
                                nreads, p, d_segs, d_nsegs, max_segs);
    if (rc) return rc;
    c->ev_valid = true;
    return SK_OK;
}

int sk_segment_batch_i16(const int16_t *sig, int64_t stride, const int32_t *len, int32_t nreads,
                         const sk_seg_params *p, int32_t *segs, int32_t *nsegs, int32_t max_segs)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    int rc = check_i16(sig, stride, len, nreads);
    if (rc) return rc;
    if ((rc = check_len_host(len, nreads, stride))) return rc;
    if ((rc = check_seg_params(p))) return rc;
    if (max_segs <= 0) return sk_fail(SK_ERR_INVALID, "max_segs must be positive");
    if (nreads == 0) return SK_OK;
    if (!segs || !nsegs) return sk_fail(SK_ERR_INVALID, "NULL segs/nsegs");
    const size_t sb = (size_t)nreads * (size_t)stride * sizeof(int16_t);
    const size_t gb = (size_t)nreads * 2 * (size_t)max_segs * sizeof(int32_t);
    if ((rc = sk_reserve(c, &c->sig, sb))) return rc;
    if ((rc = sk_reserve(c, &c->len, (size_t)nreads * sizeof(int32_t)))) return rc;
    if ((rc = sk_reserve(c, &c->out, gb))) return rc;
    if ((rc = sk_reserve(c, &c->out2, (size_t)nreads * sizeof(int32_t)))) return rc;
    const SubBatches B = sub_batches(nreads, stride);
    if (B.n > 1 && (rc = second_stream(c))) return rc;
    for (int32_t bi = 0; bi < B.n; bi++) {
        const int32_t r0 = bi * B.per;
        const int32_t nr = (nreads - r0 < B.per) ? nreads - r0 : B.per;
        if (nr <= 0) break;
        int16_t *d_sig = (int16_t *)c->sig.p + (size_t)r0 * (size_t)stride;
        int32_t *d_len = (int32_t *)c->len.p + r0;
        hipStream_t cs = (B.n > 1) ? c->stream2 : c->stream;
        SK_HIP(hipMemcpyAsync(d_sig, sig + (size_t)r0 * (size_t)stride, (size_t)nr * (size_t)stride * sizeof(int16_t),
                              hipMemcpyHostToDevice, cs));
        SK_HIP(hipMemcpyAsync(d_len, len + r0, (size_t)nr * sizeof(int32_t), hipMemcpyHostToDevice, cs));
        if (B.n > 1) {
            SK_HIP(hipEventRecord(c->ev_chunk[bi & 7], cs));
            SK_HIP(hipStreamWaitEvent(c->stream, c->ev_chunk[bi & 7], 0));
        }
        rc = sk_segment_dev_i16(d_sig, stride, d_len, nr, p, (int32_t *)c->out.p + (size_t)r0 * 2 * (size_t)max_segs,
                                (int32_t *)c->out2.p + r0, max_segs);
        if (rc) return rc;
    }
    SK_HIP(hipMemcpyAsync(segs, c->out.p, gb, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(hipMemcpyAsync(nsegs, c->out2.p, (size_t)nreads * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));
    for (int32_t r = 0; r < nreads; r++)
        if (nsegs[r] > max_segs)
            return sk_fail(SK_ERR_OVERFLOW, "read %d has %d segments, max_segs is %d", r, nsegs[r], max_segs);
    return SK_OK;
}

// device-resident core of the float64 segmenter path (d_off zero based; d_segs zeroed here)
static int segment_dev_f64(sk_ctx *c, const double *d_sig, const int64_t *d_off, int32_t nreads, int64_t total,
                           int64_t maxlen, const sk_seg_params *p, int32_t *d_segs, int32_t *d_nsegs, int32_t max_segs,
                           const int32_t *d_rlen = nullptr)
{
    int rc;
    const int64_t words = (maxlen + 63) / 64 > 0 ? (maxlen + 63) / 64 : 1;
    const size_t gb = (size_t)nreads * 2 * (size_t)max_segs * sizeof(int32_t);
    c->f64_stream = 0;
    if (sk_f64_fast_applies(maxlen, p->std_scale)) {
        c->f64_stream = 1;
        // streaming statistics with certified comparisons (sk_f64stat.hip), the numpy-order kernel over the (almost
        // always empty) list of uncertified reads, then the run-hopping walk of the int16 path over the same masks
        const int row16 = sk_f64_row16(maxlen > 0 ? maxlen : 1);
        const int grid = nreads < 2 * c->num_cu ? nreads : 2 * c->num_cu;
        const int64_t srow = (maxlen + 7) & ~(int64_t)7;
        if ((rc = sk_reserve(c, &c->prep, (size_t)nreads * sizeof(sk_prep)))) return rc;
        if ((rc = sk_reserve(c, &c->mask, (size_t)nreads * (size_t)row16 * 16))) return rc;
        if ((rc = sk_reserve(c, &c->len, (size_t)nreads * sizeof(int32_t)))) return rc;
        if ((rc = sk_reserve(c, &c->retry, ((size_t)nreads + 16) * sizeof(int32_t)))) return rc;
        if ((rc = sk_reserve(c, &c->comp, (size_t)grid * (size_t)(srow > 0 ? srow : 8) * sizeof(double)))) return rc;
        int32_t *retry = (int32_t *)c->retry.p;
        SK_HIP(hipMemsetAsync(d_segs, 0, gb, c->stream));
        SK_HIP(hipEventRecord(c->ev[0], c->stream));
        rc = sk_launch_f64_stats(c, d_sig, d_off, d_rlen, nreads, maxlen, (double)p->lim_low, (double)p->lim_hi, SK_PREP_SEGMENT,
                                 p->std_scale, (sk_prep *)c->prep.p, c->mask.p, row16, (int32_t *)c->len.p, retry, nullptr);
        if (rc) return rc;
        rc = sk_launch_prep_f64_listed(c, d_sig, d_off, retry + 1, retry, grid, (double)p->lim_low, (double)p->lim_hi,
                                       SK_PREP_SEGMENT, p->std_scale, (double *)c->comp.p, srow, (sk_prep *)c->prep.p,
                                       c->mask.p, row16, d_rlen);
        if (rc) return rc;
        SK_HIP(hipEventRecord(c->ev[1], c->stream));
        rc = sk_launch_seg_walk_masks(c, c->mask.p, row16, (const int32_t *)c->len.p, nreads, p, d_segs, d_nsegs, max_segs);
        if (rc) return rc;
        c->ev_valid = true;
        return SK_OK;
    }
    if ((rc = sk_reserve(c, &c->comp, (size_t)(total > 0 ? total : 1) * sizeof(double)))) return rc;
    if ((rc = sk_reserve(c, &c->prep, (size_t)nreads * sizeof(sk_prep)))) return rc;
    if ((rc = sk_reserve(c, &c->mask, (size_t)nreads * (size_t)words * sizeof(uint64_t)))) return rc;
    SK_HIP(hipMemsetAsync(d_segs, 0, gb, c->stream));
    SK_HIP(hipEventRecord(c->ev[0], c->stream));
    rc = sk_launch_prep_f64(c, d_sig, d_off, nreads, (double)p->lim_low,
                            (double)p->lim_hi, SK_PREP_SEGMENT, p->std_scale, (double *)c->comp.p,
                            (sk_prep *)c->prep.p, (uint64_t *)c->mask.p, nreads, d_rlen);
    if (rc) return rc;
    SK_HIP(hipEventRecord(c->ev[1], c->stream));
    rc = sk_launch_segment_walk(c, (const uint64_t *)c->mask.p, nreads, nullptr, (const sk_prep *)c->prep.p,
                                nreads, p, d_segs, d_nsegs, max_segs);
    if (rc) return rc;
    c->ev_valid = true;
    return SK_OK;
}

static int segment_batch_ragged(const void *sig, bool centi, const int64_t *off, const int32_t *len, int32_t nreads,
                                const sk_seg_params *p, int32_t *segs, int32_t *nsegs, int32_t max_segs);
int sk_segment_batch_f64_len(const double *sig, const int64_t *off, const int32_t *len, int32_t nreads,
                             const sk_seg_params *p, int32_t *segs, int32_t *nsegs, int32_t max_segs)
{
    return segment_batch_ragged(sig, false, off, len, nreads, p, segs, nsegs, max_segs);
}
int sk_segment_batch_centi_len(const int32_t *centi, const int64_t *off, const int32_t *len, int32_t nreads,
                               const sk_seg_params *p, int32_t *segs, int32_t *nsegs, int32_t max_segs)
{
    return segment_batch_ragged(centi, true, off, len, nreads, p, segs, nsegs, max_segs);
}
static int segment_batch_ragged(const void *sig, bool centi, const int64_t *off, const int32_t *len, int32_t nreads,
                                const sk_seg_params *p, int32_t *segs, int32_t *nsegs, int32_t max_segs)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (nreads < 0) return sk_fail(SK_ERR_INVALID, "nreads < 0");
    int rc = check_seg_params(p);
    if (rc) return rc;
    if (max_segs <= 0) return sk_fail(SK_ERR_INVALID, "max_segs must be positive");
    if (nreads == 0) return SK_OK;
    if (!segs || !nsegs) return sk_fail(SK_ERR_INVALID, "NULL segs/nsegs");
    int64_t total, maxlen;
    if ((rc = stage_ragged_f64(c, sig, off, nreads, &total, &maxlen, centi))) return rc;
    const int32_t *d_rlen = nullptr;
    if (len) {                                      // the caller's sig[:Num] cut: read r is its first len[r] samples
        for (int32_t r = 0; r < nreads; r++)
            if (len[r] < 0 || (int64_t)len[r] > off[r + 1] - off[r])
                return sk_fail(SK_ERR_INVALID, "len[%d] = %d is outside [0, %lld]", r, len[r], (long long)(off[r + 1] - off[r]));
        // (a buffer of its own: segment_dev_f64 hands c->len to the statistics kernel as the place for the lengths the
        // walk reads, and the numpy-order redo looks at the cut again afterwards)
        if ((rc = sk_reserve(c, &c->rlen, (size_t)nreads * sizeof(int32_t)))) return rc;
        SK_HIP(hipMemcpyAsync(c->rlen.p, len, (size_t)nreads * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        d_rlen = (const int32_t *)c->rlen.p;
        // the cut, not the slot, decides the mask row size and which statistics kernel runs (-n 3000 on 40 000-sample
        // lines takes the 4 096-sample kernel)
        maxlen = 0;
        for (int32_t r = 0; r < nreads; r++) if (len[r] > maxlen) maxlen = len[r];
    }
    const size_t gb = (size_t)nreads * 2 * (size_t)max_segs * sizeof(int32_t);
    if ((rc = sk_reserve(c, &c->out, gb))) return rc;
    if ((rc = sk_reserve(c, &c->out2, (size_t)nreads * sizeof(int32_t)))) return rc;
    rc = segment_dev_f64(c, (const double *)c->sig.p, (const int64_t *)c->off.p, nreads, total, maxlen, p,
                         (int32_t *)c->out.p, (int32_t *)c->out2.p, max_segs, d_rlen);
    if (rc) return rc;
    SK_HIP(hipMemcpyAsync(segs, c->out.p, gb, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(hipMemcpyAsync(nsegs, c->out2.p, (size_t)nreads * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));
    for (int32_t r = 0; r < nreads; r++)
        if (nsegs[r] > max_segs)
            return sk_fail(SK_ERR_OVERFLOW, "read %d has %d segments, max_segs is %d", r, nsegs[r], max_segs);
    return SK_OK;
}

int sk_segment_batch_f64(const double *sig, const int64_t *off, int32_t nreads, const sk_seg_params *p,
                         int32_t *segs, int32_t *nsegs, int32_t max_segs)
{
    return sk_segment_batch_f64_len(sig, off, nullptr, nreads, p, segs, nsegs, max_segs);
}

// Raw reads through the pA route: what segmenter.py does with fast5 / slow5 input unless --raw_signal is given
// (segmenter.py:345-349, 366-370: np.round(convert_to_pA_numpy(sig, digitisation, range, offset), 2) with range first
// cut to two decimals, float("{0:.2f}".format(range)), :385).  calib[3 r ..] = digitisation, offset, range of read r
// (what a fast5 / BLOW5 record carries) -> cal2[2 r ..] = {offset, raw_unit = range / digitisation}.
int sk_pa_calib(const double *calib, int32_t nreads, double *cal2)
{
    if (nreads < 0 || (nreads > 0 && (!calib || !cal2))) return sk_fail(SK_ERR_INVALID, "NULL calib / cal2");
    for (int32_t r = 0; r < nreads; r++) {
        const double dig = calib[3 * r], ofs = calib[3 * r + 1], rng = calib[3 * r + 2];
        char txt[512];
        snprintf(txt, sizeof txt, "%.2f", rng);              // float("{0:.2f}".format(range))
        cal2[2 * r] = ofs;
        cal2[2 * r + 1] = strtod(txt, nullptr) / dig;        // raw_unit = range / digitisation
    }
    return SK_OK;
}

// Device-resident core.  Since round 6 the values stay int16: the pA conversion is a monotone map of the sample, so
// limits, median, std and the two thresholds are found in the raw domain (k_seg_stats<.., PA>, sk_segstat.hip: 2 bytes a
// sample instead of 8; reads it cannot certify are redone from their float64 values in numpy's order).  Rows the
// streaming kernel does not take (stride not a multiple of 8, unaligned): the float64 image of every row, then the
// float64 segmenter path -- what every call did before round 6.
static int segment_dev_i16_pa(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads,
                              const double *d_cal2, const sk_seg_params *p, int32_t *d_segs, int32_t *d_nsegs, int32_t max_segs)
{
    int rc;
    c->pa_raw = 0;
    if (sk_segment_pa_applies(d_sig, stride, p->std_scale)) {
        c->pa_raw = 1;
        const size_t mb = (size_t)nreads * (size_t)sk_segment_fast_row16(stride) * 16;
        if ((rc = sk_reserve(c, &c->prep, (size_t)nreads * sizeof(sk_prep)))) return rc;
        if ((rc = sk_reserve(c, &c->mask, mb))) return rc;
        if ((rc = sk_reserve(c, &c->retry, ((size_t)nreads + 16) * sizeof(int32_t)))) return rc;
        if ((rc = sk_reserve(c, &c->comp, (size_t)c->num_cu * (size_t)stride * sizeof(double)))) return rc;
        SK_HIP(hipMemsetAsync(d_segs, 0, (size_t)nreads * 2 * (size_t)max_segs * sizeof(int32_t), c->stream));
        rc = sk_launch_segment_fast(c, d_sig, stride, d_len, nreads, p, p->lim_low, p->lim_hi, (sk_prep *)c->prep.p, c->mask.p,
                                    (int32_t *)c->retry.p, d_segs, d_nsegs, max_segs, d_cal2, (double *)c->comp.p);
        if (rc) return rc;
        c->ev_valid = true;
        return SK_OK;
    }
    // every read in a slot of `stride` doubles; the cut to len[r] is the float64 path's per-read length
    const int64_t total = (int64_t)nreads * stride;
    c->pa_off_host.resize((size_t)nreads + 1);
    for (int32_t r = 0; r <= nreads; r++) c->pa_off_host[r] = (int64_t)r * stride;
    if ((rc = sk_reserve(c, &c->sig, (size_t)(total > 0 ? total : 1) * sizeof(double)))) return rc;
    if ((rc = sk_reserve(c, &c->off, c->pa_off_host.size() * sizeof(int64_t)))) return rc;
    SK_HIP(hipMemcpyAsync(c->off.p, c->pa_off_host.data(), c->pa_off_host.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));                 // (the vector may be resized by the next call)
    rc = sk_launch_rows_to_pa(c, d_sig, stride, nreads, (const int64_t *)c->off.p, d_cal2, (double *)c->sig.p);
    if (rc) return rc;
    return segment_dev_f64(c, (const double *)c->sig.p, (const int64_t *)c->off.p, nreads, total, stride, p,
                           d_segs, d_nsegs, max_segs, d_len);
}

// Reads of the most recent sk_segment_*_i16_pa call (its last sub-batch) that the raw-domain kernel could not certify and
// that were redone from their float64 values; -1 when that call expanded every read to float64 instead.
int sk_last_pa_retries(void)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!c->pa_raw) return -1;
    SK_HIP(hipStreamSynchronize(c->stream));
    int total = 0;
    for (const int32_t *ptr : c->pa_retry_ptrs) {
        int32_t v = 0;
        SK_HIP(hipMemcpy(&v, ptr, sizeof v, hipMemcpyDeviceToHost));
        total += v;
    }
    return total;
}

int sk_segment_dev_i16_pa(const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads, const double *d_cal2,
                          const sk_seg_params *p, int32_t *d_segs, int32_t *d_nsegs, int32_t max_segs)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    int rc = check_i16(d_sig, stride, d_len, nreads);
    if (rc) return rc;
    if ((rc = check_seg_params(p))) return rc;
    if (max_segs <= 0) return sk_fail(SK_ERR_INVALID, "max_segs must be positive");
    if (nreads == 0) return SK_OK;
    if (!d_cal2 || !d_segs || !d_nsegs) return sk_fail(SK_ERR_INVALID, "NULL cal2/segs/nsegs");
    return segment_dev_i16_pa(c, d_sig, stride, d_len, nreads, d_cal2, p, d_segs, d_nsegs, max_segs);
}

int sk_segment_batch_i16_pa(const int16_t *sig, int64_t stride, const int32_t *len, int32_t nreads, const double *calib,
                            const sk_seg_params *p, int32_t *segs, int32_t *nsegs, int32_t max_segs)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    int rc = check_i16(sig, stride, len, nreads);
    if (rc) return rc;
    if ((rc = check_len_host(len, nreads, stride))) return rc;
    if ((rc = check_seg_params(p))) return rc;
    if (max_segs <= 0) return sk_fail(SK_ERR_INVALID, "max_segs must be positive");
    if (nreads == 0) return SK_OK;
    if (!calib || !segs || !nsegs) return sk_fail(SK_ERR_INVALID, "NULL calib/segs/nsegs");
    std::vector<double> cal((size_t)nreads * 2);
    if ((rc = sk_pa_calib(calib, nreads, cal.data()))) return rc;
    const size_t sb = (size_t)nreads * (size_t)stride * sizeof(int16_t);
    const size_t gb = (size_t)nreads * 2 * (size_t)max_segs * sizeof(int32_t);
    if ((rc = sk_reserve(c, &c->misc, sb))) return rc;
    if ((rc = sk_reserve(c, &c->len, (size_t)nreads * sizeof(int32_t)))) return rc;
    if ((rc = sk_reserve(c, &c->pacal, cal.size() * sizeof(double)))) return rc;
    if ((rc = sk_reserve(c, &c->out, gb))) return rc;
    if ((rc = sk_reserve(c, &c->out2, (size_t)nreads * sizeof(int32_t)))) return rc;
    SK_HIP(hipMemcpyAsync(c->pacal.p, cal.data(), cal.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    SK_HIP(hipMemcpyAsync(c->len.p, len, (size_t)nreads * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    // rows in sub-batches, the copy of one beside the kernels of the one before (as sk_segment_batch_i16)
    const SubBatches B = sub_batches(nreads, stride);
    if (B.n > 1 && (rc = second_stream(c))) return rc;
    SK_HIP(hipStreamSynchronize(c->stream));                 // cal goes out of scope; len / cal are there for every sub-batch
    // (the float64 fallback keeps its lengths in c->len as well: it gets a copy of its own)
    const bool raw_domain = sk_segment_pa_applies(c->misc.p, stride, p->std_scale);
    if (!raw_domain) {
        if ((rc = sk_reserve(c, &c->rlen, (size_t)nreads * sizeof(int32_t)))) return rc;
        SK_HIP(hipMemcpyAsync(c->rlen.p, len, (size_t)nreads * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    }
    for (int32_t bi = 0; bi < (raw_domain ? B.n : 1); bi++) {
        const int32_t per = raw_domain ? B.per : nreads;
        const int32_t r0 = bi * per;
        const int32_t nr = (nreads - r0 < per) ? nreads - r0 : per;
        if (nr <= 0) break;
        int16_t *d_sig = (int16_t *)c->misc.p + (size_t)r0 * (size_t)stride;
        hipStream_t cs = (raw_domain && B.n > 1) ? c->stream2 : c->stream;
        SK_HIP(hipMemcpyAsync(d_sig, sig + (size_t)r0 * (size_t)stride, (size_t)nr * (size_t)stride * sizeof(int16_t),
                              hipMemcpyHostToDevice, cs));
        if (cs != c->stream) {
            SK_HIP(hipEventRecord(c->ev_chunk[bi & 7], cs));
            SK_HIP(hipStreamWaitEvent(c->stream, c->ev_chunk[bi & 7], 0));
        }
        rc = segment_dev_i16_pa(c, d_sig, stride, (const int32_t *)(raw_domain ? c->len.p : c->rlen.p) + r0, nr,
                                (const double *)c->pacal.p + 2 * (size_t)r0, p,
                                (int32_t *)c->out.p + (size_t)r0 * 2 * (size_t)max_segs, (int32_t *)c->out2.p + r0, max_segs);
        if (rc) return rc;
    }
    SK_HIP(hipMemcpyAsync(segs, c->out.p, gb, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(hipMemcpyAsync(nsegs, c->out2.p, (size_t)nreads * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));
    for (int32_t r = 0; r < nreads; r++)
        if (nsegs[r] > max_segs)
            return sk_fail(SK_ERR_OVERFLOW, "read %d has %d segments, max_segs is %d", r, nsegs[r], max_segs);
    return SK_OK;
}

int sk_segment_dev_f64(const double *d_sig, const int64_t *d_off, int32_t nreads, int64_t total, int64_t max_len,
                       const sk_seg_params *p, int32_t *d_segs, int32_t *d_nsegs, int32_t max_segs)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (nreads < 0 || total < 0 || max_len < 0 || max_len > 0x7fffff00) return sk_fail(SK_ERR_INVALID, "bad sizes");
    int rc = check_seg_params(p);
    if (rc) return rc;
    if (max_segs <= 0) return sk_fail(SK_ERR_INVALID, "max_segs must be positive");
    if (nreads == 0) return SK_OK;
    if (!d_sig || !d_off || !d_segs || !d_nsegs) return sk_fail(SK_ERR_INVALID, "NULL sig/off/segs/nsegs");
    return segment_dev_f64(c, d_sig, d_off, nreads, total, max_len, p, d_segs, d_nsegs, max_segs);
}

// ------------------------------------------------------------------ dRNA adapter segmenter
// ------------------------------------------------------------------ dRNA --signal branch (rolling mean)
// device-resident cores of the two dRNA branches (d_sig / d_len / outputs are device pointers)
static int drna_roll_dev(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads,
                         const sk_roll_params *p, int32_t *d_xy, int32_t *d_found)
{
    int rc;
    int32_t lo = p->lim_low, hi = p->lim_hi;
    clamp_limits(&lo, &hi);
    const int64_t words = (stride + 63) / 64;
    const size_t sb = (size_t)nreads * (size_t)stride * sizeof(int16_t);
    if ((rc = sk_reserve(c, &c->prep, (size_t)nreads * sizeof(sk_prep)))) return rc;
    if ((rc = sk_reserve(c, &c->mask, (size_t)nreads * (size_t)words * 2 * sizeof(uint64_t)))) return rc;
    uint64_t *below = (uint64_t *)c->mask.p, *above = below + (size_t)nreads * (size_t)words;
    // one look (round 5): a workgroup per read, prefix sums in LDS -- rows of up to ~35 000 samples, w < 65 536
    // ... or as a stream, a wavefront per read with certified thresholds (windows of up to 12 000 samples, rows of any length)
    const bool stream = sk_roll_stream_ok(stride, p->w, lo, hi) && sk_tune("SK_ROLL_ONE_LOOK") == nullptr;
    if ((stream || sk_roll_one_lds(stride, p->w)) && sk_tune("SK_ROLL_TWO_KERNELS") == nullptr && sk_tune("SK_DRNA_STEP") == nullptr) {
        if (stream && (rc = sk_reserve(c, &c->misc, ((size_t)nreads + 2) * sizeof(int32_t)))) return rc;
        SK_HIP(hipEventRecord(c->ev[0], c->stream));
        if (stream)
            rc = sk_launch_roll_stream(c, d_sig, stride, d_len, nreads, lo, hi, p->w, p->std_scale, (sk_prep *)c->prep.p,
                                       below, above, (int32_t *)c->misc.p);
        else
            rc = sk_launch_roll_one(c, d_sig, stride, d_len, nreads, lo, hi, p->w, p->std_scale, (sk_prep *)c->prep.p,
                                    below, above);
        if (rc) return rc;
        SK_HIP(hipEventRecord(c->ev[1], c->stream));
        if ((rc = sk_launch_roll_walk(c, below, stream ? below + 1 : above, (const sk_prep *)c->prep.p, nreads, p, d_xy,
                                      d_found, stream ? words : 0))) return rc;
        c->ev_valid = true;
        return SK_OK;
    }
    if ((rc = sk_reserve(c, &c->comp, sb))) return rc;
    // prefix sums of the filtered samples: 4 bytes each when every window sum fits 31 bits (sk_launch_roll_stats)
    const size_t pbytes = (p->w < 65536 && sk_tune("SK_DRNA_STEP") == nullptr) ? sizeof(uint32_t) : sizeof(int64_t);
    if ((rc = sk_reserve(c, &c->misc, (size_t)nreads * (size_t)(stride + 1) * pbytes))) return rc;
    SK_HIP(hipEventRecord(c->ev[0], c->stream));
    // filter + order-preserving compaction (the medmad kernel: its statistics are not used here)
    rc = sk_launch_prep_i16(c, d_sig, stride, d_len, nreads, lo, hi,
                            SK_PREP_MEDMAD, 0.0, (int16_t *)c->comp.p, (sk_prep *)c->prep.p, nullptr, 0);
    if (rc) return rc;
    rc = sk_launch_roll_stats(c, (const int16_t *)c->comp.p, stride, (sk_prep *)c->prep.p, nreads, p->w,
                              p->std_scale, (int64_t *)c->misc.p, below, above);
    if (rc) return rc;
    SK_HIP(hipEventRecord(c->ev[1], c->stream));
    rc = sk_launch_roll_walk(c, below, above, (const sk_prep *)c->prep.p, nreads, p, d_xy, d_found);
    if (rc) return rc;
    c->ev_valid = true;
    return SK_OK;
}

static int drna_segment_dev(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads,
                            const sk_drna_params *p, int32_t *d_segs, int32_t *d_nsegs, int32_t max_segs)
{
    int rc;
    int32_t lo = p->lim_low, hi = p->lim_hi;
    clamp_limits(&lo, &hi);
    const int64_t words = (stride + 63) / 64;
    const size_t sb = (size_t)nreads * (size_t)stride * sizeof(int16_t);
    const size_t gb = (size_t)nreads * 2 * (size_t)max_segs * sizeof(int32_t);
    if ((rc = sk_reserve(c, &c->comp, sb))) return rc;
    if ((rc = sk_reserve(c, &c->prep, (size_t)nreads * sizeof(sk_prep)))) return rc;
    if ((rc = sk_reserve(c, &c->mask, (size_t)nreads * (size_t)words * sizeof(uint64_t)))) return rc;
    SK_HIP(hipMemsetAsync(d_segs, 0, gb, c->stream));
    SK_HIP(hipEventRecord(c->ev[0], c->stream));
    rc = sk_launch_prep_i16(c, d_sig, stride, d_len, nreads, lo, hi,
                            SK_PREP_DRNA, p->std_scale, (int16_t *)c->comp.p, (sk_prep *)c->prep.p,
                            (uint64_t *)c->mask.p, nreads, p->t_start, p->t_end);
    if (rc) return rc;
    SK_HIP(hipEventRecord(c->ev[1], c->stream));
    rc = sk_launch_drna_walk(c, (const uint64_t *)c->mask.p, nreads, (const sk_prep *)c->prep.p, nreads, p,
                             d_segs, d_nsegs, max_segs);
    if (rc) return rc;
    c->ev_valid = true;
    return SK_OK;
}

static int check_roll(const sk_roll_params *p)
{
    if (!p) return sk_fail(SK_ERR_INVALID, "NULL sk_roll_params");
    if (p->w <= 0) return sk_fail(SK_ERR_INVALID, "the rolling window w must be positive");
    return SK_OK;
}

static int check_drna(const sk_drna_params *p, int32_t max_segs)
{
    if (!p) return sk_fail(SK_ERR_INVALID, "NULL sk_drna_params");
    if (p->w <= 0) return sk_fail(SK_ERR_INVALID, "w must be positive (the scan takes c %% w)");
    if (p->t_start < 0 || p->t_end < p->t_start) return sk_fail(SK_ERR_INVALID, "bad statistics window");
    if (max_segs <= 0) return sk_fail(SK_ERR_INVALID, "max_segs must be positive");
    return SK_OK;
}

int sk_drna_roll_dev_i16(const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads,
                         const sk_roll_params *p, int32_t *d_xy, int32_t *d_found)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    int rc = check_i16(d_sig, stride, d_len, nreads);
    if (rc) return rc;
    if ((rc = check_roll(p))) return rc;
    if (nreads == 0) return SK_OK;
    if (!d_xy || !d_found) return sk_fail(SK_ERR_INVALID, "NULL xy/found");
    return drna_roll_dev(c, d_sig, stride, d_len, nreads, p, d_xy, d_found);
}

int sk_drna_segment_dev_i16(const int16_t *d_sig, int64_t stride, const int32_t *d_len, int32_t nreads,
                            const sk_drna_params *p, int32_t *d_segs, int32_t *d_nsegs, int32_t max_segs)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    int rc = check_i16(d_sig, stride, d_len, nreads);
    if (rc) return rc;
    if ((rc = check_drna(p, max_segs))) return rc;
    if (nreads == 0) return SK_OK;
    if (!d_segs || !d_nsegs) return sk_fail(SK_ERR_INVALID, "NULL segs/nsegs");
    return drna_segment_dev(c, d_sig, stride, d_len, nreads, p, d_segs, d_nsegs, max_segs);
}

int sk_drna_roll_batch_i16(const int16_t *sig, int64_t stride, const int32_t *len, int32_t nreads,
                           const sk_roll_params *p, int32_t *xy, int32_t *found)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    int rc = check_i16(sig, stride, len, nreads);
    if (rc) return rc;
    if ((rc = check_len_host(len, nreads, stride))) return rc;
    if ((rc = check_roll(p))) return rc;
    if (nreads == 0) return SK_OK;
    if (!xy || !found) return sk_fail(SK_ERR_INVALID, "NULL xy/found");
    const size_t sb = (size_t)nreads * (size_t)stride * sizeof(int16_t);
    if ((rc = sk_reserve(c, &c->sig, sb))) return rc;
    if ((rc = sk_reserve(c, &c->len, (size_t)nreads * sizeof(int32_t)))) return rc;
    if ((rc = sk_reserve(c, &c->out, (size_t)nreads * 2 * sizeof(int32_t)))) return rc;
    if ((rc = sk_reserve(c, &c->out2, (size_t)nreads * sizeof(int32_t)))) return rc;
    SK_HIP(hipMemcpyAsync(c->sig.p, sig, sb, hipMemcpyHostToDevice, c->stream));
    SK_HIP(hipMemcpyAsync(c->len.p, len, (size_t)nreads * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    if ((rc = drna_roll_dev(c, (const int16_t *)c->sig.p, stride, (const int32_t *)c->len.p, nreads, p,
                            (int32_t *)c->out.p, (int32_t *)c->out2.p))) return rc;
    SK_HIP(hipMemcpyAsync(xy, c->out.p, (size_t)nreads * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    SK_HIP(hipMemcpyAsync(found, c->out2.p, (size_t)nreads * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));
    return SK_OK;
}

int sk_drna_segment_batch_i16(const int16_t *sig, int64_t stride, const int32_t *len, int32_t nreads,
                              const sk_drna_params *p, int32_t *segs, int32_t *nsegs, int32_t max_segs)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    int rc = check_i16(sig, stride, len, nreads);
    if (rc) return rc;
    if ((rc = check_len_host(len, nreads, stride))) return rc;
    if ((rc = check_drna(p, max_segs))) return rc;
    if (nreads == 0) return SK_OK;
    if (!segs || !nsegs) return sk_fail(SK_ERR_INVALID, "NULL segs/nsegs");
    const size_t sb = (size_t)nreads * (size_t)stride * sizeof(int16_t);
    const size_t gb = (size_t)nreads * 2 * (size_t)max_segs * sizeof(int32_t);
    if ((rc = sk_reserve(c, &c->sig, sb))) return rc;
    if ((rc = sk_reserve(c, &c->len, (size_t)nreads * sizeof(int32_t)))) return rc;
    if ((rc = sk_reserve(c, &c->out, gb))) return rc;
    if ((rc = sk_reserve(c, &c->out2, (size_t)nreads * sizeof(int32_t)))) return rc;
    SK_HIP(hipMemcpyAsync(c->sig.p, sig, sb, hipMemcpyHostToDevice, c->stream));
    SK_HIP(hipMemcpyAsync(c->len.p, len, (size_t)nreads * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    if ((rc = drna_segment_dev(c, (const int16_t *)c->sig.p, stride, (const int32_t *)c->len.p, nreads, p,
                               (int32_t *)c->out.p, (int32_t *)c->out2.p, max_segs))) return rc;
    SK_HIP(hipMemcpyAsync(segs, c->out.p, gb, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(hipMemcpyAsync(nsegs, c->out2.p, (size_t)nreads * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));
    for (int32_t r = 0; r < nreads; r++)
        if (nsegs[r] > max_segs)
            return sk_fail(SK_ERR_OVERFLOW, "read %d has %d segments, max_segs is %d", r, nsegs[r], max_segs);
    return SK_OK;
}

// ------------------------------------------------------------------ bench input
int sk_synth_squiggles_dev(int16_t *d_sig, int64_t stride, int32_t nreads, int32_t nsamples,
                           uint64_t seed, const double *motif, int32_t nmotif)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!d_sig || stride < nsamples || nreads < 0 || nsamples < 0)
        return sk_fail(SK_ERR_INVALID, "bad arguments");
    const int16_t *d_m = nullptr;
    if (motif && nmotif > 0) {
        std::vector<int16_t> mi((size_t)nmotif);
        for (int i = 0; i < nmotif; i++) {
            double v = rint(motif[i] * 93.4 + 511.0);
            mi[i] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
        }
        int rc = sk_reserve(c, &c->misc, mi.size() * 2);
        if (rc) return rc;
        SK_HIP(hipMemcpyAsync(c->misc.p, mi.data(), mi.size() * 2, hipMemcpyHostToDevice, c->stream));
        SK_HIP(hipStreamSynchronize(c->stream));
        d_m = (const int16_t *)c->misc.p;
    }
    int rc = sk_launch_synth(c, d_sig, stride, nreads, nsamples, seed, d_m, nmotif);
    if (rc) return rc;
    SK_HIP(hipStreamSynchronize(c->stream));
    return SK_OK;
}

int sk_synth_variant_dev(int16_t *d_sig, int64_t stride, int32_t nreads, int32_t nsamples, uint64_t seed,
                         const double *motif, int32_t nmotif, const sk_synth_opts *o)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!d_sig || !o || stride < nsamples || nreads < 0 || nsamples < 0 || o->row0 < 0)
        return sk_fail(SK_ERR_INVALID, "bad arguments");
    int rc;
    if (o->tmpl && o->ntmpl > 0) {                       // windows of a measured squiggle + noise
        if ((rc = sk_reserve(c, &c->misc, (size_t)o->ntmpl * 2))) return rc;
        SK_HIP(hipMemcpyAsync(c->misc.p, o->tmpl, (size_t)o->ntmpl * 2, hipMemcpyHostToDevice, c->stream));
        SK_HIP(hipStreamSynchronize(c->stream));
        rc = sk_launch_synth_windows(c, d_sig, stride, nreads, nsamples, seed, o->row0, (const int16_t *)c->misc.p,
                                     o->ntmpl, (float)o->tmpl_noise);
    } else {
        const int16_t *d_m = nullptr;
        if (motif && nmotif > 0) {
            std::vector<int16_t> mi((size_t)nmotif);
            for (int i = 0; i < nmotif; i++) {
                double v = rint(motif[i] * 93.4 + 511.0);
                mi[i] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
            }
            if ((rc = sk_reserve(c, &c->misc, mi.size() * 2))) return rc;
            SK_HIP(hipMemcpyAsync(c->misc.p, mi.data(), mi.size() * 2, hipMemcpyHostToDevice, c->stream));
            SK_HIP(hipStreamSynchronize(c->stream));
            d_m = (const int16_t *)c->misc.p;
        }
        rc = sk_launch_synth(c, d_sig, stride, nreads, nsamples, seed, d_m, nmotif, o->row0, o->hit_permille,
                             o->stretch_permille, o->stretch);
    }
    if (rc) return rc;
    SK_HIP(hipStreamSynchronize(c->stream));
    return SK_OK;
}

// The float64 pA image of an int16 batch, as SquigglePull.py:183-189,238-240 writes it (bench / test input for the
// float64 entry points): d_out[nreads * nsamples] doubles, d_off[nreads + 1] zero-based offsets.
int sk_synth_pa_dev(const int16_t *d_raw, int64_t stride, int32_t nreads, int32_t nsamples,
                    double offset, double range, double digitisation, double *d_out, int64_t *d_off)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!d_raw || !d_out || !d_off || stride < nsamples || nreads < 0 || nsamples < 0 || !(digitisation > 0))
        return sk_fail(SK_ERR_INVALID, "bad arguments");
    int rc = sk_launch_raw_to_pa(c, d_raw, stride, nreads, nsamples, offset, range / digitisation, d_out, d_off);
    if (rc) return rc;
    SK_HIP(hipStreamSynchronize(c->stream));
    return SK_OK;
}

// Reads of the most recent float64 call (segmenter or MotifSeq medmad) that the streaming statistics kernel handed
// to the numpy-order kernel (diagnostic); -1 when that call did not take the streaming kernel.
int sk_last_f64_retries(void)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!c->f64_stream || !c->retry.p) return -1;
    int32_t n = 0;
    SK_HIP(hipStreamSynchronize(c->stream));
    SK_HIP(hipMemcpy(&n, c->retry.p, sizeof n, hipMemcpyDeviceToHost));
    return n;
}

// mlpy.dtw_subsequence(x, y) in the reference's own C arithmetic, NaN / inf included (k_dtw_cref above): what
// `MotifSeq.py --strict-compat` prints for reads whose MAD is 0.  Full cost matrix in device memory, one lane.
int sk_dtw_subsequence_cref(const double *x, int32_t nx, const double *y, int32_t ny,
                            double *dist, int32_t *start, int32_t *end)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!x || !y || nx <= 0 || ny <= 0) return sk_fail(SK_ERR_INVALID, "empty x or y");
    const size_t cells = (size_t)nx * (size_t)ny;
    if (cells > ((size_t)1 << 28)) return sk_fail(SK_ERR_UNSUPPORTED, "%d x %d cost matrix is over 2 GB", nx, ny);
    int rc;
    if ((rc = sk_reserve(c, &c->sig, ((size_t)nx + (size_t)ny) * sizeof(double)))) return rc;
    if ((rc = sk_reserve(c, &c->misc, cells * sizeof(double)))) return rc;
    if ((rc = sk_reserve(c, &c->out, sizeof(sk_hit)))) return rc;
    double *d_x = (double *)c->sig.p, *d_y = d_x + nx;
    SK_HIP(hipMemcpyAsync(d_x, x, (size_t)nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
    SK_HIP(hipMemcpyAsync(d_y, y, (size_t)ny * sizeof(double), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_dtw_cref, dim3(1), dim3(64), 0, c->stream, (const double *)d_x, nx, (const double *)d_y, ny,
                       (double *)c->misc.p, (sk_hit *)c->out.p);
    SK_HIP(hipGetLastError());
    sk_hit h;
    SK_HIP(hipMemcpyAsync(&h, c->out.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));
    if (dist) *dist = h.dist;
    if (start) *start = h.start;
    if (end) *end = h.end;
    return SK_OK;
}

} // extern "C"
