// sk_runtime.hip -- device contexts, error reporting, device memory.
// One context per SLOT, created by sk_init() (slot == device) or sk_init_slot(); each host
// thread is bound to one slot (thread_local), so a multi-GPU host drives one thread per GPU
// or, as bench.py also does, one process per GPU.  Several slots may serve the same device
// (sk_init_slot): that is how the N > 1 code paths are exercised on a one-GPU box.
#include "sk_common.h"
#include <string>
#include <string.h>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static thread_local char  g_err[512] = "";
static thread_local int   g_cur = -1;
static sk_ctx             g_ctx[SK_MAX_DEVICES];
static std::mutex         g_mu;
static std::recursive_mutex g_ctx_mu[SK_MAX_DEVICES];      // one per context slot (sk_ctx_guard)

void sk_ctx_lock(sk_ctx *c)   { g_ctx_mu[c - g_ctx].lock(); }
void sk_ctx_unlock(sk_ctx *c) { g_ctx_mu[c - g_ctx].unlock(); }

int sk_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

sk_ctx *sk_cur(void)
{
    if (g_cur < 0 || !g_ctx[g_cur].ready) {
        sk_fail(SK_ERR_NO_DEVICE, "no device bound: call sk_init(device) first (no CPU fallback exists)");
        return nullptr;
    }
    if (hipSetDevice(g_ctx[g_cur].device) != hipSuccess) {
        sk_fail(SK_ERR_NO_DEVICE, "hipSetDevice(%d) failed", g_ctx[g_cur].device);
        return nullptr;
    }
    return &g_ctx[g_cur];
}

sk_ctx *sk_ctx_of(int device) { return (device >= 0 && device < SK_MAX_DEVICES) ? &g_ctx[device] : nullptr; }
int sk_bound_device(void) { return g_cur; }

int sk_reserve(sk_ctx *c, sk_buf *b, size_t bytes)
{
    (void)c;
    if (bytes <= b->cap) return SK_OK;
    if (b->p) { SK_HIP(hipFree(b->p)); b->p = nullptr; b->cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b->p, want);
    if (e != hipSuccess) {
        b->p = nullptr;
        return sk_fail(SK_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b->cap = want;
    return SK_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The tuning surface: every environment switch of the library, the values the parity test flips it to
// (tests/test_gpu_tuning.py runs each one alone and compares the records byte for byte), what it does.
// ---------------------------------------------------------------------------------------------------------------
struct sk_tunable { const char *name, *test_values, *what; };
static const sk_tunable SK_TUNABLES[] = {
    {"SK_DTW_SCHEME",         "full exact2", "DTW scheme: full = one exact pass, exact2 = two exact passes (default: fixed-point screening + certified window)"},
    {"SK_DTW_QL",             "8 16 64",     "lanes per read of the screening scheme (default: by motif length and batch size)"},
    {"SK_DTW_NOFUSE",         "1",           "filter + medmad as their own kernel instead of the screening pass's prologue"},
    {"SK_DTW_FORCE_RETRY_PM", "100 1000",    "per-mille of reads sent through the exact retry whatever the window pass certified (sensitivity runs)"},
    {"SK_DTW_SPAN",           "150 300",     "look-back of the window pass in columns (one tier)"},
    {"SK_DTW_SPAN2",          "0 500",       "look-back of the window pass's second tier (0: none)"},
    {"SK_DTW_CK",             "64 256",      "steps between checkpoints of the screening pass (multiple of 64)"},
    {"SK_DTW_NOSORT",         "1",           "window passes take the reads in file order"},
    {"SK_DTW_SORT_MIN",       "1",           "window passes sort chunks of at least this many reads"},
    {"SK_DTW_NO_EARLY",       "1",           "no early exact retry beside the window passes"},
    {"SK_DTW_NO_SIBLINGS",    "1",           "reads whose candidate columns fall into two clusters take the exact pass (no second window)"},
    {"SK_DTW_NOGUARD",        "1",           "screening scheme without its run-time guard (premise test in the window pass, audit, gated exact fallback): A/B cost runs"},
    {"SK_DTW_AUDIT_PERIOD",   "0 1 64",      "the audit re-runs one read in this many with the exact pass (default 4096; 0: no audit)"},
    {"SK_DTW_HOLE",           "qerr1 fma64 fma64x", "tests: a known precision hole put back (E = 1; the fma sample image for float64 reads, with / without the image-error guard) -- the guard has to notice and the records must not change"},
    {"SK_DTW_SCRATCH_MB",     "8 64",        "checkpoint scratch budget in MB (small values force many chunks)"},
    {"SK_DTW_NO_SMALL",       "1",           "small batches keep the batch lane layout"},
    {"SK_DTW_SMALL_MAX",      "0 100000",    "largest batch that spreads a read over 64 lanes"},
    {"SK_PREP_BLOCK",         "1",           "medmad statistics by the workgroup-per-read kernel"},
    {"SK_PREP_ROUNDS",        "1 3",         "grid rounds of the persistent statistics kernels"},
    {"SK_PREP_PERCU",         "1 2",         "workgroups per CU of the persistent statistics kernels"},
    {"SK_SEG_OLD",            "1",           "segmenter: numpy-order statistics kernels for every read"},
    {"SK_SEG_DELTA_SCALE",    "1e13",        "segmenter: certification margin multiplier (large: every read takes the numpy-order redo)"},
    {"SK_SEG_PA_F64",         "1",           "segmenter, raw reads through the pA conversion: expand to float64 and take the float64 kernels instead of the raw-domain kernel"},
    {"SK_SEG_NO_WG",          "1",           "segmenter, reads of 4 097 .. 65 536 samples: the wavefront-per-read statistics kernel (two looks at a read) instead of the workgroup-per-read one"},
    {"SK_SEG_WG_ALL",         "1",           "segmenter: the workgroup-per-read statistics kernel for every row of 4 097 .. 65 536 samples (default: where it is faster)"},
    {"SK_SEG_OCC",            "7 8",         "segmenter statistics kernel: waves per SIMD the registers are sized for"},
    {"SK_SEG_CHUNKS",         "2 8",         "segmenter: chunks of a large batch (walk of one beside the statistics of the next)"},
    {"SK_WALK_STEP",          "1",           "segmenter walk: per-sample straight-line step instead of run hopping"},
    {"SK_WALK_GENERAL",       "1",           "segmenter walk: general step (corrector test live)"},
    {"SK_WALK_SYNC",          "1",           "segmenter walk: run hopping with the 64 lanes of a wavefront on the same word (k_seg_walk3)"},
    {"SK_WALK_OWNPASS",       "1",           "segmenter walk: finds the quiet stretches and anchors in a pass of its own instead of taking the statistics kernel's hints"},
    {"SK_WALK_NOJUMP",        "1",           "segmenter walk: every run is hopped through, no jumps between the stretches of quiet entries"},
    {"SK_WALK_NOWAVE",        "1",           "segmenter walk: a lane per read also for long rows / small batches (no wavefront-per-read walk)"},
    {"SK_WALK_WAVE_MAXREADS", "0 1000000",   "segmenter walk: batches of up to this many reads take the wavefront-per-read walk (default: 16384, 131072 for rows beyond 4 096 samples)"},
    {"SK_DRNA_STEP",          "1",           "dRNA_segmenter, both branches: the per-sample scans instead of the scans by runs / by transitions"},
    {"SK_ROLL_TWO_KERNELS",   "1",           "dRNA_segmenter rolling-mean branch: filter kernel + prefix sums through HBM instead of the one-look kernel (prefix sums in LDS)"},
    {"SK_ROLL_ONE_LOOK",      "1",           "dRNA_segmenter rolling-mean branch: the workgroup-per-read kernel in numpy's order for every read instead of the streaming kernel with certified thresholds"},
    {"SK_ROLL_DELTA_SCALE",   "1e13",        "dRNA_segmenter rolling-mean branch: multiplier of the certification margin (large: every read is redone in numpy's order)"},
    {"SK_INGEST_MB",          "1 4",         "sub-batch size of the host entry points in MB"},
    {"SK_F64_OLD",            "1",           "float64 reads: numpy-order statistics kernel for every read"},
    {"SK_F64_LONG_LOOKS",     "1",           "float64 reads of 4 097 .. 40 960 samples: the window-by-window kernel (three to five looks at a read) instead of the workgroup-per-read one (one look)"},
};

const char *sk_tune(const char *name)
{
    const char *on = getenv("SK_TUNING");
    if (!on || on[0] != '1') return nullptr;
    for (const sk_tunable &t : SK_TUNABLES)
        if (strcmp(t.name, name) == 0) return getenv(name);
    return nullptr;                                     // not in the table: not a switch
}

static thread_local int g_dtw_shrink = 0;       // halvings of the scratch budget after a failed allocation
void sk_dtw_scratch_shrink(int reset) { g_dtw_shrink = reset ? 0 : (g_dtw_shrink < 12 ? g_dtw_shrink + 1 : 12); }

int64_t sk_dtw_chunk_reads(size_t per_read, int64_t nreads)
{
    // 64 GB of the 288: whole-chip "rounds" of equally long wavefronts make a launch's last, partly filled round
    // pure loss, so few large chunks beat many small ones (C4, 1 M reads: 75.3 ms per step with 12 GB = 4 chunks,
    // 72.9 ms with one chunk).  A caller short of memory gets smaller chunks (sk_dtw_scratch_shrink).
    size_t budget = (size_t)64 << 30;
    int64_t floor_reads = 1024;
    if (const char *e = sk_tune("SK_DTW_SCRATCH_MB")) {
        const long v = atol(e);
        if (v > 0) { budget = (size_t)v << 20; floor_reads = 64; }
    }
    budget >>= g_dtw_shrink;
    int64_t chunk = (int64_t)(budget / (per_read ? per_read : 1));
    if (chunk < floor_reads) chunk = floor_reads;
    if (chunk > nreads) chunk = nreads;
    return chunk > 0 ? chunk : 1;
}

extern "C" {

// The tuning table as text: one line per switch, "name<TAB>test values<TAB>description".  Returns the bytes needed.
int sk_tunables(char *buf, int cap)
{
    std::string out;
    for (const sk_tunable &t : SK_TUNABLES) { out += t.name; out += '\t'; out += t.test_values; out += '\t'; out += t.what; out += '\n'; }
    if (buf && cap > 0) { const size_t n = out.size() < (size_t)cap - 1 ? out.size() : (size_t)cap - 1; memcpy(buf, out.data(), n); buf[n] = 0; }
    return (int)out.size() + 1;
}

const char *sk_version(void) { return "squigglekit-hip 0.1.0 (gfx950)"; }
const char *sk_last_error(void) { return g_err; }

int sk_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        sk_fail(SK_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
        return 0;
    }
    return n;
}

int sk_init(int device) { return sk_init_slot(device, device); }

int sk_init_slot(int slot, int device)
{
    int n = sk_device_count();
    if (n <= 0) return sk_fail(SK_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= n || device >= SK_MAX_DEVICES)
        return sk_fail(SK_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    if (slot < 0 || slot >= SK_MAX_DEVICES)
        return sk_fail(SK_ERR_INVALID, "context slot %d out of range (0..%d)", slot, SK_MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lk(g_mu);
    sk_ctx *c = &g_ctx[slot];
    sk_ctx_guard c_lock(c);
    if (c->ready && c->device != device)
        return sk_fail(SK_ERR_INVALID, "context slot %d already serves device %d", slot, c->device);
    SK_HIP(hipSetDevice(device));
    if (!c->ready) {
        hipDeviceProp_t prop;
        SK_HIP(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return sk_fail(SK_ERR_NO_DEVICE, "device %d is %s; this build targets gfx950 only",
                           device, prop.gcnArchName);
        c->device = device;
        c->num_cu = prop.multiProcessorCount;
        SK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        for (int i = 0; i < 4; i++) SK_HIP(hipEventCreate(&c->ev[i]));
        c->ready = true;
    }
    g_cur = slot;
    return SK_OK;
}

static void free_buf(sk_buf *b) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }

int sk_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (int d = 0; d < SK_MAX_DEVICES; d++) {
        sk_ctx *c = &g_ctx[d];
        sk_ctx_guard c_lock(c);                            // (a call in flight on another thread finishes first)
        if (!c->ready) continue;
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        sk_buf *bufs[] = {&c->sig, &c->len, &c->off, &c->comp, &c->prep, &c->mask,
                          &c->motif, &c->out, &c->out2, &c->misc, &c->ckpt, &c->retry, &c->motifq, &c->lastq, &c->qflag,
                          &c->motif64, &c->commbuf, &c->dtwcnt, &c->wsoft, &c->wstate, &c->wrec, &c->motifw, &c->lsum, &c->wrecq, &c->order, &c->pacal, &c->seghints, &c->audit, &c->rlen, &c->sib, &c->sibout, &c->sibstate};
        for (sk_buf *b : bufs) free_buf(b);
        for (int i = 0; i < 4; i++) (void)hipEventDestroy(c->ev[i]);
        for (hipEvent_t e : c->evpool) (void)hipEventDestroy(e);
        if (c->stream2) {
            (void)hipStreamSynchronize(c->stream2);
            for (int i = 0; i < 9; i++) if (c->ev_chunk[i]) (void)hipEventDestroy(c->ev_chunk[i]);
            (void)hipStreamDestroy(c->stream2);
        }
        if (c->stream3) {
            (void)hipStreamSynchronize(c->stream3);
            for (int i = 0; i < 2; i++) if (c->ev_r[i]) (void)hipEventDestroy(c->ev_r[i]);
            (void)hipStreamDestroy(c->stream3);
        }
        // (the sibling-round events belong to the third stream's launches, not to the audit's: they exist whenever
        // stream3 does, also under SK_DTW_NOGUARD / SK_DTW_AUDIT_PERIOD=0 where stream4 never is created)
        for (int i = 0; i < 2; i++) if (c->ev_s[i]) (void)hipEventDestroy(c->ev_s[i]);
        if (c->stream4) {
            (void)hipStreamSynchronize(c->stream4);
            if (c->ev_a) (void)hipEventDestroy(c->ev_a);
            (void)hipStreamDestroy(c->stream4);
        }
        (void)hipStreamDestroy(c->stream);
        *c = sk_ctx();
    }
    g_cur = -1;
    return SK_OK;
}

int sk_sync(void)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    SK_HIP(hipStreamSynchronize(c->stream));
    return SK_OK;
}

int sk_device_name(char *buf, int cap)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!buf || cap <= 0) return sk_fail(SK_ERR_INVALID, "bad buffer");
    hipDeviceProp_t prop;
    SK_HIP(hipGetDeviceProperties(&prop, c->device));
    snprintf(buf, (size_t)cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return SK_OK;
}

// "0000:c1:00.0" of the bound device: what /sys/bus/pci/devices/<id>/local_cpulist and numa_node are keyed on
int sk_device_pci_bus_id(char *buf, int cap)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!buf || cap < 16) return sk_fail(SK_ERR_INVALID, "bad buffer");
    SK_HIP(hipDeviceGetPCIBusId(buf, cap, c->device));
    return SK_OK;
}

void *sk_dev_alloc(size_t bytes)
{
    sk_ctx *c = sk_cur();
    if (!c) return nullptr;
    sk_ctx_guard c_lock(c);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) {
        sk_fail(SK_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

int sk_dev_free(void *dptr)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (dptr) SK_HIP(hipFree(dptr));
    return SK_OK;
}

int sk_dev_upload(void *dst_dev, const void *src_host, size_t bytes)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (bytes && (!dst_dev || !src_host)) return sk_fail(SK_ERR_INVALID, "NULL pointer");
    SK_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));
    return SK_OK;
}

int sk_dev_download(void *dst_host, const void *src_dev, size_t bytes)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (bytes && (!dst_host || !src_dev)) return sk_fail(SK_ERR_INVALID, "NULL pointer");
    SK_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(hipStreamSynchronize(c->stream));
    return SK_OK;
}

int sk_last_kernel_ms(float *prep_ms, float *main_ms)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!c->ev_valid) return sk_fail(SK_ERR_INVALID, "no timed call yet");
    SK_HIP(hipEventSynchronize(c->ev[3]));
    float a = 0.f, b = 0.f;
    SK_HIP(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
    SK_HIP(hipEventElapsedTime(&b, c->ev[2], c->ev[3]));
    if (prep_ms) *prep_ms = a;
    if (main_ms) *main_ms = b;
    return SK_OK;
}

int sk_last_dtw_profile(float *dist_ms, int *dist_launches, float *start_ms, int *start_launches,
                        int *reads_per_launch)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    float a = 0.f, b = 0.f;
    int mx = 0;
    for (int i = 0; i < c->prof_chunks; i++) {
        hipEvent_t *ev = &c->evpool[3 * (size_t)i];
        SK_HIP(hipEventSynchronize(ev[2]));
        float t = 0.f;
        SK_HIP(hipEventElapsedTime(&t, ev[0], ev[1])); a += t;
        SK_HIP(hipEventElapsedTime(&t, ev[1], ev[2])); b += t;
        const int rr = c->prof_reads[i < 64 ? i : 63];
        if (rr > mx) mx = rr;
    }
    if (dist_ms) *dist_ms = a;
    if (start_ms) *start_ms = b;
    if (dist_launches) *dist_launches = c->prof_chunks;
    if (start_launches) *start_launches = c->prof_chunks;
    if (reads_per_launch) *reads_per_launch = mx;
    return SK_OK;
}

int sk_last_dtw_retries(void)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (c->retry_dev && c->dtwcnt.p) {              // the count stayed on the device: fetch it now
        int32_t n = 0;
        if (hipMemcpyAsync(&n, c->dtwcnt.p, sizeof n, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            return sk_fail(SK_ERR_HIP, "reading the retry count failed");
        c->last_retry = n;
    }
    return c->last_retry;
}

int sk_last_dtw_clock(double *ghz)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!ghz) return sk_fail(SK_ERR_INVALID, "NULL pointer");
    *ghz = 0.0;
    if (!(c->retry_dev && c->dtwcnt.p)) return SK_OK;
    unsigned long long t[2] = {0, 0};
    if (hipMemcpyAsync(t, (const char *)c->dtwcnt.p + 16, sizeof t, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        return sk_fail(SK_ERR_HIP, "reading the clock sample failed");
    if (t[1]) *ghz = (double)t[0] / ((double)t[1] / 100e6) / 1e9;
    return SK_OK;
}

// The guard counters of the last DTW call (sk_sdtw_dev.h SK_GUARD_*): out[0] premise violations found by the window
// pass, [1] reads audited by the exact pass, [2] audit mismatches, [3] reads whose sample image could not be bounded
// (exact pass by design), [4] alarm (= [0] + [2]), [5] 1 if the whole call was redone by the exact pass.  All zero
// for a call that did not take the screening scheme.
int sk_last_dtw_guard(int32_t *out)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!out) return sk_fail(SK_ERR_INVALID, "NULL pointer");
    memset(out, 0, 8 * sizeof(int32_t));
    if (!(c->retry_dev && c->dtwcnt.p)) return SK_OK;
    if (hipMemcpyAsync(out, (const int32_t *)c->dtwcnt.p + 8, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        return sk_fail(SK_ERR_HIP, "reading the guard counters failed");
    return SK_OK;
}

int sk_last_dtw_premise_violations(void)
{
    int32_t g[8];
    const int rc = sk_last_dtw_guard(g);
    return rc ? rc : g[0];
}

int sk_last_dtw_audit_mismatches(void)
{
    int32_t g[8];
    const int rc = sk_last_dtw_guard(g);
    return rc ? rc : g[2];
}

// out[0] = steps the wavefronts of the window pass (k_sdtw_w, every tier) ran in the last DTW call, out[1] = the steps their
// reads asked for, summed over the read groups (a wavefront's groups step together: out[0] * groups per wave >= out[1])
int sk_last_dtw_window_steps(uint64_t *out)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!out) return sk_fail(SK_ERR_INVALID, "NULL pointer");
    out[0] = out[1] = 0;
    if (!(c->retry_dev && c->dtwcnt.p && c->dtwcnt.cap >= 128)) return SK_OK;
    if (hipMemcpyAsync(out, (const char *)c->dtwcnt.p + 64, 16, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        return sk_fail(SK_ERR_HIP, "reading the window-step counters failed");
    return SK_OK;
}

int sk_last_dtw_tier2(void)
{
    sk_ctx *c = sk_cur();
    if (!c) return SK_ERR_NO_DEVICE;
    sk_ctx_guard c_lock(c);
    if (!(c->retry_dev && c->dtwcnt.p)) return 0;
    int32_t n = 0;
    if (hipMemcpyAsync(&n, (const int32_t *)c->dtwcnt.p + 1, sizeof n, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        return sk_fail(SK_ERR_HIP, "reading the second-tier count failed");
    return n;
}

} // extern "C"
