// sk_tsv.cpp -- multi-threaded tokenizer for SquigglePull TSV text (host code, no GPU).
//
// The reference parses each line with str.split + int()/float() per token
// (/root/reference/segmenter.py:192-201, MotifSeq.py:265-270): ~0.4 ms per 4 000-sample read,
// two orders of magnitude slower than the kernels.  This is the same conversion done natively:
// lines are indexed with memchr, tokens are converted on worker threads, the numbers land in
// one flat float64 array with per-line offsets.  Numeric semantics are Python's: decimal
// literals are converted exactly like float()/strtod (Clinger fast path: a mantissa below 2^53
// divided or multiplied by an exactly representable power of ten is one correctly rounded
// operation; everything else goes through strtod).  A line holding any token outside the plain
// grammar [+-]digits[.digits][e[+-]digits] is flagged SK_TSV_SLOW and left to the caller's
// Python path, so odd inputs keep the reference's behaviour (including its exceptions).
#include "squigglekit_hip.h"
#include <errno.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <pthread.h>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// A pool of worker threads that lives as long as the library (round 6).  Every chunk of a TSV goes through three calls
// (line index, token counts, parse), each of which used to create and join its 16-32 threads: ~100 thread creations per
// 48 MB chunk, about 2 of the 4.5 ms a chunk took on the 256-thread bench host.  run(n, f) executes f(0) .. f(n - 1),
// f(0) on the caller; the workers sleep on a condition variable between jobs.  One job at a time: a second host thread
// that arrives while the pool is busy runs its job on freshly created threads, as before.
// (a forked child inherits the pool's bookkeeping but none of its threads: it creates threads per call, as before round 6)
static bool g_forked_child = false;
class Pool {
public:
    Pool() { pthread_atfork(nullptr, nullptr, [] { g_forked_child = true; }); }
    void run(int n, const std::function<void(int)> &f)
    {
        if (n <= 1) { f(0); return; }
        std::unique_lock<std::mutex> own(busy_, std::defer_lock);
        if (!g_forked_child) own.try_lock();
        if (!own.owns_lock()) {                              // pool in use by another host thread (or not usable: forked child)
            std::vector<std::thread> th;
            for (int t = 1; t < n; t++) th.emplace_back(f, t);
            f(0);
            for (auto &x : th) x.join();
            return;
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            while ((int)workers_.size() < n - 1) {
                const int id = (int)workers_.size() + 1;
                workers_.emplace_back([this, id] { loop(id); });
            }
            job_ = &f; njob_ = n; pending_ = n - 1; gen_++;
        }
        cv_.notify_all();
        f(0);
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
    }
    // (never destroyed: the object lives on the heap for the life of the process, its sleeping workers end with it --
    // no join at exit, nothing to unwind in a forked child)
private:
    void loop(int id)
    {
        unsigned long seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        if (id < njob_ && job_) seen = gen_ - 1;             // created for the job being posted right now
        else seen = gen_;
        while (true) {
            cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            if (id >= njob_) continue;                       // this job uses fewer workers
            const std::function<void(int)> *f = job_;
            lk.unlock();
            (*f)(id);
            lk.lock();
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::mutex busy_, mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> workers_;
    const std::function<void(int)> *job_ = nullptr;
    int njob_ = 0, pending_ = 0;
    unsigned long gen_ = 0;
    bool stop_ = false;
};
Pool &g_pool = *new Pool;

const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14,
                        1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// Convert token [p, e) to double.  Returns 0 ok, 1 = not plain grammar (caller falls back).
// *isint: token is [+-]digits only.
inline int conv(const char *p, const char *e, double *out, int *isint)
{
    const char *s = p;
    if (s == e) return 1;
    bool neg = false;
    if (*s == '+' || *s == '-') { neg = (*s == '-'); s++; }
    if (s == e) return 1;
    unsigned long long mant = 0;
    int nd = 0, ndec = 0, dropped = 0;
    bool any = false, dot = false, exact = true;
    for (; s < e; s++) {
        const char ch = *s;
        if (ch >= '0' && ch <= '9') {
            any = true;
            if (nd < 19) {                          // 19 significant digits always fit 64 bits
                mant = mant * 10 + (unsigned)(ch - '0');
                if (mant != 0) nd++;                // leading zeros are not significant
                if (dot) ndec++;
            } else {
                exact = false;                      // too long for the fast path: strtod below
                if (!dot) dropped++;
            }
        } else if (ch == '.' && !dot) {
            dot = true;
        } else {
            break;
        }
    }
    if (!any) return 1;
    int ex = 0;
    bool hasexp = false;
    if (s < e && (*s == 'e' || *s == 'E')) {
        hasexp = true;
        s++;
        bool eneg = false;
        if (s < e && (*s == '+' || *s == '-')) { eneg = (*s == '-'); s++; }
        if (s == e) return 1;
        int ev = 0;
        for (; s < e; s++) {
            if (*s < '0' || *s > '9') return 1;
            if (ev < 100000) ev = ev * 10 + (*s - '0');
        }
        ex = eneg ? -ev : ev;
    }
    if (s != e) return 1;
    *isint = (!dot && !hasexp) ? 1 : 0;
    const int e10 = ex - ndec + dropped;
    if (exact && mant < (1ull << 53) && e10 >= -22 && e10 <= 22) {
        double v = (double)mant;
        v = (e10 < 0) ? v / P10[-e10] : v * P10[e10];
        *out = neg ? -v : v;
        return 0;
    }
    // rare: long mantissa or big exponent -> the C library's correctly rounded conversion
    char tmp[64];
    const size_t len = (size_t)(e - p);
    if (len >= sizeof tmp) return 1;
    memcpy(tmp, p, len);
    tmp[len] = 0;
    char *endp = nullptr;
    const double v = strtod(tmp, &endp);
    if (endp != tmp + len) return 1;
    *out = v;
    return 0;
}

struct LineInfo { const char *b, *e; };

// Line starts of one buffer, found on several threads (every thread scans its slice for newlines) and kept per host
// thread between sk_tsv_count_lines / _count_tokens / _parse*, which a caller runs one after the other on the same
// chunk: three serial memchr passes over the chunk had become most of the tokenizer's time.
struct LineIndex {
    const char *buf = nullptr;
    size_t len = 0;
    uint64_t print = 0;                // fingerprint of the bytes (an allocator may hand the same address out again)
    std::vector<int64_t> off;          // off[i] = start of line i; off[n] = len
};
thread_local LineIndex g_lines;

uint64_t fingerprint(const char *buf, size_t len)
{
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)len;
    for (int k = 0; k < 16 && len >= 8; k++) {
        uint64_t w;
        memcpy(&w, buf + (len - 8) * (size_t)k / 15, 8);
        h = (h ^ w) * 0xBF58476D1CE4E5B9ull;
        h ^= h >> 29;
    }
    return h;
}

// fresh: rebuild even if the cached index looks like this buffer's (sk_tsv_count_lines, the first call on a chunk)
const LineIndex &line_index(const char *buf, size_t len, bool fresh = false)
{
    LineIndex &L = g_lines;
    const uint64_t fp = fingerprint(buf, len);
    if (!fresh && L.buf == buf && L.len == len && L.print == fp && !L.off.empty()) {
        // same address, length and sampled words: still make sure every cached line start sits behind a newline of
        // THIS content (a caller that reuses a buffer and skips sk_tsv_count_lines must not get stale offsets)
        bool ok = true;
        for (size_t i = 0; ok && i + 1 < L.off.size(); i++)
            ok = L.off[i] == 0 || (L.off[i] > 0 && (size_t)L.off[i] <= len && buf[L.off[i] - 1] == '\n');
        if (ok) return L;
    }
    L.buf = buf; L.len = len; L.print = fp; L.off.clear();
    int T = (int)std::thread::hardware_concurrency();
    if (T > 16) T = 16;
    if ((size_t)T > len / (1u << 20) + 1) T = (int)(len / (1u << 20)) + 1;
    if (T < 1) T = 1;
    std::vector<std::vector<int64_t>> part((size_t)T);
    auto scan = [&](int t) {
        const size_t a = len * (size_t)t / (size_t)T, b = len * (size_t)(t + 1) / (size_t)T;
        std::vector<int64_t> &v = part[(size_t)t];
        const char *p = buf + a, *end = buf + b;
        while (p < end) {
            const char *q = (const char *)memchr(p, '\n', (size_t)(end - p));
            if (!q) break;
            v.push_back((int64_t)(q - buf) + 1);         // a line starts behind every newline
            p = q + 1;
        }
    };
    g_pool.run(T, scan);
    size_t total = 1;
    for (auto &v : part) total += v.size();
    L.off.reserve(total + 1);
    if (len > 0) L.off.push_back(0);
    for (auto &v : part) L.off.insert(L.off.end(), v.begin(), v.end());
    if (!L.off.empty() && L.off.back() == (int64_t)len) L.off.pop_back();   // the file's last newline starts no line
    L.off.push_back((int64_t)len);                                          // sentinel: off[n] = len
    return L;
}

} // namespace

extern "C" {

// Count lines ('\n'-terminated; a last line without newline counts) in buf[0..len).
int64_t sk_tsv_count_lines(const char *buf, size_t len)
{
    if (!buf) return -1;
    return (int64_t)line_index(buf, len, true).off.size() - 1;
}

// Count the tokens from column start_col on, per line, so the caller can size `values`.
// ntok[i] = number of data tokens of line i (0 if the line has fewer columns).
int sk_tsv_count_tokens(const char *buf, size_t len, int32_t start_col, int64_t nlines, int64_t *ntok,
                        int32_t nthreads)
{
    if (!buf || !ntok || start_col < 0 || nlines < 0) return SK_ERR_INVALID;
    const LineIndex &LI = line_index(buf, len);
    if ((int64_t)LI.off.size() - 1 != nlines) return SK_ERR_INVALID;
    const int64_t *lo = LI.off.data();
    if (nthreads < 1) nthreads = 1;
    auto work = [&](int t) {
        for (int64_t i = t; i < nlines; i += nthreads) {
            const char *p = buf + lo[i], *e = buf + lo[i + 1];
            if (e > p && e[-1] == '\n') e--;
            int64_t tabs = 0;
            for (const char *s = p; s < e; s++) tabs += (*s == '\t');
            const int64_t cols = tabs + 1;
            ntok[i] = cols > start_col ? cols - start_col : 0;
        }
    };
    g_pool.run(nthreads, work);
    return SK_OK;
}

// Parse.  off[i] .. off[i+1] (prefix sums of ntok, supplied by the caller) is line i's slice of
// `values`.  Per line: name_off/name_len = column 0, id_off/id_len = column 1 (byte ranges in
// buf), flags = SK_TSV_* bits.
int sk_tsv_parse(const char *buf, size_t len, int32_t start_col, int64_t nlines, const int64_t *off,
                 double *values, int64_t *name_off, int32_t *name_len, int64_t *id_off, int32_t *id_len,
                 int32_t *flags, int32_t nthreads)
{
    if (!buf || !off || !values || !flags || start_col < 0 || nlines < 0) return SK_ERR_INVALID;
    std::vector<LineInfo> lines((size_t)nlines);
    {
        const char *p = buf, *end = buf + len;
        for (int64_t i = 0; i < nlines; i++) {
            const char *q = (const char *)memchr(p, '\n', (size_t)(end - p));
            lines[(size_t)i].b = p;
            lines[(size_t)i].e = q ? q : end;
            p = q ? q + 1 : end;
        }
    }
    if (nthreads < 1) nthreads = 1;
    auto work = [&](int t) {
        for (int64_t i = t; i < nlines; i += nthreads) {
            const char *p = lines[(size_t)i].b, *e = lines[(size_t)i].e;
            int32_t fl = 0;
            int col = 0;
            int64_t k = off[i];
            const int64_t kend = off[i + 1];
            bool allint = true, anynz = false, firstdot = false;
            const char *s = p;
            if (name_off) { name_off[i] = 0; name_len[i] = 0; }
            if (id_off) { id_off[i] = 0; id_len[i] = 0; }
            while (true) {
                const char *q = (const char *)memchr(s, '\t', (size_t)(e - s));
                const char *te = q ? q : e;
                if (col == 0 && name_off) { name_off[i] = s - buf; name_len[i] = (int32_t)(te - s); }
                if (col == 1 && id_off) { id_off[i] = s - buf; id_len[i] = (int32_t)(te - s); }
                if (col >= start_col && k < kend) {
                    double v = 0.0;
                    int isint = 0;
                    if (conv(s, te, &v, &isint)) { fl |= SK_TSV_SLOW; v = 0.0; isint = 0; }
                    if (col == start_col) firstdot = memchr(s, '.', (size_t)(te - s)) != nullptr;
                    allint = allint && isint;
                    anynz = anynz || (v != 0.0);
                    values[k++] = v;
                }
                col++;
                if (!q) break;
                s = q + 1;
            }
            if (allint) fl |= SK_TSV_ALLINT;
            if (anynz) fl |= SK_TSV_ANY;
            if (firstdot) fl |= SK_TSV_FIRSTDOT;
            if (col <= start_col) fl |= SK_TSV_SHORT;
            flags[i] = fl;
        }
    };
    g_pool.run(nthreads, work);
    return SK_OK;
}

// Decimal lines as CENTI-UNITS (round 6).  What SquigglePull writes by default is np.round(pA, 2) (SquigglePull.py:
// 183-189, 222): tokens with at most two decimals, "96.5", "103.25", "88.0".  float("ddd.dd") is the double nearest to
// c / 100 for the integer c = ddddd -- and so is the IEEE quotient (double)c / 100.0 (both operands exact, one correctly
// rounded operation: the same fast path conv() above takes).  So such a line is carried as int32 centi-units: half the
// bytes on the way to the GPU, no division and no second pass on the host; the device makes value = c / 100.0, bit for
// bit float()'s.  Same layout and flags as sk_tsv_parse (values[off[i] .. off[i+1]) = line i's tokens) plus
// SK_TSV_CENTI: every data token of the line is [+-]digits[.digits] with at most two decimals (further decimals must be
// zeros), |c| < 2^31, and is not a negative zero (float("-0.0") keeps its sign; c cannot).  A line without the flag
// holds unspecified values: the caller sends the chunk through sk_tsv_parse instead.
int sk_tsv_parse_centi(const char *buf, size_t len, int32_t start_col, int64_t nlines, const int64_t *off,
                       int32_t *values, int64_t *name_off, int32_t *name_len, int64_t *id_off, int32_t *id_len,
                       int32_t *flags, int32_t nthreads)
{
    if (!buf || !off || !values || !flags || start_col < 0 || nlines < 0) return SK_ERR_INVALID;
    const LineIndex &LI = line_index(buf, len);
    if ((int64_t)LI.off.size() - 1 != nlines) return SK_ERR_INVALID;
    const int64_t *lo = LI.off.data();
    if (nthreads < 1) nthreads = 1;
    auto work = [&](int t) {
        const int64_t per = (nlines + nthreads - 1) / nthreads;         // contiguous lines: contiguous output
        const int64_t i0 = t * per, i1 = (i0 + per < nlines) ? i0 + per : nlines;
        for (int64_t i = i0; i < i1; i++) {
            const char *p = buf + lo[i], *e = buf + lo[i + 1];
            if (e > p && e[-1] == '\n') e--;
            if (name_off) { name_off[i] = 0; name_len[i] = 0; }
            if (id_off) { id_off[i] = 0; id_len[i] = 0; }
            int col = 0;
            const char *s = p;
            while (col < start_col) {                                   // the leading columns: name, read id, ...
                const char *q = (const char *)memchr(s, '\t', (size_t)(e - s));
                const char *te = q ? q : e;
                if (col == 0 && name_off) { name_off[i] = s - buf; name_len[i] = (int32_t)(te - s); }
                if (col == 1 && id_off) { id_off[i] = s - buf; id_len[i] = (int32_t)(te - s); }
                col++;
                if (!q) { s = e + 1; break; }
                s = q + 1;
            }
            int32_t fl = 0;
            if (s > e) { flags[i] = SK_TSV_SHORT; continue; }          // fewer than start_col + 1 columns
            int64_t k = off[i];
            const int64_t kend = off[i + 1];
            bool centi = true, anynz = false, anydot = false, firstdot = false, first = true;
            while (true) {                                              // s: start of a data token
                bool neg = false;
                if (s < e && (*s == '-' || *s == '+')) { neg = (*s == '-'); s++; }
                uint32_t ip = 0, fr = 0;
                int nd = 0, nf = 0;
                while (s < e && (unsigned)(*s - '0') <= 9u) { ip = ip * 10u + (unsigned)(*s - '0'); nd++; s++; }
                bool dot = false;
                if (s < e && *s == '.') {
                    dot = true;
                    s++;
                    while (s < e && (unsigned)(*s - '0') <= 9u) {
                        if (nf < 2) fr = fr * 10u + (unsigned)(*s - '0');
                        else if (*s != '0') centi = false;              // a third significant decimal
                        nf++; s++;
                    }
                }
                if ((nd == 0 && nf == 0) || nd > 7 || (s < e && *s != '\t')) {
                    centi = false;                                      // not [+-]digits[.digits] (or too long): the other parser
                    while (s < e && *s != '\t') s++;
                }
                const uint32_t c = ip * 100u + (nf == 0 ? 0u : nf == 1 ? fr * 10u : fr);
                if (neg && c == 0u) centi = false;                      // -0.0
                if (k < kend) values[k++] = neg ? -(int32_t)c : (int32_t)c;
                anynz = anynz || c != 0u;
                anydot = anydot || dot;
                if (first) { firstdot = dot; first = false; }
                col++;
                if (s >= e) break;
                s++;                                                    // the tab
            }
            if (k != kend) centi = false;                               // (the caller's offsets are another line's)
            if (!anydot) fl |= SK_TSV_ALLINT;
            if (anynz) fl |= SK_TSV_ANY;
            if (firstdot) fl |= SK_TSV_FIRSTDOT;
            if (centi) fl |= SK_TSV_CENTI;
            flags[i] = fl;
        }
    };
    g_pool.run(nthreads, work);
    return SK_OK;
}

// Lines of integer samples straight into int16 rows: the common case (SquigglePull's raw TSV), without the float64
// detour and without a Python object per read.  rows[i * stride ..] receives line i's data tokens (columns
// start_col ..), nsamp[i] their number.  flags[i]: SK_TSV_ALLINT only if EVERY data token is [+-]digits, fits int16
// and the line has at most `stride` of them -- only then is the row valid; any other line gets SK_TSV_SLOW (or
// SK_TSV_SHORT) and is left to the caller's general path.  SK_TSV_ANY: some value is non-zero.  line_off[i] =
// byte offset of line i in buf (line_off[nlines] = end), so the caller can slice the odd lines out.
int sk_tsv_parse_i16(const char *buf, size_t len, int32_t start_col, int64_t nlines, int64_t stride, int16_t *rows,
                     int32_t *nsamp, int64_t *name_off, int32_t *name_len, int64_t *id_off, int32_t *id_len,
                     int32_t *flags, int64_t *line_off, int32_t nthreads)
{
    if (!buf || !rows || !nsamp || !flags || !line_off || start_col < 0 || nlines < 0 || stride <= 0)
        return SK_ERR_INVALID;
    {
        const LineIndex &LI = line_index(buf, len);
        if ((int64_t)LI.off.size() - 1 != nlines) return SK_ERR_INVALID;
        memcpy(line_off, LI.off.data(), (size_t)(nlines + 1) * sizeof(int64_t));
    }
    if (nthreads < 1) nthreads = 1;
    auto work = [&](int t) {
        // contiguous blocks of lines per thread: neighbouring rows are written by one thread
        const int64_t per = (nlines + nthreads - 1) / nthreads;
        const int64_t i0 = t * per, i1 = (i0 + per < nlines) ? i0 + per : nlines;
        for (int64_t i = i0; i < i1; i++) {
            const char *p = buf + line_off[i], *e = buf + line_off[i + 1];
            if (e > p && e[-1] == '\n') e--;
            int16_t *row = rows + i * stride;
            int col = 0;
            int64_t k = 0;
            bool ok = true, anynz = false;
            if (name_off) { name_off[i] = 0; name_len[i] = 0; }
            if (id_off) { id_off[i] = 0; id_len[i] = 0; }
            const char *s = p;
            while (true) {
                const char *te = s;
                if (col >= start_col && e - s >= 8) {
                    // the common token -- one to five digits and a tab -- eight bytes at a time: where the first
                    // non-digit sits, then the digits as one multiply-and-shift cascade (what follows handles
                    // everything else, and the last tokens of a line)
                    uint64_t w;
                    memcpy(&w, s, 8);
                    const uint64_t d = w ^ 0x3030303030303030ull;                 // digits -> 0 .. 9
                    const uint64_t nd = (((d & 0x7f7f7f7f7f7f7f7full) + 0x7676767676767676ull) | d) & 0x8080808080808080ull;
                    const int n = nd ? (__builtin_ctzll(nd) >> 3) : 8;            // bytes before the first non-digit
                    if (n >= 1 && n <= 5 && ((w >> (8 * n)) & 0xff) == '\t') {
                        uint64_t v8 = d << (8 * (8 - n));                          // right-aligned, leading zeros
                        v8 = (v8 * 2561) >> 8;
                        v8 = ((v8 & 0x00FF00FF00FF00FFull) * 6553601) >> 16;
                        v8 = ((v8 & 0x0000FFFF0000FFFFull) * 42949672960001ull) >> 32;
                        const int v = (int)v8;
                        if (v > 32767 || k >= stride) ok = false;
                        else { row[k] = (int16_t)v; anynz = anynz || v != 0; }
                        k++;
                        col++;
                        s += n + 1;                                                // (s + n < e: not the line's end)
                        continue;
                    }
                }
                if (col >= start_col) {                     // integer token, converted while looking for its end
                    bool neg = false;
                    if (te < e && (*te == '-' || *te == '+')) { neg = (*te == '-'); te++; }
                    const char *d0 = te;
                    int v = 0;
                    while (te < e && (unsigned)(*te - '0') <= 9u) { if (v < 100000) v = v * 10 + (*te - '0'); te++; }
                    const bool plain = te > d0 && (te == e || *te == '\t');
                    if (!plain) { ok = false; while (te < e && *te != '\t') te++; }
                    else {
                        if (neg) v = -v;
                        if (v < -32768 || v > 32767 || k >= stride) ok = false;
                        else { row[k] = (int16_t)v; anynz = anynz || v != 0; }
                        k++;
                    }
                } else {
                    const char *q = (const char *)memchr(s, '\t', (size_t)(e - s));
                    te = q ? q : e;
                    if (col == 0 && name_off) { name_off[i] = s - buf; name_len[i] = (int32_t)(te - s); }
                    if (col == 1 && id_off) { id_off[i] = s - buf; id_len[i] = (int32_t)(te - s); }
                }
                col++;
                if (te >= e) break;
                s = te + 1;
            }
            int32_t fl = 0;
            if (col <= start_col) fl |= SK_TSV_SHORT;
            else if (ok) fl |= SK_TSV_ALLINT;
            else fl |= SK_TSV_SLOW;
            if (anynz) fl |= SK_TSV_ANY;
            nsamp[i] = (int32_t)(k < 0x7fffffff ? k : 0x7fffffff);
            flags[i] = fl;
        }
    };
    g_pool.run(nthreads, work);
    return SK_OK;
}

} // extern "C"
