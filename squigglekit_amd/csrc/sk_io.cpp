// sk_io.cpp -- host-side I/O helpers of the drop-in command-line tools (no GPU involved):
//   * sk_fmt_rows        the result tables as text, with Python's own float formatting, on all cores --
//                        what `print("\t".join("{}".format(v) ...))` (MotifSeq.py:446-449, segmenter.py:222-227)
//                        produces one row at a time;
//   * sk_blow5_index /   BLOW5 records (slow5lib's binary SLOW5; the reference reads it through pyslow5,
//     sk_blow5_rows_i16  segmenter.py:321-396, dRNA_segmenter.py:85-100) straight into int16 rows.
// Both exist because the kernels handle millions of reads per second and a Python loop per read handles
// a few hundred thousand.
#include "squigglekit_hip.h"
#include <charconv>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>
#include <zlib.h>

namespace {

// repr(float) of CPython (float_repr_style "short": shortest digits that round-trip, fixed notation when
// -4 < decpt <= 16, else d.ddde[+-]XX; always a ".0" on integral fixed values)
inline void put_pyfloat(std::string &o, double v)
{
    if (v != v) { o += "nan"; return; }
    if (v == HUGE_VAL) { o += "inf"; return; }
    if (v == -HUGE_VAL) { o += "-inf"; return; }
    char b[40];
    const auto r = std::to_chars(b, b + sizeof b, v, std::chars_format::scientific);   // [-]d[.ddd]e[+-]XX, shortest
    const char *p = b, *end = r.ptr;
    if (*p == '-') { o += '-'; p++; }
    char dig[24];
    int nd = 0;
    const char *e = p;
    while (e < end && *e != 'e') { if (*e != '.') dig[nd++] = *e; e++; }
    int ex = 0;
    {
        const char *q = e + 1;
        const bool neg = (*q == '-');
        if (*q == '+' || *q == '-') q++;
        while (q < end) ex = ex * 10 + (*q++ - '0');
        if (neg) ex = -ex;
    }
    if (nd == 1 && dig[0] == '0') { o += "0.0"; return; }
    const int decpt = ex + 1;                               // value = 0.d1d2... x 10^decpt
    if (decpt > -4 && decpt <= 16) {
        if (decpt <= 0) {
            o += "0.";
            o.append((size_t)(-decpt), '0');
            o.append(dig, (size_t)nd);
        } else if (decpt >= nd) {
            o.append(dig, (size_t)nd);
            o.append((size_t)(decpt - nd), '0');
            o += ".0";
        } else {
            o.append(dig, (size_t)decpt);
            o += '.';
            o.append(dig + decpt, (size_t)(nd - decpt));
        }
    } else {
        o += dig[0];
        if (nd > 1) { o += '.'; o.append(dig + 1, (size_t)(nd - 1)); }
        o += 'e';
        int x = decpt - 1;
        o += (x < 0) ? '-' : '+';
        if (x < 0) x = -x;
        char t[8];
        int nt = 0;
        do { t[nt++] = (char)('0' + x % 10); x /= 10; } while (x);
        if (nt < 2) t[nt++] = '0';
        while (nt) o += t[--nt];
    }
}

inline void put_int(std::string &o, long long v)
{
    char b[24];
    const auto r = std::to_chars(b, b + sizeof b, v);
    o.append(b, (size_t)(r.ptr - b));
}

int clamp_threads(int nthreads, int64_t items, int64_t per_thread_min)
{
    int t = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
    if (t < 1) t = 1;
    if (t > 64) t = 64;
    const int64_t cap = items / (per_thread_min > 0 ? per_thread_min : 1);
    if (cap < t) t = cap < 1 ? 1 : (int)cap;
    return t;
}

// ---- the standard normal CDF, as scipy.special.ndtr computes it -------------------------------------------------
// MotifSeq.py:444 prints scipy.stats.norm.cdf(z) = scipy.special.ndtr(z) with all of repr()'s digits, so the
// drop-in needs the SAME double, not just an accurate one.  scipy's ndtr is the Cephes Math Library's (S. L.
// Moshier; ndtr.c, shipped in scipy.special as xsf/cephes/ndtr.h): erf by a rational T/U in x^2 for |x| <= 1, erfc by
// exp(-x^2) times a rational P/Q (|x| < 8) or R/S, Horner evaluation, no fused multiply-add (this file is built with
// -ffp-contract=off, the scipy wheels for a baseline x86-64), exp from the C library.  tests/test_fastio.py holds it
// against scipy.special.ndtr bit for bit on a few million arguments; importing scipy.special costs a command-line
// run 0.1-0.35 s, this costs nothing.
const double ND_P[] = {2.46196981473530512524E-10, 5.64189564831068821977E-1, 7.46321056442269912687E0,
                       4.86371970985681366614E1, 1.96520832956077098242E2, 5.26445194995477358631E2,
                       9.34528527171957607540E2, 1.02755188689515710272E3, 5.57535335369399327526E2};
const double ND_Q[] = {1.32281951154744992508E1, 8.67072140885989742329E1, 3.54937778887819891062E2,
                       9.75708501743205489753E2, 1.82390916687909736289E3, 2.24633760818710981792E3,
                       1.65666309194161350182E3, 5.57535340817727675546E2};
const double ND_R[] = {5.64189583547755073984E-1, 1.27536670759978104416E0, 5.01905042251180477414E0,
                       6.16021097993053585195E0, 7.40974269950448939160E0, 2.97886665372100240670E0};
const double ND_S[] = {2.26052863220117276590E0, 9.39603524938001434673E0, 1.20489539808096656605E1,
                       1.70814450747565897222E1, 9.60896809063285878198E0, 3.36907645100081516050E0};
const double ND_T[] = {9.60497373987051638749E0, 9.00260197203842689217E1, 2.23200534594684319226E3,
                       7.00332514112805075473E3, 5.55923013010394962768E4};
const double ND_U[] = {3.35617141647503099647E1, 5.21357949780152679795E2, 4.59432382970980127987E3,
                       2.26290000613890934246E4, 4.92673942608635921086E4};
const double ND_MAXLOG = 7.09782712893383996843E2;

template <int N> inline double horner(double x, const double (&c)[N])          // c[0] x^(N-1) + ... + c[N-1]
{
    double r = c[0];
    for (int i = 1; i < N; i++) r = r * x + c[i];
    return r;
}
template <int N> inline double horner1(double x, const double (&c)[N])         // x^N + c[0] x^(N-1) + ... + c[N-1]
{
    double r = x + c[0];
    for (int i = 1; i < N; i++) r = r * x + c[i];
    return r;
}

double nd_erfc(double a);
double nd_erf(double x)
{
    if (x != x) return x;
    if (x < 0.0) return -nd_erf(-x);
    if (fabs(x) > 1.0) return 1.0 - nd_erfc(x);
    const double z = x * x;
    return x * horner(z, ND_T) / horner1(z, ND_U);
}
double nd_erfc(double a)
{
    if (a != a) return a;
    const double x = a < 0.0 ? -a : a;
    if (x < 1.0) return 1.0 - nd_erf(a);
    double z = -a * a;
    if (!(z < -ND_MAXLOG)) {
        z = exp(z);
        const double p = x < 8.0 ? horner(x, ND_P) : horner(x, ND_R);
        const double q = x < 8.0 ? horner1(x, ND_Q) : horner1(x, ND_S);
        double y = (z * p) / q;
        if (a < 0) y = 2.0 - y;
        if (y != 0.0) return y;
    }
    return a < 0 ? 2.0 : 0.0;                                                  // underflow
}
inline double nd_ndtr(double a)
{
    if (a != a) return a;
    const double x = a * 0.70710678118654752440;                               // M_SQRT1_2
    const double z = fabs(x);
    if (z < 1.0) return 0.5 + 0.5 * nd_erf(x);
    const double y = 0.5 * nd_erfc(z);
    return x > 0 ? 1.0 - y : y;
}

} // namespace

extern "C" {

void sk_ndtr(const double *z, double *out, int64_t n)
{
    for (int64_t i = 0; i < n; i++) out[i] = nd_ndtr(z[i]);
}

void *sk_fmt_rows(int64_t nrows, int32_t ncols, const sk_fmt_col *cols, const uint8_t *skip, int32_t nthreads,
                  int64_t *out_len)
{
    if (out_len) *out_len = 0;
    if (nrows < 0 || ncols <= 0 || !cols || !out_len) return nullptr;
    const int T = clamp_threads(nthreads, nrows, 2048);
    std::vector<std::string> parts((size_t)T);
    auto work = [&](int t) {
        const int64_t lo = nrows * t / T, hi = nrows * (t + 1) / T;
        std::string &o = parts[(size_t)t];
        o.reserve((size_t)(hi - lo) * 96);
        for (int64_t i = lo; i < hi; i++) {
            if (skip && skip[i]) continue;
            for (int32_t c = 0; c < ncols; c++) {
                const sk_fmt_col &k = cols[c];
                if (c) o += '\t';
                switch (k.kind) {
                case SK_FMT_STR: {
                    const char *s = (const char *)k.data;
                    o.append(s + k.off[i], (size_t)(k.off[i + 1] - k.off[i]));
                    break;
                }
                case SK_FMT_STRSPAN: {
                    const char *s = (const char *)k.data;
                    o.append(s + k.off[2 * i], (size_t)(k.off[2 * i + 1] - k.off[2 * i]));
                    break;
                }
                case SK_FMT_CONST: {
                    const char *s = (const char *)k.data;
                    o.append(s + k.off[0], (size_t)(k.off[1] - k.off[0]));
                    break;
                }
                case SK_FMT_I32: put_int(o, ((const int32_t *)k.data)[i]); break;
                case SK_FMT_F64: put_pyfloat(o, ((const double *)k.data)[i]); break;
                case SK_FMT_I32LIST: {
                    const int32_t *v = (const int32_t *)k.data;
                    for (int64_t j = k.off[i]; j < k.off[i + 1]; j++) {
                        if (j > k.off[i]) o += ',';
                        put_int(o, v[j]);
                    }
                    break;
                }
                default: break;
                }
            }
            o += '\n';
        }
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    size_t total = 0;
    for (auto &p : parts) total += p.size();
    char *out = (char *)malloc(total ? total : 1);
    if (!out) return nullptr;
    size_t at = 0;
    for (auto &p : parts) { memcpy(out + at, p.data(), p.size()); at += p.size(); }
    *out_len = (int64_t)total;
    return out;
}

void sk_fmt_free(void *p) { free(p); }

// ---- BLOW5 ------------------------------------------------------------------------------------------------
// File: "BLOW5\1" magic, version (3 bytes), record compression (0 none, 1 zlib), [signal compression, >= 0.2.0],
// padding to 64 bytes, uint32 ASCII-header length, ASCII header, then records: uint64 size | payload, payload =
// uint16 idlen | id | uint32 read_group | f64 digitisation | f64 offset | f64 range | f64 sampling_rate |
// uint64 n | int16[n] | aux fields; the file ends with "5WOLB".

// What may follow the last record: nothing, or exactly the end marker.
static bool blow5_clean_end(const unsigned char *buf, int64_t len, int64_t pos)
{
    const int64_t left = len - pos;
    if (left == 0) return true;
    return left == 5 && memcmp(buf + pos, "5WOLB", 5) == 0;
}

int64_t sk_blow5_index(const void *buf_, int64_t len, int64_t first, int64_t *rec_off, int64_t *rec_size, int64_t cap)
{
    const unsigned char *buf = (const unsigned char *)buf_;
    if (!buf || len < 0 || first < 0) return SK_ERR_INVALID;
    int64_t pos = first, n = 0;
    while (pos + 8 <= len) {
        if (pos + 5 <= len && memcmp(buf + pos, "5WOLB", 5) == 0) break;
        uint64_t size;
        memcpy(&size, buf + pos, 8);
        pos += 8;
        if (size > (uint64_t)(len - pos)) return SK_ERR_INVALID;          // truncated file
        if (rec_off && n < cap) { rec_off[n] = pos; rec_size[n] = (int64_t)size; }
        n++;
        pos += (int64_t)size;
    }
    if (!blow5_clean_end(buf, len, pos)) return SK_ERR_INVALID;          // cut inside a size field / trailing junk
    return n;
}

// The same walk, at most `max_rec` records from byte `pos` on: lets a reader index the next chunk while the current
// one is decoded instead of touching every record of the file before the first sample moves.  *next_pos = where
// the following call continues; fewer than max_rec records returned = end of the file.
int64_t sk_blow5_index_some(const void *buf_, int64_t len, int64_t pos, int64_t max_rec, int64_t *rec_off,
                            int64_t *rec_size, int64_t *next_pos)
{
    const unsigned char *buf = (const unsigned char *)buf_;
    if (!buf || len < 0 || pos < 0 || max_rec < 0 || !rec_off || !rec_size || !next_pos) return SK_ERR_INVALID;
    int64_t n = 0;
    while (n < max_rec && pos + 8 <= len) {
        if (pos + 5 <= len && memcmp(buf + pos, "5WOLB", 5) == 0) break;
        uint64_t size;
        memcpy(&size, buf + pos, 8);
        pos += 8;
        if (size > (uint64_t)(len - pos)) return SK_ERR_INVALID;          // truncated file
        rec_off[n] = pos; rec_size[n] = (int64_t)size;
        n++;
        pos += (int64_t)size;
    }
    if (n < max_rec && !blow5_clean_end(buf, len, pos)) return SK_ERR_INVALID;   // (the walk stopped at the file's end)
    *next_pos = pos;
    return n;
}

int sk_blow5_rows_i16(const void *buf_, int64_t len, const int64_t *rec_off, const int64_t *rec_size, int64_t nrec,
                      int32_t comp, int64_t stride, int16_t *rows, int32_t *nsamp, char *ids, int32_t id_width, double *calib,
                      int32_t *flags, int32_t nthreads)
{
    const unsigned char *buf = (const unsigned char *)buf_;
    if (!buf || len < 0 || !rec_off || !rec_size || nrec < 0 || stride <= 0 || !rows || !nsamp || !flags || (comp != 0 && comp != 1))
        return SK_ERR_INVALID;
    const int T = clamp_threads(nthreads, nrec, 256);
    auto work = [&](int t) {
        std::vector<unsigned char> tmp;
        const int64_t lo = nrec * t / T, hi = nrec * (t + 1) / T;
        for (int64_t i = lo; i < hi; i++) {
            flags[i] = 0; nsamp[i] = 0;
            if (ids && id_width > 0) memset(ids + (size_t)i * (size_t)id_width, 0, (size_t)id_width);
            // offsets / sizes are the caller's (normally sk_blow5_index's): anything outside the buffer is unreadable
            if (rec_off[i] < 0 || rec_size[i] < 0 || rec_off[i] > len || rec_size[i] > len - rec_off[i]) { flags[i] = 2; continue; }
            const unsigned char *p = buf + rec_off[i];
            size_t sz = (size_t)rec_size[i];
            if (comp == 1) {
                // inflate into a buffer that grows from 4x the stored size; a record that still does not fit in 256x
                // (or 1 GB) is not a squiggle record
                size_t cap = sz * 4 + 4096;
                int zr = Z_BUF_ERROR;
                for (int tries = 0; tries < 7 && zr == Z_BUF_ERROR && cap <= ((size_t)1 << 30); tries++, cap *= 2) {
                    tmp.resize(cap);
                    uLongf dl = (uLongf)cap;
                    zr = uncompress(tmp.data(), &dl, p, (uLong)sz);
                    if (zr == Z_OK) sz = (size_t)dl;
                }
                if (zr != Z_OK) { flags[i] = 2; continue; }
                p = tmp.data();
            }
            if (sz < 2) { flags[i] = 2; continue; }
            uint16_t idlen;
            memcpy(&idlen, p, 2);
            const size_t fixed = 2 + (size_t)idlen + 4 + 32 + 8;
            if (sz < fixed) { flags[i] = 2; continue; }
            if (ids && id_width > 0) {
                char *d = ids + (size_t)i * (size_t)id_width;
                size_t n = idlen;
                while (n > 0 && p[2 + n - 1] == 0) n--;                       // (ids are NUL-padded in some writers)
                if (n > (size_t)id_width) { n = (size_t)id_width; flags[i] |= 4; }
                memcpy(d, p + 2, n);
                if (n < (size_t)id_width) memset(d + n, 0, (size_t)id_width - n);
            }
            const unsigned char *q = p + 2 + idlen + 4;
            if (calib) memcpy(calib + 3 * i, q, 24);                         // digitisation, offset, range
            uint64_t n;
            memcpy(&n, q + 32, 8);
            if (n > (sz - fixed) / 2) { flags[i] = 2; continue; }
            nsamp[i] = n > 0x7fffffffull ? 0x7fffffff : (int32_t)n;
            if (n > (uint64_t)stride) { flags[i] |= 1; n = (uint64_t)stride; }   // longer than a row: caller's slow path
            memcpy(rows + (size_t)i * (size_t)stride, q + 40, (size_t)n * 2);
        }
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    return SK_OK;
}

} // extern "C"
