// sk_synth.hip -- deterministic synthetic squiggle generator on the device.
//
// Bench tooling, not a reference function (the reference ships no generator).  Same squiggle
// model as squigglekit_amd/synth.py: event levels ~ N(500, 80) with
// dwell 1 + Poisson(8), N(0, 8) noise, a stall plateau near the start, with p = 0.5 a second
// plateau, with p = 0.5 an implanted copy of the motif, and four spike samples from
// {-5, 0, 950, 1100}.  Counter-based RNG (splitmix64 of seed/read/counter), one lane per read,
// eight samples per 16-byte store.  The numbers differ from the numpy generator's; parity
// tests never depend on them -- they read the generated batch back.
#include "sk_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct Rng {
    uint64_t key, ctr;
    __device__ uint64_t next() { return mix64(key + (ctr++) * 0xD1342543DE82EF95ull); }
    __device__ float uni() { return ((float)(next() >> 40) + 0.5f) * (1.0f / 16777216.0f); }   // (0,1)
    __device__ int below(int n) { return (int)((next() >> 33) % (uint64_t)n); }
    __device__ float gauss()
    {
        const float u1 = uni(), u2 = uni();
        return sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2);
    }
    __device__ int poisson8()
    {
        const float limit = 3.3546263e-4f;      // exp(-8)
        int k = 0;
        float p = uni();
        while (p > limit && k < 64) { k++; p *= uni(); }
        return k;
    }
};

// Real-signal mode (bench sensitivity runs): read r = a window of M samples of one measured squiggle (tmpl, T
// samples) at a random offset, plus rounded N(0, sigma) noise -- no plateaus, implants or spikes.
__global__ __launch_bounds__(64)
void k_synth_windows(int16_t *__restrict__ sig, int64_t stride, int nreads, int M, uint64_t seed, int64_t row0,
                     const int16_t *__restrict__ tmpl, int T, float sigma)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nreads) return;
    Rng g;
    g.key = mix64(seed ^ mix64((uint64_t)(row0 + r) + 0x7654321ull));
    g.ctr = 0;
    const int off = (T > M) ? g.below(T - M + 1) : 0;
    int16_t *row = sig + (int64_t)r * stride;
    for (int i = 0; i < M; i++) {
        const int t = tmpl[(off + i) % T];
        int q = t + (int)rintf(sigma * g.gauss());
        row[i] = (int16_t)min(32767, max(-32768, q));
    }
}

// hit_pm / stretch_pm: per-mille of reads that carry the motif as it is / repeated `stretch` times per point
// (a time-stretched copy: its optimal path is `stretch` times wider than the motif, so the windowed DTW pass
// cannot certify the read and it takes the exact retry).  row0: index of the first generated row in the stream
// (row r of this launch is row row0 + r of the seed's batch, whatever the launch covers).
__global__ __launch_bounds__(64)
void k_synth(int16_t *__restrict__ sig, int64_t stride, int nreads, int M, uint64_t seed,
             const int16_t *__restrict__ motif, int N, int64_t row0, int hit_pm, int stretch_pm, int stretch)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nreads) return;
    Rng g;
    g.key = mix64(seed ^ mix64((uint64_t)(row0 + r) + 0x1234567ull));
    g.ctr = 0;
    const int scale = max(1, M / 4000);
    const int s0 = g.below(60), l0 = 100 + g.below(500);
    const bool has2 = g.uni() < 0.5f;
    const int s1 = 1200 * scale + g.below(2200 * scale), l1 = 160 + g.below(340);
    const bool hit = (motif != nullptr) && (N > 0) && (N < M) && (g.uni() < 0.001f * (float)hit_pm);
    const int moff = (N < M) ? g.below(M - N) : 0;
    // (drawn from a second stream, so that the default batch -- stretch_pm == 0 -- is unchanged by the option)
    bool wide = false;
    int woff = 0;
    if (stretch_pm > 0 && motif != nullptr && N * stretch < M) {
        Rng g2;
        g2.key = mix64(g.key ^ 0x5bd1e995ull);
        g2.ctr = 0;
        wide = g2.uni() < 0.001f * (float)stretch_pm;
        woff = g2.below(M - N * stretch);
    }
    int spos[4], sval[4];
    const int spikes[4] = {-5, 0, 950, 1100};
    for (int k = 0; k < 4; k++) { spos[k] = g.below(M); sval[k] = spikes[g.below(4)]; }

    int16_t *row = sig + (int64_t)r * stride;
    float level = 500.0f + 80.0f * g.gauss();
    int left = 1 + g.poisson8();
    const bool vec = ((((uintptr_t)row) & 15) == 0);
    for (int base = 0; base < M; base += 8) {
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = base + k;
            if (left == 0) { level = 500.0f + 80.0f * g.gauss(); left = 1 + g.poisson8(); }
            left--;
            float x = level + 8.0f * g.gauss();
            if (i >= s0 && i < s0 + l0) x = 505.0f + 12.0f * g.gauss();
            if (has2 && i >= s1 && i < s1 + l1) x = 495.0f + 10.0f * g.gauss();
            int q = (int)rintf(x);
            q = min(32767, max(-32768, q));
            if (hit && i >= moff && i < moff + N) q = motif[i - moff];
            if (wide && i >= woff && i < woff + N * stretch) q = motif[(i - woff) / stretch];
#pragma unroll
            for (int s = 0; s < 4; s++) if (i == spos[s]) q = sval[s];
            v[k] = q;
        }
        if (vec && base + 8 <= M) {
            int4 o;
            o.x = (v[0] & 0xffff) | (v[1] << 16);
            o.y = (v[2] & 0xffff) | (v[3] << 16);
            o.z = (v[4] & 0xffff) | (v[5] << 16);
            o.w = (v[6] & 0xffff) | (v[7] << 16);
            *(int4 *)(row + base) = o;
        } else {
            for (int k = 0; k < 8 && base + k < M; k++) row[base + k] = (int16_t)v[k];
        }
    }
}

// int16 raw rows -> the float64 pA values SquigglePull writes (SquigglePull.py:183-189,238-240):
// np.round((raw + offset) * (range / digitisation), 2) = rint(v * 100) / 100 -- three correctly rounded operations --
// as one contiguous ragged batch: read r at out[r * nsamples ..), off[r] = r * nsamples.
__global__ __launch_bounds__(256)
void k_raw_to_pa(const int16_t *__restrict__ sig, int64_t stride, int nreads, int nsamples, double offset,
                 double raw_unit, double *__restrict__ out, int64_t *__restrict__ off)
{
    const int64_t total = (int64_t)nreads * nsamples;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / nsamples;
        const int k = (int)(i - r * nsamples);
        const double v = ((double)sig[r * stride + k] + offset) * raw_unit;
        out[i] = rint(v * 100.0) / 100.0;
    }
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= nreads; r += (int64_t)gridDim.x * blockDim.x)
        off[r] = r * nsamples;
}

// the same conversion for a batch of raw rows with their own channel constants (what segmenter.py:345-349 / :366-370 do
// read by read for fast5 / slow5 input): read r -> out[off[r] .. off[r+1])
__global__ __launch_bounds__(256)
void k_rows_to_pa(const int16_t *__restrict__ sig, int64_t stride, int nreads, const int64_t *__restrict__ off,
                  const double *__restrict__ cal, double *__restrict__ out)
{
    for (int r = blockIdx.x; r < nreads; r += gridDim.x) {
        const int64_t o0 = off[r];
        const int n = (int)(off[r + 1] - o0);
        const double offset = cal[2 * r], unit = cal[2 * r + 1];
        const int16_t *row = sig + (int64_t)r * stride;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const double v = ((double)row[i] + offset) * unit;
            out[o0 + i] = rint(v * 100.0) / 100.0;
        }
    }
}

// int32 centi-units (what sk_tsv_parse_centi makes of "ddd.dd" tokens) -> float64: c / 100.0, one correctly rounded
// division of two exact operands = float("ddd.dd") bit for bit (segmenter.py:198-199, MotifSeq.py:270)
__global__ __launch_bounds__(256)
void k_centi_to_f64(const int32_t *__restrict__ c, int64_t total, double *__restrict__ out)
{
    const int64_t n4 = total >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int4 q = ((const int4 *)c)[i];
        double2 a, b;
        a.x = (double)q.x / 100.0; a.y = (double)q.y / 100.0; b.x = (double)q.z / 100.0; b.y = (double)q.w / 100.0;
        ((double2 *)out)[2 * i] = a; ((double2 *)out)[2 * i + 1] = b;
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (double)c[i] / 100.0;
}

} // namespace

int sk_launch_centi_to_f64(sk_ctx *c, const int32_t *d_centi, int64_t total, double *d_out)
{
    if (total <= 0) return SK_OK;
    const int64_t need = (total / 4 + 255) / 256 + 1;
    const int grid = (int)(need < (int64_t)c->num_cu * 16 ? need : (int64_t)c->num_cu * 16);
    hipLaunchKernelGGL(k_centi_to_f64, dim3(grid), dim3(256), 0, c->stream, d_centi, total, d_out);
    SK_HIP(hipGetLastError());
    return SK_OK;
}

int sk_launch_rows_to_pa(sk_ctx *c, const int16_t *d_sig, int64_t stride, int32_t nreads, const int64_t *d_off,
                         const double *d_cal, double *d_out)
{
    if (nreads <= 0) return SK_OK;
    const int grid = nreads < c->num_cu * 16 ? nreads : c->num_cu * 16;
    hipLaunchKernelGGL(k_rows_to_pa, dim3(grid), dim3(256), 0, c->stream, d_sig, stride, nreads, d_off, d_cal, d_out);
    SK_HIP(hipGetLastError());
    return SK_OK;
}

int sk_launch_raw_to_pa(sk_ctx *c, const int16_t *d_sig, int64_t stride, int32_t nreads, int32_t nsamples,
                        double offset, double raw_unit, double *d_out, int64_t *d_off)
{
    if (nreads <= 0 || nsamples <= 0) return SK_OK;
    hipLaunchKernelGGL(k_raw_to_pa, dim3(c->num_cu * 8), dim3(256), 0, c->stream, d_sig, stride, nreads, nsamples,
                       offset, raw_unit, d_out, d_off);
    SK_HIP(hipGetLastError());
    return SK_OK;
}

int sk_launch_synth(sk_ctx *c, int16_t *d_sig, int64_t stride, int32_t nreads, int32_t nsamples,
                    uint64_t seed, const int16_t *d_motif_i16, int32_t nmotif, int64_t row0,
                    int hit_pm, int stretch_pm, int stretch)
{
    if (nreads <= 0 || nsamples <= 0) return SK_OK;
    const int grid = (nreads + 63) / 64;
    hipLaunchKernelGGL(k_synth, dim3(grid), dim3(64), 0, c->stream, d_sig, stride, nreads, nsamples, seed,
                       d_motif_i16, nmotif, row0, hit_pm, stretch_pm, stretch < 1 ? 1 : stretch);
    SK_HIP(hipGetLastError());
    return SK_OK;
}

int sk_launch_synth_windows(sk_ctx *c, int16_t *d_sig, int64_t stride, int32_t nreads, int32_t nsamples,
                            uint64_t seed, int64_t row0, const int16_t *d_tmpl, int32_t ntmpl, float sigma)
{
    if (nreads <= 0 || nsamples <= 0) return SK_OK;
    const int grid = (nreads + 63) / 64;
    hipLaunchKernelGGL(k_synth_windows, dim3(grid), dim3(64), 0, c->stream, d_sig, stride, nreads, nsamples, seed,
                       row0, d_tmpl, ntmpl, sigma);
    SK_HIP(hipGetLastError());
    return SK_OK;
}
