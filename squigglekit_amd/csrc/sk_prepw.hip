// sk_prepw.hip -- filter + medmad statistics with ONE WAVEFRONT PER READ (int16 reads, MotifSeq path).
//
// Same results as the medmad instantiation of k_prep_i16 (sk_prep.hip), different shape.  The
// workgroup-per-read kernel is bound by the latency of its barrier-separated phases with 8 reads in
// flight per CU.  medmad needs no second look at the samples, so here a read belongs to one wavefront
// from the first load to the last store: no workgroup barrier, 32 reads in flight per CU, and the only
// LDS a wave owns is its value histogram (<= 8 KB).
//
//   pass 1   stream the read (16-byte loads, 8 samples per lane, packed): filter, write the survivors
//            in order for the DTW kernels, histogram (LDS atomics)
//   median   rank select on the histogram from registers (lane l owns bins [l*B, (l+1)*B)), DPP scan
//   MAD      rank select on the histogram folded around the median -- no second histogram
//
// 6.7 -> 3.9 ms per 1 M reads x 4 000 samples.  (Mean/std + mask variants of this shape were built and
// measured too, bit-exact but slower than the workgroup kernel's 7.5 ms: 13-22 ms when the std / mask
// passes re-read the samples from global memory with per-lane 2-byte loads, 10.4 ms with the compacted
// read in LDS -- 12 KB per wave leave 13 waves per CU.  zscale and the segmenter stay on k_prep_i16.)
//
// Reference: scale_outliers MotifSeq.py:317-324, medmad MotifSeq.py:192-200.
#include "sk_prepw_dev.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int WPB = 4;            // wavefronts (= reads in flight) per workgroup

// NQ: 16-byte chunks of histogram per lane (bins <= 256 NQ).
template <int NQ>
__global__ __launch_bounds__(64 * WPB, 8)
void k_prepw_medmad(const int16_t *__restrict__ sig, int64_t stride, const int32_t *__restrict__ len, int nreads,
                    int lo, int hi, int vec_ok, int16_t *__restrict__ comp, sk_prep *__restrict__ prep)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nb4 = ((hi - lo - 1) + 3) & ~3;
    const prepw_env E = prepw_setup<NQ>((unsigned *)lds_raw + (size_t)w * nb4, lane, lo, hi, vec_ok);
    const int nwaves = gridDim.x * WPB;
    for (int r = blockIdx.x * WPB + w; r < nreads; r += nwaves)
        (void)prepw_read<NQ>(E, sig, stride, len, r, lane, comp, prep);
}

typedef void (*prepw_fn)(const int16_t *, int64_t, const int32_t *, int, int, int, int, int16_t *, sk_prep *);

} // namespace

// Returns SK_OK after launching, or 1 when this kernel does not apply (caller uses k_prep_i16).
int sk_launch_prepw_medmad(sk_ctx *c, const int16_t *d_sig, int64_t stride, const int32_t *d_len,
                           int32_t nreads, int32_t lo, int32_t hi, int16_t *d_comp, sk_prep *d_prep)
{
    if (sk_tune("SK_PREP_BLOCK")) return 1;                  // tuning / A-B switch: workgroup-per-read kernel
    const int64_t nbins = (int64_t)hi - lo - 1;
    if (nbins < 1 || nbins > 2048) return 1;
    const int per_lane = (int)((nbins + 63) / 64);
    prepw_fn fn = per_lane <= 16 ? k_prepw_medmad<4> : per_lane <= 20 ? k_prepw_medmad<5> : k_prepw_medmad<8>;
    const size_t nb4 = (size_t)((nbins + 3) & ~(int64_t)3);
    const size_t lds = (size_t)WPB * nb4 * 4;
    int per_cu = (int)((160 * 1024) / (lds + 256));
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) return 1;
    const int vec_ok = ((((uintptr_t)d_sig & 15) == 0 && (stride % 8) == 0) ? 1 : 0) |
                       ((((uintptr_t)d_comp & 15) == 0 && (stride % 8) == 0) ? 2 : 0);
    int rounds = 4;
    if (const char *e = sk_tune("SK_PREP_ROUNDS")) { int v = atoi(e); if (v > 0) rounds = v; }
    const long long g = (long long)c->num_cu * per_cu * rounds;
    const long long need = ((long long)nreads + WPB - 1) / WPB;
    const int grid = (int)(g > need ? need : g);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * WPB), lds, c->stream, d_sig, stride, d_len, nreads, lo, hi,
                       vec_ok, d_comp, d_prep);
    SK_HIP(hipGetLastError());
    return SK_OK;
}
