// hbm_stream.hip -- what a plain streaming kernel gets out of this GPU's HBM, to put the segmenter's statistics
// kernel (k_seg_stats: 8 KB rows in, 1 KB of masks out per read) next to something simpler than itself.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/hbm_stream tools/ubench/hbm_stream.hip && tools/ubench/hbm_stream
// Patterns (all 16-byte loads, one wavefront-wide 1 KB line group per instruction):
//   read      grid-stride sum over the buffer, one 4-byte store per workgroup
//   rows      one wavefront per 8 KB row, all eight 1 KB loads of the row issued before the first use (the
//             statistics kernel's pattern), persistent grid
//   rows+w    the same plus one 1 KB store per row into a second buffer (its masks)
//   copy      read + write of everything (hipMemcpyDtoD's job)
// Prints GB/s = bytes moved (read + written) / HIP-event time, best of 5.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ p, size_t n16, unsigned *__restrict__ out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 q = p[i];
        acc += q.x ^ q.y ^ q.z ^ q.w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;             // (never: keeps the loads alive)
}

template <bool WRITE>
__global__ __launch_bounds__(256) void k_rows(const uint4 *__restrict__ p, size_t nrows, uint4 *__restrict__ masks, unsigned *__restrict__ out)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned acc = 0;
    for (size_t r = (size_t)blockIdx.x * 4 + w; r < nrows; r += (size_t)gridDim.x * 4) {
        const uint4 *row = p + r * 512;                        // 8 KB
        uint4 q[8];
#pragma unroll
        for (int t = 0; t < 8; t++) q[t] = row[t * 64 + lane];
        unsigned a = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) a += q[t].x ^ q[t].y ^ q[t].z ^ q[t].w;
        acc += a;
        if (WRITE) masks[r * 64 + lane] = make_uint4(a, a, a, a);
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_copy(const uint4 *__restrict__ p, uint4 *__restrict__ d, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) d[i] = p[i];
}

int main(int argc, char **argv)
{
    const size_t gb = argc > 1 ? (size_t)atoi(argv[1]) : 8;
    const size_t bytes = gb << 30, n16 = bytes / 16, nrows = bytes / 8192;
    uint4 *src, *dst, *masks;
    unsigned *out;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&dst, bytes));
    CK(hipMalloc(&masks, nrows * 1024));
    CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(src, 1, bytes));
    CK(hipMemset(dst, 0, bytes));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("# %s, %d CUs, %zu GB buffer\n", prop.gcnArchName, cus, gb);
    for (int per_cu : {4, 6, 8}) {
        for (int which = 0; which < 4; which++) {
            float best = 1e30f;
            const int grid = cus * per_cu;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0, 0));
                if (which == 0) hipLaunchKernelGGL(k_read, dim3(grid * 4), dim3(256), 0, 0, src, n16, out);
                if (which == 1) hipLaunchKernelGGL(k_rows<false>, dim3(grid), dim3(256), 0, 0, src, nrows, masks, out);
                if (which == 2) hipLaunchKernelGGL(k_rows<true>, dim3(grid), dim3(256), 0, 0, src, nrows, masks, out);
                if (which == 3) hipLaunchKernelGGL(k_copy, dim3(grid * 4), dim3(256), 0, 0, src, dst, n16);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double moved = which == 3 ? 2.0 * bytes : which == 2 ? bytes + nrows * 1024.0 : (double)bytes;
            const char *nm[] = {"read", "rows", "rows+w", "copy"};
            printf("%-7s %d workgroups of 4 waves per CU: %7.3f ms  %7.1f GB/s\n", nm[which], per_cu * (which == 0 || which == 3 ? 4 : 1),
                   best, moved / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
