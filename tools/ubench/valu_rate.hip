// valu_rate.hip -- issue cost (cycles per wave-instruction per SIMD) of the VALU ops the DTW
// kernels are built from, measured with 8 independent chains per wave and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP8(S) S S S S S S S S
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed, int iters)
{
    unsigned a[8], b = seed + threadIdx.x, c = seed * 3 + 1;
    for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x;
    double d[8];
    for (int i = 0; i < 8; i++) d[i] = (double)(seed + i + threadIdx.x);
    double db = (double)b;
    for (int it = 0; it < iters; it++) {
#define ONE(i)                                                                                         \
        if (OP == 0) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));          \
        if (OP == 1) asm volatile("v_sad_u32 %0, %0, %1, %2 clamp" : "+v"(a[i]) : "v"(b), "v"(c));     \
        if (OP == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                       \
        if (OP == 3) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                       \
        if (OP == 4) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));          \
        if (OP == 5) asm volatile("v_add_f32 %0, |%0|, %1" : "+v"(a[i]) : "v"(b));                     \
        if (OP == 6) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));                      \
        if (OP == 7) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));                      \
        if (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));              \
        if (OP == 9) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(db) : "vcc");          \
        if (OP == 10) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b)); \
        if (OP == 11) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));         \
        if (OP == 12) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                      \
        if (OP == 13) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                      \
        if (OP == 14) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(db));                  \
        if (OP == 15) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        for (int u = 0; u < 8; u++) { ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5) ONE(6) ONE(7) }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + (unsigned)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*fn)(unsigned *, unsigned, int);
int main()
{
    const char *names[16] = {"v_min3_u32", "v_sad_u32 clamp", "v_add_u32", "v_min_u32", "v_min3_f32", "v_add_f32 |a|",
                             "v_min_f64", "v_add_f64", "v_cndmask_b32", "v_cmp_lt_f64", "v_mov_b32_dpp", "v_max3_u32",
                             "v_sub_u32", "v_min_f32", "v_pk_add_f32", "v_max_i32"};
    fn fns[16] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>, k<15>};
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 4000;
    unsigned *out; hipMalloc(&out, (size_t)cus * 4 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int op = 0; op < 16; op++) {
        fns[op]<<<cus * 4, 256>>>(out, 1, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        fns[op]<<<cus * 4, 256>>>(out, 1, iters);           // 4 blocks/CU = 4 waves per SIMD
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = 4.0 * iters * 64;      // waves/SIMD x iters x 64 instr per iter
        printf("%-18s %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles at 2.4 GHz)\n", names[op], ms,
               ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
    return 0;
}
