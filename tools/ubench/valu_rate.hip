// valu_rate.hip -- issue cost (cycles per wave-instruction per SIMD) of the VALU ops the kernels
// are built from, measured with 8 independent chains per wave and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

// X(index, name, asm statement on chain register; U = 32-bit chain a[i], D = 64-bit chain d[i])
#define OPS(X)                                                                                         \
    X(0, "v_min3_u32", asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))       \
    X(1, "v_sad_u32 clamp", asm volatile("v_sad_u32 %0, %0, %1, %2 clamp" : "+v"(a[i]) : "v"(b), "v"(c))) \
    X(2, "v_add_u32", asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))                     \
    X(3, "v_min_u32", asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))                     \
    X(4, "v_min3_f32", asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))       \
    X(5, "v_add_f32 |a|", asm volatile("v_add_f32 %0, |%0|, %1" : "+v"(a[i]) : "v"(b)))               \
    X(6, "v_min_f64", asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db)))                    \
    X(7, "v_add_f64", asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db)))                    \
    X(8, "v_cndmask_b32", asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b)))        \
    X(9, "v_cmp_lt_f64", asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(db) : "vcc"))     \
    X(10, "v_mov_b32_dpp", asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b))) \
    X(11, "v_max3_u32", asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))      \
    X(12, "v_sub_u32", asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))                    \
    X(13, "v_min_f32", asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))                    \
    X(14, "v_pk_add_f32", asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(db)))             \
    X(15, "v_max_i32", asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))                    \
    X(16, "v_mad_u32_u24", asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c))) \
    X(17, "v_mul_u32_u24", asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b)))            \
    X(18, "v_bfe_u32", asm volatile("v_bfe_u32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b)))                 \
    X(19, "v_and_b32", asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))                    \
    X(20, "v_bfi_b32", asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))        \
    X(21, "v_and_or_b32", asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))  \
    X(22, "v_add3_u32", asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))      \
    X(23, "v_lshl_add_u32", asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b)))       \
    X(24, "v_cmp_lt_u32 vcc", asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc")) \
    X(25, "v_bitop3_b32", asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x32" : "+v"(a[i]) : "v"(b), "v"(c))) \
    X(26, "v_cvt_f64_i32", asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(a[i])))             \
    X(27, "v_mul_f64", asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db)))                   \
    X(28, "v_fma_f64", asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(db)))               \
    X(29, "v_pk_sub_u16", asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b)))              \
    X(30, "v_pk_max_u16", asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b)))              \
    X(31, "v_mul_lo_u32", asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))              \
    X(32, "v_alignbit_b32", asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a[i]) : "v"(b)))      \
    X(33, "v_lshlrev_b32", asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(a[i])))                      \
    X(34, "v_xor_b32", asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))                    \
    X(35, "v_cndmask_b32 sgpr", asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(m))) \
    X(36, "v_cmp_ge_u32 sgpr", asm volatile("v_cmp_ge_u32 %0, %1, %2" : "=s"(m) : "v"(a[i]), "v"(b))) \
    X(37, "v_mov_b32", asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b)))                        \
    X(38, "v_sub_f64 (add neg)", asm volatile("v_add_f64 %0, %0, -%1" : "+v"(d[i]) : "v"(db)))         \
    X(39, "v_cvt_i32_f64", asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i])))             \
    X(40, "v_pk_add_u16", asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b)))              \
    X(41, "v_mbcnt_lo", asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b)))          \
    X(42, "v_lshl_or_b32", asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b)))
#define NOPS 43

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed, int iters)
{
    unsigned a[8], b = seed + threadIdx.x, c = seed * 3 + 1;
    for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x;
    double d[8];
    for (int i = 0; i < 8; i++) d[i] = (double)(seed + i + threadIdx.x);
    double db = (double)b;
    unsigned long long m = seed;
    for (int it = 0; it < iters; it++) {
#define X(idx, name, stmt) if (OP == idx) { stmt; }
#define ONE(ii) { constexpr int i = ii; OPS(X) }
        for (int u = 0; u < 8; u++) { ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5) ONE(6) ONE(7) }
#undef X
    }
    unsigned s = (unsigned)m;
    for (int i = 0; i < 8; i++) s += a[i] + (unsigned)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*fn)(unsigned *, unsigned, int);
template <int... I> struct seq {};
template <int N, int... I> struct mk : mk<N - 1, N - 1, I...> {};
template <int... I> struct mk<0, I...> { typedef seq<I...> type; };
template <int... I> void fill(fn *f, seq<I...>) { fn t[] = {k<I>...}; memcpy(f, t, sizeof t); }

int main()
{
    const char *names[NOPS];
#define X(idx, name, stmt) names[idx] = name;
    OPS(X)
#undef X
    fn fns[NOPS];
    fill(fns, mk<NOPS>::type());
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 4000;
    unsigned *out; hipMalloc(&out, (size_t)cus * 4 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int op = 0; op < NOPS; op++) {
        fns[op]<<<cus * 4, 256>>>(out, 1, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        fns[op]<<<cus * 4, 256>>>(out, 1, iters);           // 4 blocks/CU = 4 waves per SIMD
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = 4.0 * iters * 64;      // waves/SIMD x iters x 64 instr per iter
        printf("%-20s %.3f ms  -> %.2f cycles per wave-instr per SIMD at 2.4 GHz\n", names[op], ms,
               ms * 1e6 / instr_per_simd * 2.4);
    }
    return 0;
}
