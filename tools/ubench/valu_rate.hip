// valu_rate.hip -- issue cost (shader cycles per wave-instruction per SIMD) of the VALU ops the kernels are
// built from.  Round 3 rewrite, answering the round-2 review:
//   * cycles are COUNTED (s_memtime around the loop, per wave), not derived from an assumed clock; the clock the
//     loop actually ran at is reported too (s_memtime ticks / s_memrealtime ticks x 100 MHz, and HIP-event wall time);
//   * a loop body is ONE asm statement of 64 instructions over 8 independent chains -- the compiler pads the
//     boundary between two asm statements that touch the same register with an s_nop (it cannot see inside),
//     which is what made the round-2 table read 2.5 / 4.4 instead of 2 / 4 (1 s_nop per 8 instructions);
//   * calibration rows: v_fma_f32 (guide: 2 cycles), v_pk_fma_f32, v_add_f64 / v_fma_f64 (16 lanes per clock: 4);
//   * sweep of 1..8 waves per SIMD for the ops of the screening cell, and the dependent min3 -> sad chain of one
//     column of the screening pass (what a single wave can sustain).
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
//   (disassembly of two loops: llvm-objdump -d on the extracted code object, see tools/profile_round.sh)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <map>
#include <stdio.h>
#include <string.h>
#include <vector>

#define G1(i, pre, mid, post) pre "%" #i mid "%" #i post
#define G8(pre, mid, post) G1(0, pre, mid, post) G1(1, pre, mid, post) G1(2, pre, mid, post) G1(3, pre, mid, post) \
                           G1(4, pre, mid, post) G1(5, pre, mid, post) G1(6, pre, mid, post) G1(7, pre, mid, post)
#define B64(pre, mid, post) G8(pre, mid, post) G8(pre, mid, post) G8(pre, mid, post) G8(pre, mid, post) \
                            G8(pre, mid, post) G8(pre, mid, post) G8(pre, mid, post) G8(pre, mid, post)
// One loop body = 64 instructions in one asm statement; chain i is operand %i and appears twice in an instruction's
// text (pre %i mid %i post): as destination and as first source, or -- for ops without a VGPR destination or
// without a chain source -- once for real and once behind the comment character.  %8 %9 = b c (or db dc).
#define BODY_A(pre, mid, post)                                                                           \
    asm volatile(B64(pre, mid, post)                                                                      \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                 : "v"(b), "v"(c) : "s10", "s11", "vcc")
#define BODY_D(pre, mid, post)                                                                           \
    asm volatile(B64(pre, mid, post)                                                                      \
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) \
                 : "v"(db), "v"(dc) : "s10", "s11", "vcc")
#define BODY_DA(pre, mid, post)                                                                          \
    asm volatile(B64(pre, mid, post)                                                                      \
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) \
                 : "v"(b), "v"(c) : "s10", "s11", "vcc")

#define OPS(X)                                                                                           \
    X(0,  "v_fma_f32 (calib: 2)",   BODY_A("v_fma_f32 ", ", ", ", %8, %9\n"))                            \
    X(1,  "v_mul_f32",              BODY_A("v_mul_f32 ", ", ", ", %8\n"))                                \
    X(2,  "v_add_f32",              BODY_A("v_add_f32 ", ", ", ", %8\n"))                                \
    X(3,  "v_pk_fma_f32",           BODY_D("v_pk_fma_f32 ", ", ", ", %8, %9\n"))                         \
    X(4,  "v_add_f64 (calib: 4)",   BODY_D("v_add_f64 ", ", ", ", %8\n"))                                \
    X(5,  "v_fma_f64",              BODY_D("v_fma_f64 ", ", ", ", %8, %9\n"))                            \
    X(6,  "v_min_f64",              BODY_D("v_min_f64 ", ", ", ", %8\n"))                                \
    X(7,  "v_add_u32",              BODY_A("v_add_u32 ", ", ", ", %8\n"))                                \
    X(8,  "v_sub_u32",              BODY_A("v_sub_u32 ", ", ", ", %8\n"))                                \
    X(9,  "v_and_b32",              BODY_A("v_and_b32 ", ", ", ", %8\n"))                                \
    X(10, "v_xor_b32",              BODY_A("v_xor_b32 ", ", ", ", %8\n"))                                \
    X(11, "v_bitop3_b32",           BODY_A("v_bitop3_b32 ", ", ", ", %8, %9 bitop3:0xe4\n"))             \
    X(12, "v_mov_b32",              BODY_A("v_mov_b32 ", ", %8 ; ", "\n"))                               \
    X(13, "v_min3_u32   (cell)",    BODY_A("v_min3_u32 ", ", ", ", %8, %9\n"))                           \
    X(14, "v_sad_u32 clamp (cell)", BODY_A("v_sad_u32 ", ", ", ", %8, %9 clamp\n"))                      \
    X(15, "v_min_u32",              BODY_A("v_min_u32 ", ", ", ", %8\n"))                                \
    X(16, "v_max_u32",              BODY_A("v_max_u32 ", ", ", ", %8\n"))                                \
    X(17, "v_min_f32",              BODY_A("v_min_f32 ", ", ", ", %8\n"))                                \
    X(18, "v_min3_f32",             BODY_A("v_min3_f32 ", ", ", ", %8, %9\n"))                           \
    X(19, "v_add_u32 clamp (e64)",  BODY_A("v_add_u32_e64 ", ", ", ", %8 clamp\n"))                      \
    X(20, "v_sub_u32 clamp (e64)",  BODY_A("v_sub_u32_e64 ", ", ", ", %8 clamp\n"))                      \
    X(21, "v_add3_u32",             BODY_A("v_add3_u32 ", ", ", ", %8, %9\n"))                           \
    X(22, "v_lshlrev_b32",          BODY_A("v_lshlrev_b32 ", ", 1, ", "\n"))                             \
    X(23, "v_cndmask_b32 (sgpr)",   BODY_A("v_cndmask_b32 ", ", ", ", %8, s[10:11]\n"))                  \
    X(24, "v_mov_b32_dpp row_shr:1", BODY_A("v_mov_b32_dpp ", ", %8 row_shr:1 row_mask:0xf bank_mask:0xf ; ", "\n")) \
    X(25, "v_add_u32_dpp row_shr:1", BODY_A("v_add_u32_dpp ", ", ", ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n")) \
    X(26, "v_cmp_lt_f64 (sgpr dst)", BODY_D("v_cmp_lt_f64 s[10:11], ", ", %8 ; ", "\n"))                 \
    X(27, "v_cmp_lt_u32 (sgpr dst)", BODY_A("v_cmp_lt_u32 s[10:11], ", ", %8 ; ", "\n"))                 \
    X(28, "v_pk_min_u16",           BODY_A("v_pk_min_u16 ", ", ", ", %8\n"))                             \
    X(29, "v_pk_add_u16 clamp",     BODY_A("v_pk_add_u16 ", ", ", ", %8 clamp\n"))                       \
    X(30, "v_pk_sub_u16 clamp",     BODY_A("v_pk_sub_u16 ", ", ", ", %8 clamp\n"))                       \
    X(31, "v_pk_minimum3_f16",      BODY_A("v_pk_minimum3_f16 ", ", ", ", %8, %9\n"))                    \
    X(32, "v_sad_u16",              BODY_A("v_sad_u16 ", ", ", ", %8, %9\n"))                            \
    X(33, "v_mad_u32_u24",          BODY_A("v_mad_u32_u24 ", ", ", ", %8, %9\n"))                        \
    X(34, "v_cvt_f64_i32",          BODY_DA("v_cvt_f64_i32 ", ", %8 ; ", "\n"))                          \
    X(35, "v_mul_f64",              BODY_D("v_mul_f64 ", ", ", ", %8\n"))
#define NOPS 36

struct Rec { unsigned long long cyc, real, t0, t1; unsigned hwid, xcc; };

template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned *out, Rec *rec, unsigned seed, int iters)
{
    unsigned a[8], b = seed + threadIdx.x, c = seed * 3 + 1;
    for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x;
    double d[8];
    for (int i = 0; i < 8; i++) d[i] = (double)(seed + i + threadIdx.x);
    double db = (double)b, dc = (double)c;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; it++) {
#define X(idx, name, stmt) if (OP == idx) { stmt; }
        OPS(X)
#undef X
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + (unsigned)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        Rec r; r.cyc = t1 - t0; r.real = r1 - r0; r.t0 = t0; r.t1 = t1;
        r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID: simd 5:4, cu 11:8, sh 12, se 15:13
        r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
        rec[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r;
    }
}

// One column of the screening pass (sk_sdtwq.hip qcolumn<13>): 13 cells, each min3 -> sad, cell k waiting for
// cell k-1 -- the dependent chain a single wave has to get through per step.  CH independent columns per body.
template <int CH>
__global__ __launch_bounds__(1024) void kchain(unsigned *out, Rec *rec, unsigned seed, int iters)
{
    unsigned a[CH][2], x = seed + threadIdx.x, y = seed * 7 + 3;
    for (int i = 0; i < CH; i++) { a[i][0] = seed + i; a[i][1] = seed + 2 * i + threadIdx.x; }
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if constexpr (CH == 1) {
                asm volatile("v_min3_u32 %0, %1, %0, %2\n v_sad_u32 %0, %2, %3, %0 clamp\n"
                             "v_min3_u32 %1, %0, %1, %2\n v_sad_u32 %1, %2, %3, %1 clamp\n"
                             : "+v"(a[0][0]), "+v"(a[0][1]) : "v"(x), "v"(y));
            } else {
                asm volatile("v_min3_u32 %0, %1, %0, %4\n v_min3_u32 %2, %3, %2, %4\n"
                             "v_sad_u32 %0, %4, %5, %0 clamp\n v_sad_u32 %2, %4, %5, %2 clamp\n"
                             "v_min3_u32 %1, %0, %1, %4\n v_min3_u32 %3, %2, %3, %4\n"
                             "v_sad_u32 %1, %4, %5, %1 clamp\n v_sad_u32 %3, %4, %5, %3 clamp\n"
                             : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]) : "v"(x), "v"(y));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned s = 0;
    for (int i = 0; i < CH; i++) s += a[i][0] + a[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        Rec r; r.cyc = t1 - t0; r.real = r1 - r0; r.t0 = t0; r.t1 = t1;
        r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID: simd 5:4, cu 11:8, sh 12, se 15:13
        r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
        rec[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r;
    }
}

typedef void (*fn)(unsigned *, Rec *, unsigned, int);
template <int... I> struct seq {};
template <int N, int... I> struct mk : mk<N - 1, N - 1, I...> {};
template <int... I> struct mk<0, I...> { typedef seq<I...> type; };
template <int... I> void fill(fn *f, seq<I...>) { fn t[] = {k<I>...}; memcpy(f, t, sizeof t); }

struct Result { double cyc_per_instr, cyc_span, ghz_mem, ghz_wall, ms; int wmin, wmax; };

// Placement is forced, not assumed: up to 4 waves per SIMD = ONE workgroup of 256 x wps threads per CU (96 KB of
// dynamic LDS each, so two cannot share a CU); 6 or 8 = two workgroups of 128 x wps threads per CU (64 KB each, a third
// does not fit).  Every wave records the SIMD it ran on (HW_ID / XCC_ID); the table reports the fewest and the most
// waves any SIMD hosted.
static Result run(fn f, int cus, int wps, int iters, double instr_per_iter, unsigned *out, Rec *rec, hipEvent_t e0, hipEvent_t e1)
{
    const bool two = wps > 4;
    const int blocks = two ? 2 * cus : cus, threads = two ? 128 * wps : 256 * wps;
    const size_t lds = two ? 64 << 10 : 96 << 10;
    hipFuncSetAttribute((const void *)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(threads), lds, 0, out, rec, 1u, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(threads), lds, 0, out, rec, 1u, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const size_t nw = (size_t)blocks * (threads / 64);
    std::vector<Rec> h(nw);
    hipMemcpy(h.data(), rec, h.size() * sizeof(Rec), hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_simd;
    std::map<unsigned, std::pair<unsigned long long, unsigned long long>> span;   // SIMD -> (first start, last end)
    for (auto &r : h) {
        const unsigned key = ((r.xcc & 0xF) << 16) | (r.hwid & 0xFF30);
        per_simd[key]++;
        auto it = span.find(key);
        if (it == span.end()) span[key] = std::make_pair(r.t0, r.t1);
        else { it->second.first = std::min(it->second.first, r.t0); it->second.second = std::max(it->second.second, r.t1); }
    }
    std::vector<double> spans;
    for (auto &kv : span) spans.push_back((double)(kv.second.second - kv.second.first));
    std::sort(spans.begin(), spans.end());
    int wmin = 1 << 30, wmax = 0;
    for (auto &kv : per_simd) { wmin = std::min(wmin, kv.second); wmax = std::max(wmax, kv.second); }
    std::vector<double> cyc, real;
    for (auto &r : h) { cyc.push_back((double)r.cyc); real.push_back((double)r.real); }
    std::sort(cyc.begin(), cyc.end()); std::sort(real.begin(), real.end());
    const double mc = cyc[cyc.size() / 2], mr = real[real.size() / 2];
    Result r;
    // a wave's loop lasted mc cycles, during which its SIMD issued the loops of all wps co-resident waves
    r.cyc_per_instr = mc / (wps * iters * instr_per_iter);
    // the stagger-proof version: from the first start to the last end of the waves that shared a SIMD
    r.cyc_span = spans[spans.size() / 2] / (wps * iters * instr_per_iter);
    r.ghz_mem = mc / (mr / 100e6) / 1e9;                   // s_memrealtime: constant 100 MHz
    r.ghz_wall = mc / (ms * 1e-3) / 1e9;                   // loop cycles / kernel wall time (launch + ramp included)
    r.ms = ms;
    r.wmin = (int)per_simd.size() == cus * 4 ? wmin : 0;   // a SIMD that hosted no wave at all
    r.wmax = wmax;
    return r;
}

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
    unsigned *out; hipMalloc(&out, (size_t)cus * 2048 * 4);
    Rec *rec; hipMalloc(&rec, (size_t)cus * 32 * sizeof(Rec));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("# %s, %d CUs.  cyc/ins = shader cycles per wave-instruction per SIMD = median over SIMDs of (s_memtime from the "
           "first loop start to the last loop end of the waves sharing the SIMD) / (waves per SIMD x instructions per "
           "wave); (wave) = the same from one wave's own loop, which reads low because the waves of a SIMD start "
           "staggered.  64 instructions per loop body in ONE asm statement (no s_nop inside, "
           "profiles/r03_valu_rate_disasm.txt), 8 independent chains.  clk = s_memtime ticks per s_memrealtime second "
           "(100 MHz reference) while the loop ran.  waves/SIMD = fewest..most waves any of the %d SIMDs hosted (from "
           "HW_ID / XCC_ID of every wave; 0 = some SIMD hosted none).\n", p.name, cus, cus * 4);
    printf("# --- table 1: 4 waves per SIMD ---\n");
    printf("%-28s %8s %8s %8s %9s %10s\n", "op", "cyc/ins", "(wave)", "clk GHz", "wall ms", "waves/SIMD");
    for (int op = 0; op < NOPS; op++) {
        Result r = run(fns[op], cus, 4, iters, 64.0, out, rec, e0, e1);
        printf("%-28s %8.2f %8.2f %8.3f %9.3f %7d..%d\n", names[op], r.cyc_span, r.cyc_per_instr, r.ghz_mem, r.ms, r.wmin, r.wmax);
    }
    const int ws[] = {1, 2, 3, 4, 6, 8};
    printf("# --- table 2: waves per SIMD sweep (cyc/ins) ---\n");
    const int sweep_ops[] = {0, 7, 4, 13, 14};
    printf("%-28s", "op \\ waves per SIMD");
    for (int w : ws) printf(" %6d", w);
    printf("\n");
    for (int op : sweep_ops) {
        printf("%-28s", names[op]);
        for (int w : ws) printf(" %6.2f", run(fns[op], cus, w, iters, 64.0, out, rec, e0, e1).cyc_span);
        printf("\n");
    }
    printf("# --- table 3: the screening cell's dependent chain (min3 -> sad -> min3 -> ...), cyc/ins; 4.00 per "
           "instruction = 8.00 per cell is the issue floor if both ops are 4-cycle ops ---\n");
    printf("%-28s", "chains per wave \\ waves");
    for (int w : ws) printf(" %6d", w);
    printf("\n%-28s", "1 column (as in k_sdtw_q)");
    for (int w : ws) printf(" %6.2f", run(kchain<1>, cus, w, iters, 64.0, out, rec, e0, e1).cyc_span);
    printf("\n%-28s", "2 interleaved columns");
    for (int w : ws) printf(" %6.2f", run(kchain<2>, cus, w, iters, 128.0, out, rec, e0, e1).cyc_span);
    printf("\n# --- table 4: ten times longer loops (launch and ramp amortised): cyc/ins from the SIMD span, and from the "
           "kernel's wall time x clk / instructions per SIMD ---\n");
    const int long_ops[] = {0, 7, 4, 13, 14};
    for (int op : long_ops) {
        Result r = run(fns[op], cus, 4, iters * 10, 64.0, out, rec, e0, e1);
        printf("%-28s span %5.2f   wave %5.2f   wall-derived %5.2f   clk %.3f GHz   wall %.3f ms\n", names[op], r.cyc_span,
               r.cyc_per_instr, r.ms * 1e-3 * r.ghz_mem * 1e9 / (4.0 * iters * 10 * 64.0), r.ghz_mem, r.ms);
    }
    printf("\n");
    return 0;
}
