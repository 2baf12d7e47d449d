// Dev aid (GPU): VALU issue rate of the instructions the VP8 kernels are made of, at 1 / 2 / 4 waves per SIMD,
// as independent streams (16 registers round-robin) and as one dependent chain.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_ubench tools/valu_ubench.hip && /tmp/valu_ubench
// Prints shader cycles (s_memtime) per wave-instruction as seen by one wave, and the SIMD-level rate
// (= that / waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL(NAME, INDEP_ASM, DEP_ASM)                                                                       \
    __global__ void __launch_bounds__(256) k_##NAME(unsigned long long *out, int iters, int dep, unsigned seed) \
    {                                                                                                          \
        unsigned r[16];                                                                                        \
        for (int i = 0; i < 16; i++) r[i] = seed * (threadIdx.x + 1) + i * 0x01010101u;                        \
        unsigned a = seed | 0x00030003u, b = (seed >> 3) | 0x00010001u;                                        \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                  \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                    \
        if (!dep) {                                                                                            \
            for (int it = 0; it < iters; it++) {                                                               \
                asm volatile(INDEP_ASM INDEP_ASM INDEP_ASM INDEP_ASM INDEP_ASM INDEP_ASM INDEP_ASM INDEP_ASM                                                                         \
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]),   \
                               "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), \
                               "+v"(r[14]), "+v"(r[15])                                                        \
                             : "v"(a), "v"(b));                                                                \
            }                                                                                                  \
        } else {                                                                                               \
            for (int it = 0; it < iters; it++) {                                                               \
                asm volatile(DEP_ASM DEP_ASM DEP_ASM DEP_ASM DEP_ASM DEP_ASM DEP_ASM DEP_ASM : "+v"(r[0]) : "v"(a), "v"(b));                                           \
            }                                                                                                  \
        }                                                                                                      \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                  \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                    \
        unsigned s = 0;                                                                                        \
        for (int i = 0; i < 16; i++) s ^= r[i];                                                                \
        if (threadIdx.x % 64 == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = (t1 - t0) + (s == 0x12345 ? 1 : 0); \
    }

// three-operand form: op dst, src0(=dst), a   /  two-source ops written out per instruction
#define I3(OP)                                                                                               \
    OP " %0, %0, %16\n" OP " %1, %1, %17\n" OP " %2, %2, %16\n" OP " %3, %3, %17\n" OP " %4, %4, %16\n"        \
    OP " %5, %5, %17\n" OP " %6, %6, %16\n" OP " %7, %7, %17\n" OP " %8, %8, %16\n" OP " %9, %9, %17\n"        \
    OP " %10, %10, %16\n" OP " %11, %11, %17\n" OP " %12, %12, %16\n" OP " %13, %13, %17\n" OP " %14, %14, %16\n" \
    OP " %15, %15, %17\n"
#define D3(OP)                                                                                               \
    OP " %0, %0, %1\n" OP " %0, %0, %2\n" OP " %0, %0, %1\n" OP " %0, %0, %2\n" OP " %0, %0, %1\n" OP " %0, %0, %2\n" \
    OP " %0, %0, %1\n" OP " %0, %0, %2\n" OP " %0, %0, %1\n" OP " %0, %0, %2\n" OP " %0, %0, %1\n" OP " %0, %0, %2\n" \
    OP " %0, %0, %1\n" OP " %0, %0, %2\n" OP " %0, %0, %1\n" OP " %0, %0, %2\n"
// four-operand form: op dst, dst, a, b
#define I4(OP)                                                                                               \
    OP " %0, %0, %16, %17\n" OP " %1, %1, %16, %17\n" OP " %2, %2, %16, %17\n" OP " %3, %3, %16, %17\n"        \
    OP " %4, %4, %16, %17\n" OP " %5, %5, %16, %17\n" OP " %6, %6, %16, %17\n" OP " %7, %7, %16, %17\n"        \
    OP " %8, %8, %16, %17\n" OP " %9, %9, %16, %17\n" OP " %10, %10, %16, %17\n" OP " %11, %11, %16, %17\n"    \
    OP " %12, %12, %16, %17\n" OP " %13, %13, %16, %17\n" OP " %14, %14, %16, %17\n" OP " %15, %15, %16, %17\n"
#define D4(OP)                                                                                               \
    OP " %0, %0, %1, %2\n" OP " %0, %0, %1, %2\n" OP " %0, %0, %1, %2\n" OP " %0, %0, %1, %2\n"                \
    OP " %0, %0, %1, %2\n" OP " %0, %0, %1, %2\n" OP " %0, %0, %1, %2\n" OP " %0, %0, %1, %2\n"                \
    OP " %0, %0, %1, %2\n" OP " %0, %0, %1, %2\n" OP " %0, %0, %1, %2\n" OP " %0, %0, %1, %2\n"                \
    OP " %0, %0, %1, %2\n" OP " %0, %0, %1, %2\n" OP " %0, %0, %1, %2\n" OP " %0, %0, %1, %2\n"
#define I2(OP)                                                                                               \
    OP " %0, %1\n" OP " %1, %2\n" OP " %2, %3\n" OP " %3, %4\n" OP " %4, %5\n" OP " %5, %6\n" OP " %6, %7\n" OP " %7, %8\n" \
    OP " %8, %9\n" OP " %9, %10\n" OP " %10, %11\n" OP " %11, %12\n" OP " %12, %13\n" OP " %13, %14\n" OP " %14, %15\n" OP " %15, %0\n"
#define D2(OP)                                                                                               \
    OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n" \
    OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n"
// DPP move: v_mov_b32 dst, dst wave_shr:1
#define IDPP                                                                                                 \
    "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b32_dpp %8, %9 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %9, %10 wave_shr:1 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b32_dpp %10, %11 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %11, %12 wave_shr:1 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b32_dpp %12, %13 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %13, %14 wave_shr:1 row_mask:0xf bank_mask:0xf\n" \
    "v_mov_b32_dpp %14, %15 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %15, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define DDPP                                                                                                 \
    "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n" \
    "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n" \
    "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n" \
    "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n" \
    "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n" \
    "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n" \
    "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n" \
    "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"

KERNEL(add_u32, I3("v_add_u32"), D3("v_add_u32"))
KERNEL(xor_b32, I3("v_xor_b32"), D3("v_xor_b32"))
KERNEL(pk_add_u16, I3("v_pk_add_u16"), D3("v_pk_add_u16"))
KERNEL(pk_sub_i16_clamp, I3("v_pk_sub_i16") , D3("v_pk_sub_i16"))
KERNEL(pk_max_i16, I3("v_pk_max_i16"), D3("v_pk_max_i16"))
KERNEL(pk_mul_lo_u16, I3("v_pk_mul_lo_u16"), D3("v_pk_mul_lo_u16"))
KERNEL(pk_ashrrev_i16, I3("v_pk_ashrrev_i16"), D3("v_pk_ashrrev_i16"))
KERNEL(pk_mad_i16, I4("v_pk_mad_i16"), D4("v_pk_mad_i16"))
KERNEL(perm_b32, I4("v_perm_b32"), D4("v_perm_b32"))
KERNEL(alignbyte_b32, I4("v_alignbyte_b32"), D4("v_alignbyte_b32"))
KERNEL(lerp_u8, I4("v_lerp_u8"), D4("v_lerp_u8"))
KERNEL(sad_u8, I4("v_sad_u8"), D4("v_sad_u8"))
KERNEL(mul_i32_i24, I3("v_mul_i32_i24"), D3("v_mul_i32_i24"))
KERNEL(mad_i32_i24, I4("v_mad_i32_i24"), D4("v_mad_i32_i24"))
KERNEL(mul_hi_i32_i24, I3("v_mul_hi_i32_i24"), D3("v_mul_hi_i32_i24"))
KERNEL(mul_lo_u32, I3("v_mul_lo_u32"), D3("v_mul_lo_u32"))
KERNEL(mul_hi_i32, I3("v_mul_hi_i32"), D3("v_mul_hi_i32"))
KERNEL(bfe_i32, I4("v_bfe_i32"), D4("v_bfe_i32"))
KERNEL(and_or_b32, I4("v_and_or_b32"), D4("v_and_or_b32"))
KERNEL(add3_u32, I4("v_add3_u32"), D4("v_add3_u32"))
KERNEL(med3_i32, I4("v_med3_i32"), D4("v_med3_i32"))
KERNEL(dot2_i32_i16, I4("v_dot2_i32_i16"), D4("v_dot2_i32_i16"))
KERNEL(dot4_i32_i8, I4("v_dot4_i32_i8"), D4("v_dot4_i32_i8"))
KERNEL(pk_fma_f32_na, I3("v_add_f32"), D3("v_add_f32"))
KERNEL(and_b32, I3("v_and_b32"), D3("v_and_b32"))
KERNEL(lshlrev_b32, I3("v_lshlrev_b32"), D3("v_lshlrev_b32"))
KERNEL(ashrrev_i32, I3("v_ashrrev_i32"), D3("v_ashrrev_i32"))
KERNEL(sub_u32, I3("v_sub_u32"), D3("v_sub_u32"))
KERNEL(max_i32, I3("v_max_i32"), D3("v_max_i32"))
KERNEL(min_u32, I3("v_min_u32"), D3("v_min_u32"))
KERNEL(mul_u32_u24, I3("v_mul_u32_u24"), D3("v_mul_u32_u24"))
KERNEL(add_u16, I3("v_add_u16"), D3("v_add_u16"))
KERNEL(mul_f32, I3("v_mul_f32"), D3("v_mul_f32"))
KERNEL(fma_f32, I4("v_fma_f32"), D4("v_fma_f32"))
KERNEL(fmac_f32, I3("v_fmac_f32"), D3("v_fmac_f32"))
KERNEL(pk_add_f16, I3("v_pk_add_f16"), D3("v_pk_add_f16"))
KERNEL(bfi_b32, I4("v_bfi_b32"), D4("v_bfi_b32"))
KERNEL(lshl_add_u32, I4("v_lshl_add_u32"), D4("v_lshl_add_u32"))
KERNEL(or3_b32, I4("v_or3_b32"), D4("v_or3_b32"))
KERNEL(max3_i32, I4("v_max3_i32"), D4("v_max3_i32"))
KERNEL(cvt_f32_i32, I2("v_cvt_f32_i32"), D2("v_cvt_f32_i32"))
KERNEL(mov_b32, I2("v_mov_b32"), D2("v_mov_b32"))
KERNEL(mov_dpp_wave_shr, IDPP, DDPP)

typedef void (*kern_t)(unsigned long long *, int, int, unsigned);
struct Entry { const char *name; kern_t k; };
#define E(N) { #N, k_##N }
static Entry entries[] = { E(add_u32), E(xor_b32), E(pk_add_u16), E(pk_sub_i16_clamp), E(pk_max_i16), E(pk_mul_lo_u16), E(pk_ashrrev_i16),
    E(pk_mad_i16), E(perm_b32), E(alignbyte_b32), E(lerp_u8), E(sad_u8), E(mul_i32_i24), E(mad_i32_i24), E(mul_hi_i32_i24),
    E(mul_lo_u32), E(mul_hi_i32), E(bfe_i32), E(and_or_b32), E(add3_u32), E(med3_i32), E(dot2_i32_i16), E(dot4_i32_i8),
    E(pk_fma_f32_na), E(and_b32), E(lshlrev_b32), E(ashrrev_i32), E(sub_u32), E(max_i32), E(min_u32), E(mul_u32_u24), E(add_u16), E(mul_f32), E(fma_f32), E(fmac_f32), E(pk_add_f16), E(bfi_b32), E(lshl_add_u32), E(or3_b32), E(max3_i32), E(cvt_f32_i32), E(mov_b32), E(mov_dpp_wave_shr) };

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int ncu = prop.multiProcessorCount;
    unsigned long long *d;
    hipMalloc(&d, 8 * ncu * 4 * 8);
    std::vector<unsigned long long> h(ncu * 4 * 8);
    const int iters = 1024;
    printf("%-20s %s\n", "instruction", "cycles per wave-instruction seen by one wave [indep: 1,2,4 waves/SIMD | dep chain: 1 wave/SIMD]  (SIMD-level = value / waves)");
    for (const Entry &e : entries) {
        printf("%-20s", e.name);
        for (int dep = 0; dep < 2; dep++) {
            for (int w = 1; w <= (dep ? 1 : 4); w *= 2) {
                const int grid = ncu * w;      // 256-thread blocks: 4 waves, one per SIMD; w blocks per CU
                hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, d, 16, dep, 12345u);
                hipDeviceSynchronize();
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                hipEventRecord(a);
                hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, d, iters, dep, 12345u);
                hipEventRecord(b);
                hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, a, b);
                hipMemcpy(h.data(), d, 8 * grid * 4, hipMemcpyDeviceToHost);
                std::vector<unsigned long long> v(h.begin(), h.begin() + grid * 4);
                std::sort(v.begin(), v.end());
                const double cyc = (double)v[v.size() / 2] / (iters * 128.0);
                printf("  %s%dw: %5.2f cyc (%.0f us)", dep ? "dep " : "", w, cyc, ms * 1e3);
            }
        }
        printf("\n");
    }
    return 0;
}
