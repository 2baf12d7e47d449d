// Dev aid (GPU): issue rate of the fp32 / conversion instructions a float formulation of the six-tap filters would be made of, at
// 1 / 2 / 3 / 4 waves per SIMD (independent streams), beside two references of tools/valu_ubench.hip's two classes (v_add_u32: full
// rate with two waves; v_pk_mad_u16: 4.15 cycles however many waves).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_ubench_f32 tools/valu_ubench_f32.hip && /tmp/valu_ubench_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

// 32-bit destinations: sixteen registers round-robin, sources a / b (32-bit)
#define K32(NAME, BODY)                                                                                                 \
    __global__ void __launch_bounds__(256) k_##NAME(unsigned long long *out, int iters, unsigned seed)                    \
    {                                                                                                                   \
        unsigned r[16];                                                                                                 \
        for (int i = 0; i < 16; i++) r[i] = seed * (threadIdx.x + 1) + i * 0x01010101u;                                 \
        unsigned a = seed | 0x00030003u, b = (seed >> 3) | 0x00010001u;                                                 \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                           \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                             \
        for (int it = 0; it < iters; it++)                                                                              \
            asm volatile(BODY BODY BODY BODY BODY BODY BODY BODY                                                        \
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
                           "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) \
                         : "v"(a), "v"(b));                                                                             \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                           \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                             \
        unsigned s = 0;                                                                                                 \
        for (int i = 0; i < 16; i++) s ^= r[i];                                                                         \
        if (threadIdx.x % 64 == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = (t1 - t0) + (s == 0x12345 ? 1 : 0); \
    }
// 64-bit destinations (packed fp32): eight register pairs round-robin, sources a / b (64-bit)
#define K64(NAME, BODY)                                                                                                 \
    __global__ void __launch_bounds__(256) k_##NAME(unsigned long long *out, int iters, unsigned seed)                    \
    {                                                                                                                   \
        double r[8];                                                                                                    \
        for (int i = 0; i < 8; i++) r[i] = __hiloint2double(0x3f800000 + i, 0x3f800000 + (threadIdx.x & 7));            \
        double a = __hiloint2double(0x3f000000, 0x3f000000), b = __hiloint2double(0x3e800000, 0x3e800000);              \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                           \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                             \
        for (int it = 0; it < iters; it++)                                                                              \
            asm volatile(BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY                \
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
                         : "v"(a), "v"(b));                                                                             \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                           \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                             \
        double s = 0;                                                                                                   \
        for (int i = 0; i < 8; i++) s += r[i];                                                                          \
        if (threadIdx.x % 64 == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = (t1 - t0) + (s == 0.12345 ? 1 : 0); \
    }

#define I3(OP)                                                                                               \
    OP " %0, %0, %16\n" OP " %1, %1, %17\n" OP " %2, %2, %16\n" OP " %3, %3, %17\n" OP " %4, %4, %16\n"        \
    OP " %5, %5, %17\n" OP " %6, %6, %16\n" OP " %7, %7, %17\n" OP " %8, %8, %16\n" OP " %9, %9, %17\n"        \
    OP " %10, %10, %16\n" OP " %11, %11, %17\n" OP " %12, %12, %16\n" OP " %13, %13, %17\n" OP " %14, %14, %16\n" \
    OP " %15, %15, %17\n"
#define I4(OP)                                                                                               \
    OP " %0, %0, %16, %17\n" OP " %1, %1, %16, %17\n" OP " %2, %2, %16, %17\n" OP " %3, %3, %16, %17\n"        \
    OP " %4, %4, %16, %17\n" OP " %5, %5, %16, %17\n" OP " %6, %6, %16, %17\n" OP " %7, %7, %16, %17\n"        \
    OP " %8, %8, %16, %17\n" OP " %9, %9, %16, %17\n" OP " %10, %10, %16, %17\n" OP " %11, %11, %16, %17\n"    \
    OP " %12, %12, %16, %17\n" OP " %13, %13, %16, %17\n" OP " %14, %14, %16, %17\n" OP " %15, %15, %16, %17\n"
#define I2(OP)                                                                                               \
    OP " %0, %1\n" OP " %1, %2\n" OP " %2, %3\n" OP " %3, %4\n" OP " %4, %5\n" OP " %5, %6\n" OP " %6, %7\n" OP " %7, %8\n" \
    OP " %8, %9\n" OP " %9, %10\n" OP " %10, %11\n" OP " %11, %12\n" OP " %12, %13\n" OP " %13, %14\n" OP " %14, %15\n" OP " %15, %0\n"
// four-operand form whose third source is the destination (v_cvt_pk_u8_f32 d, value, byte, d)
#define I4D(OP)                                                                                              \
    OP " %0, %16, 0, %0\n" OP " %1, %17, 1, %1\n" OP " %2, %16, 2, %2\n" OP " %3, %17, 3, %3\n"                \
    OP " %4, %16, 0, %4\n" OP " %5, %17, 1, %5\n" OP " %6, %16, 2, %6\n" OP " %7, %17, 3, %7\n"                \
    OP " %8, %16, 0, %8\n" OP " %9, %17, 1, %9\n" OP " %10, %16, 2, %10\n" OP " %11, %17, 3, %11\n"            \
    OP " %12, %16, 0, %12\n" OP " %13, %17, 1, %13\n" OP " %14, %16, 2, %14\n" OP " %15, %17, 3, %15\n"
// 64-bit: op d, d, a  /  op d, d, a, b  /  op d, a, b, d
#define P3(OP) OP " %0, %0, %8\n" OP " %1, %1, %9\n" OP " %2, %2, %8\n" OP " %3, %3, %9\n" OP " %4, %4, %8\n" OP " %5, %5, %9\n" OP " %6, %6, %8\n" OP " %7, %7, %9\n"
#define P4(OP) OP " %0, %8, %9, %0\n" OP " %1, %9, %8, %1\n" OP " %2, %8, %9, %2\n" OP " %3, %9, %8, %3\n" OP " %4, %8, %9, %4\n" OP " %5, %9, %8, %5\n" OP " %6, %8, %9, %6\n" OP " %7, %9, %8, %7\n"
// ... with one source's low half broadcast to both lanes (op_sel_hi:[1,0,1]: a filter tap from one register of a pair)
#define P4S(OP) OP " %0, %8, %9, %0 op_sel_hi:[1,0,1]\n" OP " %1, %9, %8, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n" OP " %2, %8, %9, %2 op_sel_hi:[1,0,1]\n" OP " %3, %9, %8, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n" \
                OP " %4, %8, %9, %4 op_sel_hi:[1,0,1]\n" OP " %5, %9, %8, %5 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n" OP " %6, %8, %9, %6 op_sel_hi:[1,0,1]\n" OP " %7, %9, %8, %7 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"

K32(add_u32, I3("v_add_u32"))
K32(pk_mad_u16, I4("v_pk_mad_u16"))
K32(or_b32, I3("v_or_b32"))
K32(lshrrev_b32, I3("v_lshrrev_b32"))
K32(cndmask_b32, I3("v_cndmask_b32"))
K32(cvt_f32_ubyte0, I2("v_cvt_f32_ubyte0"))
K32(cvt_f32_ubyte1, I2("v_cvt_f32_ubyte1"))
K32(cvt_f32_ubyte2, I2("v_cvt_f32_ubyte2"))
K32(cvt_f32_ubyte3, I2("v_cvt_f32_ubyte3"))
K32(cvt_f32_u32, I2("v_cvt_f32_u32"))
K32(cvt_u32_f32, I2("v_cvt_u32_f32"))
K32(cvt_i32_f32, I2("v_cvt_i32_f32"))
K32(floor_f32, I2("v_floor_f32"))
K32(rndne_f32, I2("v_rndne_f32"))
K32(min_f32, I3("v_min_f32"))
K32(max_f32, I3("v_max_f32"))
K32(add_f32, I3("v_add_f32"))
K32(med3_f32, I4("v_med3_f32"))
K32(fma_f32, I4("v_fma_f32"))
K32(cvt_pk_u8_f32, I4D("v_cvt_pk_u8_f32"))
K32(cvt_pk_u16_u32, I3("v_cvt_pk_u16_u32"))
K32(cvt_pkrtz_f16_f32, I3("v_cvt_pkrtz_f16_f32"))
K32(pk_fma_f16, I4("v_pk_fma_f16"))
K32(pk_min_f16, I3("v_pk_min_f16"))
K32(fma_mix_f32, I4("v_fma_mix_f32"))
K64(pk_fma_f32, P4("v_pk_fma_f32"))
K64(pk_fma_f32_bcast, P4S("v_pk_fma_f32"))
K64(pk_mul_f32, P3("v_pk_mul_f32"))
K64(pk_add_f32, P3("v_pk_add_f32"))
K64(pk_mov_b32, "v_pk_mov_b32 %0, %8, %9\n v_pk_mov_b32 %1, %9, %8 op_sel:[1,0]\n v_pk_mov_b32 %2, %8, %9\n v_pk_mov_b32 %3, %9, %8 op_sel:[1,0]\n"
                "v_pk_mov_b32 %4, %8, %9\n v_pk_mov_b32 %5, %9, %8 op_sel:[1,0]\n v_pk_mov_b32 %6, %8, %9\n v_pk_mov_b32 %7, %9, %8 op_sel:[1,0]\n")

typedef void (*kern_t)(unsigned long long *, int, unsigned);
struct Entry { const char *name; kern_t k; int per_iter; };
#define E32(N) { #N, k_##N, 128 }
#define E64(N) { #N, k_##N, 128 }
static Entry entries[] = { E32(add_u32), E32(pk_mad_u16), E32(or_b32), E32(lshrrev_b32), E32(cndmask_b32), E32(cvt_f32_ubyte0), E32(cvt_f32_ubyte1),
    E32(cvt_f32_ubyte2), E32(cvt_f32_ubyte3), E32(cvt_f32_u32), E32(cvt_u32_f32), E32(cvt_i32_f32), E32(floor_f32), E32(rndne_f32), E32(min_f32),
    E32(max_f32), E32(add_f32), E32(med3_f32), E32(fma_f32), E32(cvt_pk_u8_f32), E32(cvt_pk_u16_u32), E32(cvt_pkrtz_f16_f32), E32(pk_fma_f16),
    E32(pk_min_f16), E32(fma_mix_f32), E64(pk_fma_f32), E64(pk_fma_f32_bcast), E64(pk_mul_f32), E64(pk_add_f32), E64(pk_mov_b32) };

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int ncu = prop.multiProcessorCount;
    unsigned long long *d;
    hipMalloc(&d, 8 * ncu * 4 * 8);
    std::vector<unsigned long long> h(ncu * 4 * 8);
    const int iters = 1024;
    printf("%-20s %s\n", "instruction", "cycles per wave-instruction seen by one wave, independent streams, 1 / 2 / 3 / 4 waves per SIMD (SIMD-level = value / waves)");
    for (const Entry &e : entries) {
        printf("%-20s", e.name);
        for (int w = 1; w <= 4; w++) {
            const int grid = ncu * w;      // 256-thread blocks: 4 waves, one per SIMD; w blocks per CU
            hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, d, 16, 12345u);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, d, iters, 12345u);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, 8 * grid * 4, hipMemcpyDeviceToHost);
            std::vector<unsigned long long> v(h.begin(), h.begin() + grid * 4);
            std::sort(v.begin(), v.end());
            const double cyc = (double)v[v.size() / 2] / (iters * (double)e.per_iter);
            printf("  %dw: %5.2f (SIMD %4.2f)", w, cyc, cyc / w);
        }
        printf("\n");
    }
    return 0;
}
