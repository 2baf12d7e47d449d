// Dev aid (GPU): what a bool-decoder decision costs a lone wave, piece by piece -- the floor under vp8_entropy_kernel.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bool_ubench tools/bool_ubench.hip && /tmp/bool_ubench
// Every lane decodes its own stream of random bytes; shader clock (s_memtime: 100 MHz; wall time beside it) per decision for
//   0  the decision alone, fixed probability, bits summed (no branch on the result)
//   1  + the probability of the next decision picked by the result (select, no branch)
//   2  + a divergent branch on the result (two small bodies)
//   3  + the next probability read from LDS at an address the result decides (a row of 12 bytes, three words)
//   4  a binary tree walk like the token tree's: depth <= 4, branch per level, row read per token
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <chrono>

typedef unsigned int u32;
struct BD { u32 value; int bits; u32 range; u32 pos; u32 n01, n2; };

__device__ __forceinline__ void request3(BD &b, const uint8_t *data, u32 pos)
{
    typedef unsigned short __attribute__((aligned(1), may_alias)) u16u;
    const uint8_t *p = data + pos;
    b.n01 = *(const u16u *)p; b.n2 = p[2];
}
__device__ __forceinline__ int bd_get(BD &b, const uint8_t *data, u32 prob)
{
    const u32 split = 1u + (__umul24(b.range - 1u, prob) >> 8);
    if (b.bits < 0) {
        const u32 nxt = (b.n01 & 255u) << 16 | (b.n01 & 0xff00u) | b.n2;
        b.value |= nxt << (-b.bits);
        b.bits += 24;
        request3(b, data, b.pos);
        b.pos += 3;
    }
    const u32 big = split << 24;
    const bool bit = b.value >= big;
    b.value -= bit ? big : 0u;
    b.range = bit ? b.range - split : split;
    const int shift = __builtin_clz(b.range) - 24;
    b.range <<= shift; b.value <<= shift; b.bits -= shift;
    return bit ? 1 : 0;
}

template <int MODE>
__global__ void __launch_bounds__(64) k(const uint8_t *data, size_t per_lane, int n, u32 *out, unsigned long long *clk, int active)
{
    __shared__ u32 rows[64 * 3 * 16 + 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 3 * 16; i += 64) rows[i] = 0x80604020u + 0x01030507u * (u32)i;
    __syncthreads();
    if (lane >= active) return;
    const uint8_t *d = data + per_lane * (blockIdx.x * 64 + lane);
    BD b; b.value = 0; b.bits = -8; b.range = 255; request3(b, d, 0); b.pos = 3;
    u32 acc = 0, prob = 128;
    const u32 *myrows = rows + lane;                      // (odd strides below: lanes on different banks)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) for (int i = 0; i < n; i++) acc += bd_get(b, d, 140);
    if (MODE == 5) { b.bits = 1 << 30; for (int i = 0; i < n; i++) { acc += bd_get(b, d, 140); b.value ^= acc << 9; } }   // no refills
    if (MODE == 1) for (int i = 0; i < n; i++) { const int bit = bd_get(b, d, prob); acc += bit; prob = bit ? 97 + (acc & 63) : 180 - (acc & 31); }
    if (MODE == 2) for (int i = 0; i < n; i++) {
        const int bit = bd_get(b, d, prob);
        if (bit) { acc = acc * 3 + 1; prob = 90 + (acc & 63); } else { acc ^= acc >> 3; prob = 200 - (acc & 63); }
    }
    if (MODE == 3) {
        u32 w0 = myrows[0], w1 = myrows[65], w2 = myrows[130];
        for (int i = 0; i < n; i++) {
            const int bit = bd_get(b, d, (w0 >> 8) & 255u | 1u);
            acc = acc * 2 + bit;
            const u32 *r = myrows + 195 * ((acc & 7) + bit);
            w0 = r[0]; w1 = r[65]; w2 = r[130];
            acc += w1 ^ w2;
        }
    }
    if (MODE == 4) {
        for (int i = 0; i < n; ) {
            const u32 *r = myrows + 195 * (acc & 15);
            const u32 w0 = r[0], w1 = r[65], w2 = r[130];
            int v;
            i++;
            if (!bd_get(b, d, w0 & 255u | 1u)) v = 0;
            else { i++; if (!bd_get(b, d, (w0 >> 8) & 255u | 1u)) v = 1;
            else { i++; if (!bd_get(b, d, (w0 >> 16) & 255u | 1u)) { i++; v = 2 + bd_get(b, d, (w1 >> 8) & 255u | 1u); }
            else { i++; if (!bd_get(b, d, w2 & 255u | 1u)) v = 5; else { i++; v = 7 + bd_get(b, d, (w2 >> 8) & 255u | 1u); } } } }
            i++;
            if (bd_get(b, d, 128)) v = -v;
            acc = acc * 5 + (u32)v;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = acc + b.value;
    if (lane == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE> static void run(const uint8_t *d_data, size_t per_lane, int n, u32 *d_out, unsigned long long *d_clk, int waves, int active = 64)
{
    for (int rep = 0; rep < 2; rep++) {
        auto w0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, d_data, per_lane, n, d_out, d_clk, active);
        hipDeviceSynchronize();
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        unsigned long long clk;
        hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost);
        if (rep) printf("mode %d, %d wave(s), %d lanes: %.1f ns per decision (%.1f s_memtime ticks)\n", MODE, waves, active, wall * 1e9 / n, (double)clk / n);
    }
}

int main()
{
    const int n = 2000000;
    const size_t per_lane = (size_t)n / 2 + 4096;
    const int maxw = 8;
    std::vector<uint8_t> h(per_lane * 64 * maxw);
    srand(7);
    for (auto &x : h) x = (uint8_t)rand();
    uint8_t *d_data; u32 *d_out; unsigned long long *d_clk;
    hipMalloc(&d_data, h.size()); hipMalloc(&d_out, 64 * maxw * 4); hipMalloc(&d_clk, 8 * maxw);
    hipMemcpy(d_data, h.data(), h.size(), hipMemcpyHostToDevice);
    run<0>(d_data, per_lane, n, d_out, d_clk, 1);
    run<1>(d_data, per_lane, n, d_out, d_clk, 1);
    run<2>(d_data, per_lane, n, d_out, d_clk, 1);
    run<3>(d_data, per_lane, n, d_out, d_clk, 1);
    run<4>(d_data, per_lane, n, d_out, d_clk, 1);
    run<4>(d_data, per_lane, n, d_out, d_clk, 8);
    run<5>(d_data, per_lane, n, d_out, d_clk, 1);
    run<0>(d_data, per_lane, n, d_out, d_clk, 1, 1);
    run<0>(d_data, per_lane, n, d_out, d_clk, 1, 4);
    run<4>(d_data, per_lane, n, d_out, d_clk, 1, 1);
    run<4>(d_data, per_lane, n, d_out, d_clk, 1, 4);
    return 0;
}
