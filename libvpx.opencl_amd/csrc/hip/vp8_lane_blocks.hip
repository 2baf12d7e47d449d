// Per-block test surface of the LANE-PER-ROW arithmetic (vp8_simt_prims.hip.h): the packed-byte intra predictors, the inverse
// transform on packed 16-bit pairs and the signed 8.8 loop filter streamed block row by block row -- the code
// vp8_keyframe_kernel / vp8_recon_simt_kernel / vp8_loopfilter_simt_* are made of, which the RTCD entries of
// vp8_rtcd_blocks.hip (wave-per-row primitives) do not reach.  Every lane of a wave takes ONE block / macroblock of the
// caller's arrays, exactly as a lane of the frame kernels does; the tests compare with the oracle's per-block functions, i.e. in
// the reference's terms (vp8/common/reconintra4x4.c:16, dequantize.c:29, loopfilter.c:259-299 over loopfilter_filters.c).
// A test surface: each call is a launch and a synchronisation.
#include <hip/hip_runtime.h>
#include "vp8hip.h"
#include "vp8_simt_prims.hip.h"

namespace {

// in / out: 400 bytes per macroblock = 20 rows x 20 pixels, rows and columns -4 .. 15 of a luma macroblock (its own 16 x 16 and
// the four pixels above and to the left of it).  par: 8 bytes per macroblock: mblim, blim, lim, hev_thr (struct loop_filter_info,
// loopfilter.h:51-57, as vp8_loop_filter_frame picks them), then whether the left macroblock edge, the inner edges and the top
// macroblock edge are filtered (loopfilter.c:265-280), and the filter type (0 normal, 1 simple).
__global__ void __launch_bounds__(64)
lane_lf_kernel(const unsigned char *__restrict__ in, unsigned char *__restrict__ out, const unsigned char *__restrict__ par, int n)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const bool live = i < n;
    const unsigned char *src = in + (size_t)(live ? i : 0) * 400, *pp = par + (size_t)(live ? i : 0) * 8;
    unsigned char *dst = out + (size_t)(live ? i : 0) * 400;
    v2u one = mku(1);
    asm volatile("" : "+v"(one));
    const Lim L = { mku(pp[0] << 8), mku(pp[1] << 8), mku(pp[2] << 8), mku(pp[3] << 8), one };
    const bool simple = pp[7] != 0, mbv = live && pp[4], inner = live && pp[5], mbh = live && pp[6];
    const bool any_normal = __builtin_amdgcn_ballot_w64(live && !simple) != 0, any_simple = __builtin_amdgcn_ballot_w64(live && simple) != 0;
    auto gate = [](bool b) { return lf_gate(b); };
    const Gates gv = { gate(mbv && !simple), gate(inner && !simple), gate(mbv && simple), gate(inner && simple), any_normal, any_simple };
    const Gates gh = { gate(mbh && !simple), gate(inner && !simple), gate(mbh && simple), gate(inner && simple), any_normal, any_simple };
    auto px = [&](int y, int x) { return (u32)src[(y + 4) * 20 + (x + 4)]; };
    auto dword = [&](int y, int x) { return px(y, x) | (px(y, x + 1) << 8) | (px(y, x + 2) << 16) | (px(y, x + 3) << 24); };
    auto put = [&](int y, int x, u32 v) { if (live) for (int k = 0; k < 4; k++) dst[(y + 4) * 20 + (x + 4) + k] = (unsigned char)(v >> (8 * k)); };
    if (live) for (int k = 0; k < 400; k++) dst[k] = src[k];
    u32 P[4][4];
    for (int j = 0; j < 4; j++) for (int x = 0; x < 4; x++) P[j][x] = dword(j - 4, 4 * x) ^ VP8_LF_BIAS;
#pragma unroll 1
    for (int by = 0; by < 4; by++) {
        u32 o[4][4], s[4], d[4][4];
        for (int j = 0; j < 4; j++) {
            s[j] = dword(4 * by + j, -4) ^ VP8_LF_BIAS;
            for (int x = 0; x < 4; x++) o[j][x] = dword(4 * by + j, 4 * x);
        }
        lf_block_row<4>(o, s, P, by == 0, gv, gh, L, d);
        for (int j = 0; j < 4; j++) {
            put(4 * by + j, -4, s[j] ^ VP8_LF_BIAS);
            for (int x = 0; x < 4; x++) put(4 * by - 4 + j, 4 * x, d[j][x] ^ VP8_LF_BIAS);
        }
    }
    for (int j = 0; j < 4; j++) for (int x = 0; x < 4; x++) put(12 + j, 4 * x, P[j][x] ^ VP8_LF_BIAS);
}

// mode[n]; ctx: 16 bytes per block = above[0..7], left[0..3], top_left, 3 x padding; out: 16 bytes per block, row-major
__global__ void __launch_bounds__(64)
lane_bpred_kernel(const unsigned char *__restrict__ mode, const unsigned char *__restrict__ ctx, unsigned char *__restrict__ out, int n)
{
    // the key-frame kernels' predictor (vp8_keyframe_simt.hip, luma): the selector table in LDS, the pool + three v_perm_b32 a row,
    // B_DC_PRED through the dword C, TM by arithmetic and a select
    __shared__ __attribute__((aligned(16))) u32 s_psel[PSEL_MODES * PSEL_WORDS];
    for (int k = threadIdx.x; k < PSEL_MODES * PSEL_WORDS; k += 64) s_psel[k] = k_pred_sel[k];
    __syncthreads();
    const int i = blockIdx.x * 64 + threadIdx.x;
    const bool live = i < n;
    const u32 *c = (const u32 *)(ctx + (size_t)(live ? i : 0) * 16);
    const u32 em = live ? mode[i] : 0u, a0 = c[0], a1 = c[1], left = c[2];
    const int tl = (int)(c[3] & 0xff);
    const u32 bdc = __builtin_amdgcn_sad_u8(left, 0u, __builtin_amdgcn_sad_u8(a0, 0u, 4u)) >> 3;
    u32 p[4];
    pred4x4_net((const u32x4 *)(s_psel + em * PSEL_WORDS), a0, a1, left, tl, em == 0, perm(bdc, bdc, 0u), false, 0u, p);
    const bool tm = em == VP8IR_B_TM_PRED;
    if (__builtin_amdgcn_ballot_w64(tm) != 0) {
        const v2s a01 = as_v2s(perm(a0, a0, 0x0c010c00u)), a23 = as_v2s(perm(a0, a0, 0x0c030c02u));
        for (int j = 0; j < 4; j++) {
            const u32 t = tm_row(a01, a23, (int)((left >> (8 * j)) & 0xff) - tl);
            p[j] = tm ? t : p[j];
        }
    }
    if (live) for (int j = 0; j < 4; j++) ((u32 *)(out + (size_t)i * 16))[j] = p[j];
}

// coef: 16 shorts per block in IR order (column-major, include/vp8_ir.h); dq: dc, ac per block; pred / out: 16 bytes per block
__global__ void __launch_bounds__(64)
lane_idct_add_kernel(const short *__restrict__ coef, const short *__restrict__ dq, const unsigned char *__restrict__ pred,
                     unsigned char *__restrict__ out, int n)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const u32x4 *c = (const u32x4 *)(coef + (size_t)i * 16);
    int res[16];
    dequant_idct(c[0], c[1], dq[2 * i], dq[2 * i + 1], false, 0, res);
    for (int j = 0; j < 4; j++) {
        const u32 pr = ((const u32 *)(pred + (size_t)i * 16))[j];
        const u32 r01 = ((u32)res[4 * j] & 0xffff) | ((u32)res[4 * j + 1] << 16), r23 = ((u32)res[4 * j + 2] & 0xffff) | ((u32)res[4 * j + 3] << 16);
        ((u32 *)(out + (size_t)i * 16))[j] = add_clamp_pack(pr, r01, r23);
    }
}

template <typename F>
int run(size_t in_bytes[], const void *in[], int nin, void *outp, size_t out_bytes, F launch)
{
    void *d[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    int rc = 0;
    for (int k = 0; k < nin && !rc; k++) {
        if (hipMalloc(&d[k], in_bytes[k] ? in_bytes[k] : 4) != hipSuccess || hipMemcpy(d[k], in[k], in_bytes[k], hipMemcpyHostToDevice) != hipSuccess) rc = -1;
    }
    if (!rc && hipMalloc(&d[nin], out_bytes ? out_bytes : 4) != hipSuccess) rc = -1;
    if (!rc) {
        launch(d);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess || hipMemcpy(outp, d[nin], out_bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = -1;
    }
    for (int k = 0; k <= nin; k++) if (d[k]) (void)hipFree(d[k]);
    return rc;
}

} // namespace

extern "C" int vp8hip_lane_loop_filter_mbs(const uint8_t *in, uint8_t *out, const uint8_t *par, int n)
{
    if (!in || !out || !par || n <= 0) return -2;
    size_t sz[2] = { (size_t)n * 400, (size_t)n * 8 };
    const void *src[2] = { in, par };
    return run(sz, src, 2, out, (size_t)n * 400, [&](void **d) {
        hipLaunchKernelGGL(lane_lf_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, (const unsigned char *)d[0], (unsigned char *)d[2], (const unsigned char *)d[1], n);
    });
}

extern "C" int vp8hip_lane_intra4x4(const uint8_t *mode, const uint8_t *ctx, uint8_t *out, int n)
{
    if (!mode || !ctx || !out || n <= 0) return -2;
    size_t sz[2] = { (size_t)n, (size_t)n * 16 };
    const void *src[2] = { mode, ctx };
    return run(sz, src, 2, out, (size_t)n * 16, [&](void **d) {
        hipLaunchKernelGGL(lane_bpred_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, (const unsigned char *)d[0], (const unsigned char *)d[1], (unsigned char *)d[2], n);
    });
}

extern "C" int vp8hip_lane_dequant_idct_add(const int16_t *coef, const int16_t *dq, const uint8_t *pred, uint8_t *out, int n)
{
    if (!coef || !dq || !pred || !out || n <= 0) return -2;
    size_t sz[3] = { (size_t)n * 32, (size_t)n * 4, (size_t)n * 16 };
    const void *src[3] = { coef, dq, pred };
    return run(sz, src, 3, out, (size_t)n * 16, [&](void **d) {
        hipLaunchKernelGGL(lane_idct_add_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, (const short *)d[0], (const short *)d[1], (const unsigned char *)d[2], (unsigned char *)d[3], n);
    });
}
