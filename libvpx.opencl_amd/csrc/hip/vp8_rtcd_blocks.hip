// The reference's 32 per-block RTCD decoder entries (vp8/common/rtcd_defs.sh:20-204), GPU-backed: part 2 of
// include/vp8_rtcd.h.  Each call stages the rectangles of the caller's planes that the entry reads through a pinned,
// device-visible buffer, launches ONE wavefront that applies the block arithmetic of vp8_block_prims.hip.h (the same
// functions the wave-per-row frame kernels are built from), waits, and writes the rectangles the entry modifies back.
// Lanes are the natural unit of each entry: a pixel line across an edge for the loop filters, a 4x4 block for the
// block drivers, a 4-pixel row segment for the predictors.
//
// This is a conformance / bring-up surface in the reference's own terms, not a fast path (a launch and a
// synchronisation per block); the decoder itself only uses part 1 of the table.
#include "vp8_block_prims.hip.h"
#include "vp8_rtcd.h"

#include <hip/hip_runtime.h>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int WS = 32;                  // row pitch of a staged plane window
constexpr int WROWS = 32;
enum Op {
    OP_DEQUANTIZE_B, OP_DEQUANT_IDCT_ADD, OP_Y_BLOCK, OP_UV_BLOCK, OP_IDCT_ADD, OP_WALSH, OP_WALSH_1, OP_DC_ONLY,
    OP_LF_MBV, OP_LF_BV, OP_LF_MBH, OP_LF_BH, OP_LFS_MBV, OP_LFS_BV, OP_LFS_MBH, OP_LFS_BH,
    OP_COPY, OP_INTRA_MB, OP_INTRA_4X4, OP_SIXTAP, OP_BILINEAR
};

// Everything a call exchanges with the device.  win[k] holds a window of one of the caller's planes; org[k] is where
// the caller's pointer sits inside it, so that device code addresses pixels relative to that pointer with pitch WS.
struct Stage {
    int org[4];
    int arg[8];
    unsigned char lim[4];               // mblim, blim, lim, hev_thr
    signed char eobs[28];
    short q[400];
    short dq[16];
    short out16[256];
    unsigned char win[4][WROWS * WS];
};

// ---------------------------------------------------------------------------------------------- device side
__device__ __forceinline__ void idct_add_block(const short *in, const unsigned char *pred, int ps, unsigned char *dst, int ds)
{
    int t[4][4];                        // t[c] = column c after the vertical pass, truncated like the reference's short
#pragma unroll
    for (int c = 0; c < 4; c++) {
        int o[4];
        idct_col(in[c], in[4 + c], in[8 + c], in[12 + c], o);
#pragma unroll
        for (int r = 0; r < 4; r++) t[c][r] = (short)o[r];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row[4] = { t[0][r], t[1][r], t[2][r], t[3][r] };
        int o[4];
        idct_row(row, o);
#pragma unroll
        for (int c = 0; c < 4; c++) dst[r * ds + c] = (unsigned char)clamp255(o[c] + pred[r * ps + c]);
    }
}

// one 4x4 block of a block driver (idct_blk.c:20-86): full transform when eob > 1, else the DC shortcut
__device__ __forceinline__ void driver_block(short *q, const short *dq, unsigned char *dst, int eob)
{
    if (eob > 1) {
        short d[16];
#pragma unroll
        for (int i = 0; i < 16; i++) { d[i] = (short)(q[i] * dq[i]); q[i] = 0; }
        idct_add_block(d, dst, WS, dst, WS);
    } else {
        const int a1 = ((short)(q[0] * dq[0]) + 4) >> 3;
        for (int i = 0; i < 16; i++) dst[(i >> 2) * WS + (i & 3)] = (unsigned char)clamp255(a1 + dst[(i >> 2) * WS + (i & 3)]);
        q[0] = 0; q[1] = 0;
    }
}

// One pixel line across an edge: q0 points at the first pixel after the edge, `step` is the distance between
// neighbours across it.  kind: 0 inner, 1 macroblock edge, 2 simple (filter_edge in vp8_block_prims.hip.h).
__device__ __forceinline__ void edge_line(unsigned char *q0, int step, int kind, const LfParams &lp, int edge_limit)
{
    int a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = q0[(i - 4) * step];
    filter_edge(a + 4, kind, lp, edge_limit);
#pragma unroll
    for (int i = 1; i < 7; i++) q0[(i - 4) * step] = (unsigned char)a[i];
}

// whole-block intra prediction of an n x n plane (reconintra.c): 4-pixel row segments, intra_pred4 per segment
__device__ __forceinline__ void intra_plane(const unsigned char *p, unsigned char *out, int n, int mode, int up, int left, int lane)
{
    const unsigned char *above = p - WS;
    int dc = 128;
    if (mode == VP8IR_DC_PRED && (up || left)) {
        int sum = 0;
        const int shift = (n == 16 ? 3 : 2) + up + left;
        if (up) for (int i = 0; i < n; i++) sum += above[i];
        if (left) for (int i = 0; i < n; i++) sum += p[i * WS - 1];
        dc = (sum + (1 << (shift - 1))) >> shift;
    }
    const int tl = above[-1];
    for (int s = lane; s < n * n / 4; s += 64) {
        const int r = s / (n / 4), c = (s % (n / 4)) * 4;
        const u32 ab = (u32)above[c] | ((u32)above[c + 1] << 8) | ((u32)above[c + 2] << 16) | ((u32)above[c + 3] << 24);
        const u32 v = intra_pred4(mode, ab, p[r * WS - 1], tl, dc);
        for (int i = 0; i < 4; i++) out[r * WS + c + i] = (unsigned char)(v >> (8 * i));
    }
}

__global__ __launch_bounds__(64) void vp8_block_op_kernel(Stage *s, int op)
{
    __shared__ int tmp[21 * 16];
    const int lane = threadIdx.x;
    unsigned char *const P0 = s->win[0] + s->org[0], *const P1 = s->win[1] + s->org[1];
    unsigned char *const P2 = s->win[2] + s->org[2], *const P3 = s->win[3] + s->org[3];
    const int *arg = s->arg;
    const LfParams lp = { s->lim[0], s->lim[1], s->lim[2], s->lim[3] };
    switch (op) {
    case OP_DEQUANTIZE_B:               // dequantize.c:17-27
        if (lane < 16) s->out16[lane] = (short)(s->q[lane] * s->dq[lane]);
        break;
    case OP_DEQUANT_IDCT_ADD:           // dequantize.c:29-44
        if (lane == 0) driver_block(s->q, s->dq, P0, 16);
        break;
    case OP_Y_BLOCK:                    // idct_blk.c:20-44
        if (lane < 16) driver_block(s->q + 16 * lane, s->dq, P0 + (lane >> 2) * 4 * WS + (lane & 3) * 4, s->eobs[lane]);
        break;
    case OP_UV_BLOCK:                   // idct_blk.c:46-86
        if (lane < 8) {
            unsigned char *d = (lane < 4 ? P0 : P1) + ((lane >> 1) & 1) * 4 * WS + (lane & 1) * 4;
            driver_block(s->q + 16 * lane, s->dq, d, s->eobs[lane]);
        }
        break;
    case OP_IDCT_ADD:                   // idctllm.c:28-110: input is already dequantised, pred and dst are separate
        if (lane == 0) idct_add_block(s->q, P0, WS, P3, WS);
        break;
    case OP_WALSH:                      // idctllm.c:140-192: lane c = column c, then lane r = row r
        if (lane < 4) {
            const short *in = s->q;
            const int a = in[lane] + in[12 + lane], b = in[4 + lane] + in[8 + lane];
            const int c = in[4 + lane] - in[8 + lane], d = in[lane] - in[12 + lane];
            tmp[lane] = (short)(a + b); tmp[4 + lane] = (short)(c + d);
            tmp[8 + lane] = (short)(a - b); tmp[12 + lane] = (short)(d - c);
        }
        __syncthreads();
        if (lane < 4) {
            const int *t = tmp + 4 * lane;
            const int a = t[0] + t[3], b = t[1] + t[2], c = t[1] - t[2], d = t[0] - t[3];
            s->out16[(4 * lane + 0) * 16] = (short)((a + b + 3) >> 3);
            s->out16[(4 * lane + 1) * 16] = (short)((c + d + 3) >> 3);
            s->out16[(4 * lane + 2) * 16] = (short)((a - b + 3) >> 3);
            s->out16[(4 * lane + 3) * 16] = (short)((d - c + 3) >> 3);
        }
        break;
    case OP_WALSH_1:                    // idctllm.c:194-204
        if (lane < 16) s->out16[lane * 16] = (short)((s->q[0] + 3) >> 3);
        break;
    case OP_DC_ONLY:                    // idctllm.c:112-138: lane = pixel
        if (lane < 16) {
            const int a1 = ((short)arg[0] + 4) >> 3, r = lane >> 2, c = lane & 3;
            P3[r * WS + c] = (unsigned char)clamp255(a1 + P0[r * WS + c]);
        }
        break;
    // loop filters (loopfilter_filters.c:316-430): lanes 0..15 = the 16 luma lines across the edge, 16..23 / 24..31 =
    // the 8 lines of U / V (arg[0] = chroma planes present).  Inner edges of one line are filtered in order by its lane.
    case OP_LF_MBV: case OP_LF_MBH: case OP_LF_BV: case OP_LF_BH: {
        const bool vertical = op == OP_LF_MBV || op == OP_LF_BV, mb = op == OP_LF_MBV || op == OP_LF_MBH;
        unsigned char *base = lane < 16 ? P0 : (lane < 24 ? P1 : P2);
        const int line = lane < 16 ? lane : (lane - 16) & 7;
        if (lane >= 32 || (lane >= 16 && !arg[0])) break;
        unsigned char *l0 = base + (vertical ? line * WS : line);
        const int step = vertical ? 1 : WS;
        if (mb) edge_line(l0, step, 1, lp, lp.mblim);
        else
            for (int e = 4; e < (lane < 16 ? 16 : 8); e += 4) edge_line(l0 + e * step, step, 0, lp, lp.blim);
        break;
    }
    case OP_LFS_MBV: case OP_LFS_MBH: case OP_LFS_BV: case OP_LFS_BH: {      // loopfilter_filters.c:316-357,432-end
        const bool vertical = op == OP_LFS_MBV || op == OP_LFS_BV, mb = op == OP_LFS_MBV || op == OP_LFS_MBH;
        if (lane >= 16) break;
        unsigned char *l0 = P0 + (vertical ? lane * WS : lane);
        const int step = vertical ? 1 : WS;
        if (mb) edge_line(l0, step, 2, lp, lp.blim);
        else
            for (int e = 4; e < 16; e += 4) edge_line(l0 + e * step, step, 2, lp, lp.blim);
        break;
    }
    case OP_COPY:                       // reconinter.c:22-130: arg = w, h
        for (int i = lane; i < arg[0] * arg[1]; i += 64) P3[(i / arg[0]) * WS + i % arg[0]] = P0[(i / arg[0]) * WS + i % arg[0]];
        break;
    case OP_INTRA_MB:                   // arg = n, mode, up, left, in_place, two planes
        intra_plane(P0, arg[4] ? P0 : P3, arg[0], arg[1], arg[2], arg[3], lane);
        if (arg[5]) intra_plane(P1, arg[4] ? P1 : P2, arg[0], arg[1], arg[2], arg[3], lane);
        break;
    case OP_INTRA_4X4:                  // reconintra4x4.c:16-303: lane = pixel; edge vector P[] as in k_bpred_tab's comment
        if (lane < 16) {
            const unsigned char *above = P0 - WS;
            const int mode = arg[0], r = lane >> 2, c = lane & 3, tl = above[-1];
            int v;
            if (mode == VP8IR_B_DC_PRED) {
                v = 4;
                for (int i = 0; i < 4; i++) v += above[i] + P0[i * WS - 1];
                v >>= 3;
            } else if (mode == VP8IR_B_TM_PRED)
                v = clamp255(above[c] - tl + P0[r * WS - 1]);
            else {
                auto edge = [&](int k) -> int {
                    return k == 0 ? P0[3 * WS - 1] : k < 5 ? P0[(4 - k) * WS - 1] : k == 5 ? tl : above[k < 14 ? k - 6 : 7];
                };
                const int e = k_bpred_tab[mode * 16 + lane], k = e & 15;
                if ((e >> 4) == 2) v = (edge(k - 1) + 2 * edge(k) + edge(k + 1) + 2) >> 2;
                else if ((e >> 4) == 1) v = (edge(k) + edge(k + 1) + 1) >> 1;
                else v = edge(k);
            }
            P3[r * WS + c] = (unsigned char)v;
        }
        break;
    case OP_SIXTAP: {                   // filter.c:41-128,186-278: both passes always; arg = w, h, xoffset, yoffset
        const int w = arg[0], h = arg[1];
        const SixTaps tx = sixtap_taps(arg[2]), ty = sixtap_taps(arg[3]);
        u32 *const H = (u32 *)tmp;     // first pass: (h + 5) rows of w / 4 dwords
        for (int i = lane; i < (h + 5) * (w / 4); i += 64) {
            const int r = i / (w / 4), c = (i % (w / 4)) * 4;
            H[i] = sixtap_hrow((g_cu8p)(P0 + (r - 2) * WS + c - 2), tx);
        }
        __syncthreads();
        for (int i = lane; i < h * (w / 4); i += 64) {
            const int r = i / (w / 4), cw = i % (w / 4);
            u32 col[6];
            for (int k = 0; k < 6; k++) col[k] = H[(r + k) * (w / 4) + cw];
            const u32 v = sixtap_vcol(col, ty);
            for (int b = 0; b < 4; b++) P3[r * WS + cw * 4 + b] = (unsigned char)(v >> (8 * b));
        }
        break;
    }
    case OP_BILINEAR: {                 // filter.c:280-494: first pass (h + 1) rows unclamped 16-bit, then vertical
        const int w = arg[0], h = arg[1];
        const int h0 = 128 - arg[2] * 16, h1 = arg[2] * 16, v0 = 128 - arg[3] * 16, v1 = arg[3] * 16;
        for (int i = lane; i < (h + 1) * w; i += 64) {
            const int r = i / w, c = i % w;
            tmp[i] = (P0[r * WS + c] * h0 + P0[r * WS + c + 1] * h1 + 64) >> 7;
        }
        __syncthreads();
        for (int i = lane; i < h * w; i += 64) P3[(i / w) * WS + i % w] = (unsigned char)((tmp[i] * v0 + tmp[i + w] * v1 + 64) >> 7);
        break;
    }
    }
}

// ---------------------------------------------------------------------------------------------- host side
struct Engine {
    std::mutex mu;
    int device = 0;
    Stage *st = nullptr;
    hipStream_t stream = nullptr;
};
Engine g_eng;

[[noreturn]] void die(const char *what, hipError_t e)
{
    fprintf(stderr, "vp8_rtcd (hip): %s: %s -- the per-block entries have no CPU fallback\n", what, hipGetErrorString(e));
    abort();
}

struct Rect { int x0, y0, w, h; };

// One call: constructed with the engine locked and the stage cleared, run() launches and waits.
struct Call {
    std::lock_guard<std::mutex> lock;
    Stage *s;
    Rect bound[4];
    Call() : lock(g_eng.mu)
    {
        hipError_t e = hipSetDevice(g_eng.device);
        if (e != hipSuccess) die("hipSetDevice", e);
        if (!g_eng.st) {
            if ((e = hipHostMalloc((void **)&g_eng.st, sizeof(Stage), hipHostMallocDefault)) != hipSuccess) die("hipHostMalloc", e);
            if ((e = hipStreamCreateWithFlags(&g_eng.stream, hipStreamNonBlocking)) != hipSuccess) die("hipStreamCreate", e);
        }
        s = g_eng.st;
        memset(s->org, 0, sizeof s->org);
        memset(s->arg, 0, sizeof s->arg);
    }
    // window k covers `b` (coordinates relative to the caller's pointer)
    void window(int k, Rect b)
    {
        if (b.w > WS || b.h > WROWS) die("window too large", hipErrorInvalidValue);
        bound[k] = b;
        s->org[k] = -b.y0 * WS - b.x0;
    }
    void put(int k, const unsigned char *p, int stride, Rect r)
    {
        for (int y = r.y0; y < r.y0 + r.h; y++)
            memcpy(s->win[k] + s->org[k] + y * WS + r.x0, p + (long)y * stride + r.x0, (size_t)r.w);
    }
    void get(int k, unsigned char *p, int stride, Rect r)
    {
        for (int y = r.y0; y < r.y0 + r.h; y++)
            memcpy(p + (long)y * stride + r.x0, s->win[k] + s->org[k] + y * WS + r.x0, (size_t)r.w);
    }
    void in(int k, const unsigned char *p, int stride, Rect r) { window(k, r); put(k, p, stride, r); }
    void run(int op)
    {
        hipLaunchKernelGGL(vp8_block_op_kernel, dim3(1), dim3(64), 0, g_eng.stream, s, op);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) die("kernel launch", e);
        if ((e = hipStreamSynchronize(g_eng.stream)) != hipSuccess) die("hipStreamSynchronize", e);
    }
};

void loop_filter(int op, unsigned char *y, unsigned char *u, unsigned char *v, int ys, int uvs, const struct loop_filter_info *lfi)
{
    Call c;
    const bool vertical = op == OP_LF_MBV || op == OP_LF_BV, mb = op == OP_LF_MBV || op == OP_LF_MBH;
    // macroblock edges touch 4 pixels either side of the edge at 0; inner edges stay inside the macroblock
    const Rect ry = mb ? (vertical ? Rect{ -4, 0, 8, 16 } : Rect{ 0, -4, 16, 8 }) : Rect{ 0, 0, 16, 16 };
    const Rect rc = mb ? (vertical ? Rect{ -4, 0, 8, 8 } : Rect{ 0, -4, 8, 8 }) : Rect{ 0, 0, 8, 8 };
    c.s->lim[0] = lfi->mblim[0]; c.s->lim[1] = lfi->blim[0]; c.s->lim[2] = lfi->lim[0]; c.s->lim[3] = lfi->hev_thr[0];
    c.in(0, y, ys, ry);
    c.s->arg[0] = u != nullptr;         // the reference filters chroma `if (u_ptr)` / `if (v_ptr)`
    if (u) c.in(1, u, uvs, rc);
    if (v) c.in(2, v, uvs, rc); else if (u) c.in(2, u, uvs, rc);
    c.run(op);
    c.get(0, y, ys, ry);
    if (u) c.get(1, u, uvs, rc);
    if (v) c.get(2, v, uvs, rc);
}

void loop_filter_simple(int op, unsigned char *y, int ys, const unsigned char *blimit)
{
    Call c;
    const bool vertical = op == OP_LFS_MBV || op == OP_LFS_BV, mb = op == OP_LFS_MBV || op == OP_LFS_MBH;
    const Rect ry = mb ? (vertical ? Rect{ -2, 0, 4, 16 } : Rect{ 0, -2, 16, 4 }) : Rect{ 0, 0, 16, 16 };
    c.s->lim[1] = blimit[0];
    // the device reads 4 either side (filter_edge); the simple filter only uses, and only these rectangles carry, 2
    c.window(0, mb ? (vertical ? Rect{ -4, 0, 8, 16 } : Rect{ 0, -4, 16, 8 }) : Rect{ 0, 0, 16, 16 });
    c.put(0, y, ys, ry);
    c.run(op);
    c.get(0, y, ys, ry);
}

void predict(int op, const unsigned char *src, int sp, int xo, int yo, unsigned char *dst, int dp, int w, int h)
{
    Call c;
    c.s->arg[0] = w; c.s->arg[1] = h; c.s->arg[2] = xo; c.s->arg[3] = yo;
    // six-tap: rows -2 .. h+2, columns -2 .. w+2 (filter.c:186-278); bilinear: one extra row and column (:376-397)
    if (op == OP_SIXTAP) c.in(0, src, sp, Rect{ -2, -2, w + 5, h + 5 });
    else c.in(0, src, sp, Rect{ 0, 0, w + 1, h + 1 });
    c.window(3, Rect{ 0, 0, w, h });
    c.run(op);
    c.get(3, dst, dp, Rect{ 0, 0, w, h });
}

void copy_mem(const unsigned char *src, int sp, unsigned char *dst, int dp, int w, int h)
{
    Call c;
    c.s->arg[0] = w; c.s->arg[1] = h;
    c.in(0, src, sp, Rect{ 0, 0, w, h });
    c.window(3, Rect{ 0, 0, w, h });
    c.run(OP_COPY);
    c.get(3, dst, dp, Rect{ 0, 0, w, h });
}

// the row above (with the top-left pixel) and the column to the left of an n x n block
void put_intra_edges(Call &c, int k, const unsigned char *p, int stride, int n)
{
    c.window(k, Rect{ -1, -1, n + 1, n + 1 });
    c.put(k, p, stride, Rect{ -1, -1, n + 1, 1 });
    c.put(k, p, stride, Rect{ -1, 0, 1, n });
}

}  // namespace

extern "C" {

int vp8_rtcd_blocks_set_device(int device)
{
    std::lock_guard<std::mutex> lock(g_eng.mu);
    if (g_eng.st && g_eng.device != device) return -1;
    g_eng.device = device;
    return 0;
}

void vp8_dequantize_b_hip(struct blockd *d, short *dqc)
{
    Call c;
    memcpy(c.s->q, d->qcoeff_base + d->qcoeff_offset, 32);
    memcpy(c.s->dq, dqc, 32);
    c.run(OP_DEQUANTIZE_B);
    memcpy(d->dqcoeff_base + d->dqcoeff_offset, c.s->out16, 32);
}

void vp8_dequant_idct_add_hip(short *input, short *dq, unsigned char *output, int stride)
{
    Call c;
    memcpy(c.s->q, input, 32);
    memcpy(c.s->dq, dq, 32);
    c.in(0, output, stride, Rect{ 0, 0, 4, 4 });
    c.run(OP_DEQUANT_IDCT_ADD);
    c.get(0, output, stride, Rect{ 0, 0, 4, 4 });
    memcpy(input, c.s->q, 32);
}

void vp8_dequant_idct_add_y_block_hip(short *q, short *dq, unsigned char *dst, int stride, char *eobs)
{
    Call c;
    memcpy(c.s->q, q, 512);
    memcpy(c.s->dq, dq, 32);
    memcpy(c.s->eobs, eobs, 16);
    c.in(0, dst, stride, Rect{ 0, 0, 16, 16 });
    c.run(OP_Y_BLOCK);
    c.get(0, dst, stride, Rect{ 0, 0, 16, 16 });
    memcpy(q, c.s->q, 512);
}

void vp8_dequant_idct_add_uv_block_hip(short *q, short *dq, unsigned char *dst_u, unsigned char *dst_v, int stride, char *eobs)
{
    Call c;
    memcpy(c.s->q, q, 256);
    memcpy(c.s->dq, dq, 32);
    memcpy(c.s->eobs, eobs, 8);
    c.in(0, dst_u, stride, Rect{ 0, 0, 8, 8 });
    c.in(1, dst_v, stride, Rect{ 0, 0, 8, 8 });
    c.run(OP_UV_BLOCK);
    c.get(0, dst_u, stride, Rect{ 0, 0, 8, 8 });
    c.get(1, dst_v, stride, Rect{ 0, 0, 8, 8 });
    memcpy(q, c.s->q, 256);
}

void vp8_loop_filter_mbv_hip(unsigned char *y, unsigned char *u, unsigned char *v, int ys, int uvs, struct loop_filter_info *lfi)
{ loop_filter(OP_LF_MBV, y, u, v, ys, uvs, lfi); }
void vp8_loop_filter_bv_hip(unsigned char *y, unsigned char *u, unsigned char *v, int ys, int uvs, struct loop_filter_info *lfi)
{ loop_filter(OP_LF_BV, y, u, v, ys, uvs, lfi); }
void vp8_loop_filter_mbh_hip(unsigned char *y, unsigned char *u, unsigned char *v, int ys, int uvs, struct loop_filter_info *lfi)
{ loop_filter(OP_LF_MBH, y, u, v, ys, uvs, lfi); }
void vp8_loop_filter_bh_hip(unsigned char *y, unsigned char *u, unsigned char *v, int ys, int uvs, struct loop_filter_info *lfi)
{ loop_filter(OP_LF_BH, y, u, v, ys, uvs, lfi); }
void vp8_loop_filter_simple_mbv_hip(unsigned char *y, int ys, const unsigned char *blimit) { loop_filter_simple(OP_LFS_MBV, y, ys, blimit); }
void vp8_loop_filter_simple_mbh_hip(unsigned char *y, int ys, const unsigned char *blimit) { loop_filter_simple(OP_LFS_MBH, y, ys, blimit); }
void vp8_loop_filter_simple_bv_hip(unsigned char *y, int ys, const unsigned char *blimit) { loop_filter_simple(OP_LFS_BV, y, ys, blimit); }
void vp8_loop_filter_simple_bh_hip(unsigned char *y, int ys, const unsigned char *blimit) { loop_filter_simple(OP_LFS_BH, y, ys, blimit); }

void vp8_short_idct4x4llm_hip(short *input, unsigned char *pred, int pitch, unsigned char *dst, int dst_stride)
{
    Call c;
    memcpy(c.s->q, input, 32);
    c.in(0, pred, pitch, Rect{ 0, 0, 4, 4 });
    c.window(3, Rect{ 0, 0, 4, 4 });
    c.run(OP_IDCT_ADD);
    c.get(3, dst, dst_stride, Rect{ 0, 0, 4, 4 });
}

static void walsh(int op, short *input, short *output)
{
    Call c;
    memcpy(c.s->q, input, 32);
    c.run(op);
    for (int i = 0; i < 16; i++) output[i * 16] = c.s->out16[i * 16];     // the DC slot of each of the 16 luma blocks
}
void vp8_short_inv_walsh4x4_hip(short *input, short *output) { walsh(OP_WALSH, input, output); }
void vp8_short_inv_walsh4x4_1_hip(short *input, short *output) { walsh(OP_WALSH_1, input, output); }

void vp8_dc_only_idct_add_hip(short input, unsigned char *pred, int pred_stride, unsigned char *dst, int dst_stride)
{
    Call c;
    c.s->arg[0] = input;
    c.in(0, pred, pred_stride, Rect{ 0, 0, 4, 4 });
    c.window(3, Rect{ 0, 0, 4, 4 });
    c.run(OP_DC_ONLY);
    c.get(3, dst, dst_stride, Rect{ 0, 0, 4, 4 });
}

void vp8_copy_mem16x16_hip(unsigned char *src, int sp, unsigned char *dst, int dp) { copy_mem(src, sp, dst, dp, 16, 16); }
void vp8_copy_mem8x8_hip(unsigned char *src, int sp, unsigned char *dst, int dp) { copy_mem(src, sp, dst, dp, 8, 8); }
void vp8_copy_mem8x4_hip(unsigned char *src, int sp, unsigned char *dst, int dp) { copy_mem(src, sp, dst, dp, 8, 4); }

void vp8_build_intra_predictors_mby_px_hip(const unsigned char *y, int y_stride, int mode, int up, int left, unsigned char *ypred)
{
    if (mode > VP8IR_TM_PRED) return;                // reconintra.c:126-136: B_PRED and the inter modes do nothing
    Call c;
    const int a[6] = { 16, mode, up, left, 0, 0 };
    memcpy(c.s->arg, a, sizeof a);
    put_intra_edges(c, 0, y, y_stride, 16);
    c.window(3, Rect{ 0, 0, 16, 16 });
    c.run(OP_INTRA_MB);
    c.get(3, ypred, 16, Rect{ 0, 0, 16, 16 });
}

void vp8_build_intra_predictors_mby_s_px_hip(unsigned char *y, int y_stride, int mode, int up, int left)
{
    if (mode > VP8IR_TM_PRED) return;
    Call c;
    const int a[6] = { 16, mode, up, left, 1, 0 };
    memcpy(c.s->arg, a, sizeof a);
    put_intra_edges(c, 0, y, y_stride, 16);
    c.run(OP_INTRA_MB);
    c.get(0, y, y_stride, Rect{ 0, 0, 16, 16 });
}

void vp8_build_intra_predictors_mbuv_px_hip(const unsigned char *u, const unsigned char *v, int uv_stride, int uv_mode, int up,
                                            int left, unsigned char *upred, unsigned char *vpred)
{
    if (uv_mode > VP8IR_TM_PRED) return;
    Call c;
    const int a[6] = { 8, uv_mode, up, left, 0, 1 };
    memcpy(c.s->arg, a, sizeof a);
    put_intra_edges(c, 0, u, uv_stride, 8);
    put_intra_edges(c, 1, v, uv_stride, 8);
    c.window(3, Rect{ 0, 0, 8, 8 });
    c.window(2, Rect{ 0, 0, 8, 8 });
    c.run(OP_INTRA_MB);
    c.get(3, upred, 8, Rect{ 0, 0, 8, 8 });
    c.get(2, vpred, 8, Rect{ 0, 0, 8, 8 });
}

void vp8_build_intra_predictors_mbuv_s_px_hip(unsigned char *u, unsigned char *v, int uv_stride, int uv_mode, int up, int left)
{
    if (uv_mode > VP8IR_TM_PRED) return;
    Call c;
    const int a[6] = { 8, uv_mode, up, left, 1, 1 };
    memcpy(c.s->arg, a, sizeof a);
    put_intra_edges(c, 0, u, uv_stride, 8);
    put_intra_edges(c, 1, v, uv_stride, 8);
    c.run(OP_INTRA_MB);
    c.get(0, u, uv_stride, Rect{ 0, 0, 8, 8 });
    c.get(1, v, uv_stride, Rect{ 0, 0, 8, 8 });
}

void vp8_intra4x4_predict_hip(unsigned char *src, int src_stride, int b_mode, unsigned char *dst, int dst_stride)
{
    Call c;
    c.s->arg[0] = b_mode;
    c.window(0, Rect{ -1, -1, 9, 5 });
    c.put(0, src, src_stride, Rect{ -1, -1, 9, 1 });      // top-left + the 8 pixels above (above-right included)
    c.put(0, src, src_stride, Rect{ -1, 0, 1, 4 });       // left column
    c.window(3, Rect{ 0, 0, 4, 4 });
    c.run(OP_INTRA_4X4);
    c.get(3, dst, dst_stride, Rect{ 0, 0, 4, 4 });
}

#define VP8_PREDICT(kind, OP, W, H)                                                                                        \
    void vp8_##kind##_predict##W##x##H##_hip(unsigned char *src, int sp, int xo, int yo, unsigned char *dst, int dp)       \
    { predict(OP, src, sp, xo, yo, dst, dp, W, H); }
VP8_PREDICT(sixtap, OP_SIXTAP, 16, 16)
VP8_PREDICT(sixtap, OP_SIXTAP, 8, 8)
VP8_PREDICT(sixtap, OP_SIXTAP, 8, 4)
VP8_PREDICT(sixtap, OP_SIXTAP, 4, 4)
VP8_PREDICT(bilinear, OP_BILINEAR, 16, 16)
VP8_PREDICT(bilinear, OP_BILINEAR, 8, 8)
VP8_PREDICT(bilinear, OP_BILINEAR, 8, 4)
VP8_PREDICT(bilinear, OP_BILINEAR, 4, 4)

}  // extern "C"
