// Output-side post-processing filters of the reference (vp8/common/postproc.c) for gfx950: the deblocking filter
// (vp8_post_proc_down_and_across_c :132-221), the two passes of the demacroblocking filter (vp8_mbpost_proc_across_ip_c
// :230-277, vp8_mbpost_proc_down_c :283-325), the noise adder (vp8_plane_add_noise_c :489-513) and the multiframe quality
// enhancement (vp8_multiframe_quality_enhance :802-900).  They run on the frame the decoder is about to show, into a separate
// output buffer, and never feed back into decoding.
//
// The reference filters in place through small ring buffers that hold every write back until the pixel can no longer be
// read, so each filter is a pure function of its input plane.  Here each is a plane -> plane kernel with one thread per
// output pixel, 256 consecutive pixels of one row per workgroup (byte accesses of a wave coalesce into 64-byte requests; the
// 5- and 15-tap neighbourhoods are served by L1/L2 or staged in LDS).  HBM-bound byte work: a plane is read once from HBM
// and written once per filter; vp8hip_postproc (vp8hip.hip) chains them on the decoder's stream.
#include "vp8_common.hip.h"

namespace {

__device__ __forceinline__ int iabs_(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// weights 1 1 4 1 1, + 4, >> 3; the centre pixel stays when any of the five differs from it by more than flimit
__device__ __forceinline__ int five_tap(int v, const int p[5], int flimit)
{
    bool keep = false;
    int k = 4 + 3 * p[2];
#pragma unroll
    for (int i = 0; i < 5; i++) { keep |= iabs_(v - p[i]) > flimit; k += p[i]; }
    return keep ? v : k >> 3;
}

__global__ __launch_bounds__(256) void vp8_pp_down_across_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                                   int stride, int rows, int cols, int flimit)
{
    __shared__ int down[256 + 4];       // vertically filtered pixels of columns c0 - 2 .. c0 + 257 (row ends replicated)
    const int t = threadIdx.x, c0 = blockIdx.x * 256;
    auto vertical = [&](int c) -> int {
        c = clampi(c, 0, cols - 1);
        int p[5];
#pragma unroll
        // (rows above / below the picture: the reference reads the frame's border there, which repeats the edge row,
        // vp8_yv12_extend_frame_borders; clamping the row says the same and asks nothing of the source's border)
        for (int i = 0; i < 5; i++) p[i] = src[(long)clampi((int)blockIdx.y + i - 2, 0, rows - 1) * stride + c];
        return five_tap(p[2], p, flimit);
    };
    down[t + 2] = vertical(c0 + t);
    if (t < 2) down[t] = vertical(c0 - 2 + t);
    if (t >= 254) down[t + 4] = vertical(c0 + t + 2);
    __syncthreads();
    if (c0 + t < cols) {
        int p[5];
#pragma unroll
        for (int i = 0; i < 5; i++) p[i] = down[t + i];
        dst[(long)blockIdx.y * stride + c0 + t] = (uint8_t)five_tap(p[2], p, flimit);
    }
}

// a pixel becomes the rounded mean of itself and the 15 pixels centred on it where that window is flat
__device__ __forceinline__ int flat_mean(int v, int sum, int sumsq, int flimit, int round)
{
    return sumsq * 15 - sum * sum < flimit ? (round + sum + v) >> 4 : v;
}

__global__ __launch_bounds__(256) void vp8_pp_mb_across_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                                 int stride, int rows, int cols, int flimit)
{
    __shared__ int row[256 + 14];       // columns c0 - 7 .. c0 + 262, row ends replicated
    const int t = threadIdx.x, c0 = blockIdx.x * 256;
    const uint8_t *s = src + (long)blockIdx.y * stride;
    row[t] = s[clampi(c0 - 7 + t, 0, cols - 1)];
    if (t < 14) row[256 + t] = s[clampi(c0 + 249 + t, 0, cols - 1)];
    __syncthreads();
    if (c0 + t >= cols) return;
    int sum = 0, sumsq = 0;
#pragma unroll
    for (int i = 0; i < 15; i++) { const int v = row[t + i]; sum += v; sumsq += v * v; }
    dst[(long)blockIdx.y * stride + c0 + t] = (uint8_t)flat_mean(row[t + 7], sum, sumsq, flimit, 8);
}

// the same along columns; rounded by the dither table: rv points at vp8_rv + (63 & rand()) of this frame
__global__ __launch_bounds__(256) void vp8_pp_mb_down_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int stride,
                                                               int rows, int cols, int flimit, const short *__restrict__ rv)
{
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= cols) return;
    int sum = 0, sumsq = 0;
#pragma unroll
    for (int i = -7; i <= 7; i++) { const int v = src[(long)clampi(r + i, 0, rows - 1) * stride + c]; sum += v; sumsq += v * v; }
    const int round = rv[((c * 17) & 127) + (r & 127)];
    dst[(long)r * stride + c] = (uint8_t)flat_mean(src[(long)r * stride + c], sum, sumsq, flimit, round);
}

// clamp away from black and white by `clamp`, add the noise row that starts row_offset[r] into the table; the sum wraps
__global__ __launch_bounds__(256) void vp8_pp_add_noise_kernel(uint8_t *plane, int stride, int rows, int cols, int clamp,
                                                                 const signed char *__restrict__ noise,
                                                                 const uint8_t *__restrict__ row_offset)
{
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= cols) return;
    int v = plane[(long)r * stride + c];
    if (v < clamp) v = clamp;                                   // blackclamp[0], then 255 + whiteclamp[0] (postproc.c:503-507)
    if (v > 255 + (signed char)clamp) v = 255 + (signed char)clamp;
    plane[(long)r * stride + c] = (uint8_t)(v + noise[row_offset[r] + c]);
}

// vp8_multiframe_quality_enhance (postproc.c:802-900) with multiframe_quality_enhance_block (:696-800).  One wave per
// macroblock: lane l holds four luma pixels of row l >> 2 (of the frame about to be shown, `show`, and of the picture shown
// before, `prev`), lanes 0..31 four chroma pixels as well.  cls (a byte per macroblock, from the host: the decision needs the
// frame type, the prediction mode and the motion vector): 0 = moved too far, copy; 1 = one 16x16 block; 2 = four 8x8 blocks
// (B_PRED / SPLITMV).  Per block: activity of the old picture (variance against zero), SAD old / new, both scaled to a pixel;
// SAD under the threshold thr = qdiff / 8 + log2(act) + log4(qprev): blend 16ths (new * f + old * (16 - f), f = 16 * sad / thr
// >> qdiff / 32; f = 0 keeps the old picture), else copy the new one.  The sums run over the lanes of a block by xor-shuffles
// (lane bits 0 and 2..4 span an 8x8 quadrant, bits 1 and 5 pick the quadrant).  out may be prev (every word is read by the
// lane that writes it, before it writes).
// The reference takes the activity from vp8_variance16x16_c, which squares the pixel sum in a signed int (encoder/
// variance_c.c:65-79): for sums from 46341 the product wraps negative and its arithmetic shift adds 2^24 to the variance.
// The reference build does exactly that and its output is what the fixtures pin, so it is reproduced: int multiply, >> 8.
__global__ __launch_bounds__(64) void vp8_pp_mfqe_kernel(const uint8_t *__restrict__ show, const uint8_t *prev, uint8_t *out,
                                                          DevGeom g, const uint8_t *__restrict__ cls, int qcurr, int qprev)
{
    const int l = threadIdx.x, mc = blockIdx.x, mr = blockIdx.y;
    const int kind = cls[mr * g.mb_cols + mc];
    const int row = l >> 2, col = (l & 3) * 4;
    const long yo = g.y_off + (long)(16 * mr + row) * g.y_stride + 16 * mc + col;
    const int crow = (l & 15) >> 1, ccol = (l & 1) * 4;
    const long co = (l < 16 ? g.u_off : g.v_off) + (long)(8 * mr + crow) * g.uv_stride + 8 * mc + ccol;
    const bool chroma = l < 32;
    const unsigned int s = *(const unsigned int *)(show + yo), d = *(const unsigned int *)(prev + yo);
    const unsigned int cs = chroma ? *(const unsigned int *)(show + co) : 0u, cd = chroma ? *(const unsigned int *)(prev + co) : 0u;
    int sum = 0, sse = 0, sad = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int a = (s >> (8 * k)) & 255, b = (d >> (8 * k)) & 255;
        sum += b; sse += b * b; sad += iabs_(a - b);
    }
    // totals of the lane's 8x8 quadrant, then (kind 1) of the macroblock
#pragma unroll
    for (int m : { 1, 4, 8, 16 }) { sum += __shfl_xor(sum, m); sse += __shfl_xor(sse, m); sad += __shfl_xor(sad, m); }
    if (kind == 1) {
#pragma unroll
        for (int m : { 2, 32 }) { sum += __shfl_xor(sum, m); sse += __shfl_xor(sse, m); sad += __shfl_xor(sad, m); }
    }
    const int sh = kind == 1 ? 8 : 6, rnd = 1 << (sh - 1);
    const int sq = (int)((unsigned)sum * (unsigned)sum);                 // see above: wraps like the reference's int
    unsigned int act = ((unsigned)sse - (unsigned)(sq >> sh) + (unsigned)rnd) >> sh;
    const unsigned int sadp = ((unsigned)sad + (unsigned)rnd) >> sh;
    const int qdiff = qcurr - qprev;
    unsigned int thr = (unsigned)(qdiff >> 3) + (act ? 31u - (unsigned)__clz((int)act) : 0u);
    for (int q = qprev >> 2; q; q >>= 2) thr++;
    // the lane's decision for its luma quadrant: 16 = copy the new picture, 0..15 = weight of the new picture
    int f = 16;
    if (kind != 0 && sadp < thr) f = (int)((sadp << 4) / thr) >> (qdiff >> 5);
    const int cq = ((crow >> 2) << 1) | (l & 1);                         // the chroma pixels' quadrant, and a luma lane inside it
    const int fc = __shfl(f, (cq >> 1) * 32 + (cq & 1) * 2);
    auto blend = [](unsigned int nw, unsigned int old, int w) -> unsigned int {
        if (w == 16) return nw;
        if (w == 0) return old;
        unsigned int r = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int a = (nw >> (8 * k)) & 255, b = (old >> (8 * k)) & 255;
            r |= (unsigned)((a * w + b * (16 - w) + 8) >> 4) << (8 * k);
        }
        return r;
    };
    *(unsigned int *)(out + yo) = blend(s, d, f);
    if (chroma) *(unsigned int *)(out + co) = blend(cs, cd, fc);
}

dim3 grid_for(int rows, int cols) { return dim3((unsigned)((cols + 255) / 256), (unsigned)rows); }

}  // namespace

// launch wrappers used by vp8hip_postproc (vp8hip.hip)
void vp8pp_down_and_across(hipStream_t st, const uint8_t *src, uint8_t *dst, int stride, int rows, int cols, int flimit)
{
    hipLaunchKernelGGL(vp8_pp_down_across_kernel, grid_for(rows, cols), dim3(256), 0, st, src, dst, stride, rows, cols, flimit);
}
void vp8pp_mb_across(hipStream_t st, const uint8_t *src, uint8_t *dst, int stride, int rows, int cols, int flimit)
{
    hipLaunchKernelGGL(vp8_pp_mb_across_kernel, grid_for(rows, cols), dim3(256), 0, st, src, dst, stride, rows, cols, flimit);
}
void vp8pp_mb_down(hipStream_t st, const uint8_t *src, uint8_t *dst, int stride, int rows, int cols, int flimit, const short *rv)
{
    hipLaunchKernelGGL(vp8_pp_mb_down_kernel, grid_for(rows, cols), dim3(256), 0, st, src, dst, stride, rows, cols, flimit, rv);
}
void vp8pp_add_noise(hipStream_t st, uint8_t *plane, int stride, int rows, int cols, int clamp, const signed char *noise,
                     const uint8_t *row_offset)
{
    hipLaunchKernelGGL(vp8_pp_add_noise_kernel, grid_for(rows, cols), dim3(256), 0, st, plane, stride, rows, cols, clamp, noise,
                       row_offset);
}
void vp8pp_mfqe(hipStream_t st, const uint8_t *show, const uint8_t *prev, uint8_t *out, const DevGeom &g, const uint8_t *cls,
                int qcurr, int qprev)
{
    hipLaunchKernelGGL(vp8_pp_mfqe_kernel, dim3((unsigned)g.mb_cols, (unsigned)g.mb_rows), dim3(64), 0, st, show, prev, out, g, cls,
                       qcurr, qprev);
}
