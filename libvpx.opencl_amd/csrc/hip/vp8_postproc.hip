// Output-side post-processing filters of the reference (vp8/common/postproc.c) for gfx950: the deblocking filter
// (vp8_post_proc_down_and_across_c :132-221), the two passes of the demacroblocking filter (vp8_mbpost_proc_across_ip_c
// :230-277, vp8_mbpost_proc_down_c :283-325) and the noise adder (vp8_plane_add_noise_c :489-513).  They run on the frame the
// decoder is about to show, into a separate output buffer, and never feed back into decoding.
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
    const uint8_t *s = src + (long)blockIdx.y * stride;
    auto vertical = [&](int c) -> int {
        c = clampi(c, 0, cols - 1);
        int p[5];
#pragma unroll
        for (int i = 0; i < 5; i++) p[i] = s[c + (i - 2) * stride];
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
