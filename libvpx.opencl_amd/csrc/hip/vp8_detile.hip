// Macroblock-tiled scratch frame -> raster frame buffer (the coded area; vp8_extend_kernel adds the borders).
//
// The one-MB-row-per-lane kernels keep a frame as one 384-byte tile per macroblock (VP8_TILE_BYTES: 16 luma
// rows of 16 B, 8 U rows of 8 B, 8 V rows of 8 B) so that every lane reads and writes whole 128-byte lines.
// The frame buffer the rest of the world sees -- reference frames for motion compensation, the frames handed
// back through vp8hip_frame_download -- is the reference decoder's raster YV12 layout
// (vpx_scale/generic/yv12config.c:55-112).  This pass is pure data movement at HBM speed: each workgroup
// takes one macroblock row of one frame, eight macroblocks per iteration; eight neighbouring threads read
// the same pixel row of eight tiles and write 128 (luma) / 64 (chroma) contiguous bytes.
#include "vp8_common.hip.h"

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

extern "C" __global__ void __launch_bounds__(256)
vp8_detile_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g)
{
    const DevJob &job = jobs[blockIdx.y];
    const int r = blockIdx.x, cols = g.mb_cols;
    const int t = threadIdx.x;
    const unsigned char *trow = job.ref[0] + (long)r * cols * VP8_TILE_BYTES;
    unsigned char *dst = job.dst;
    const int tile = t & 7;
    for (int c0 = 0; c0 < cols; c0 += 8) {
        const int c = c0 + tile;
        if (c >= cols) continue;
        const unsigned char *tp = trow + (long)c * VP8_TILE_BYTES;
        if (t < 128) {                                   // luma: row = t >> 3
            const int row = t >> 3;
            const u32x4_t v = *(const GLOBAL_AS u32x4_t *)(tp + 16 * row);
            *(GLOBAL_AS u32x4_t *)(dst + g.y_off + (long)(r * 16 + row) * g.y_stride + c * 16) = v;
        } else {                                         // chroma: U for t in 128..191, V for 192..255
            const int pl = (t - 128) >> 6, row = ((t - 128) >> 3) & 7;
            const u32x2_t v = *(const GLOBAL_AS u32x2_t *)(tp + 256 + 64 * pl + 8 * row);
            *(GLOBAL_AS u32x2_t *)(dst + (pl ? g.v_off : g.u_off) + (long)(r * 8 + row) * g.uv_stride + c * 8) = v;
        }
    }
}
