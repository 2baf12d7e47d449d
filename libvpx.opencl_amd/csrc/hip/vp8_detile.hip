// Macroblock-tiled scratch frame -> raster frame buffer, optionally with the reference's border extension.
//
// The one-MB-row-per-lane kernels keep a frame as one 384-byte tile per macroblock (VP8_TILE_BYTES: 16 luma
// rows of 16 B, 8 U rows of 8 B, 8 V rows of 8 B) so that every lane reads and writes whole 128-byte lines.
// The frame buffer the rest of the world sees -- reference frames for motion compensation, the frames handed
// back through vp8hip_frame_download -- is the reference decoder's raster YV12 layout with its 32-pixel
// borders (vpx_scale/generic/yv12config.c:55-112).  This pass is pure data movement: each workgroup takes one
// macroblock row of one frame, eight macroblocks per iteration; eight neighbouring threads read the same
// pixel row of eight tiles and write 128 (luma) / 64 (chroma) contiguous bytes.  With `extend` set it also
// does vp8_yv12_extend_frame_borders (vpx_scale/generic/yv12extend.c:24-145) on the fly -- the thread that
// holds the first / last pixels of a row replicates them into the left / right border, the first / last
// macroblock row replicates its outer pixel row (borders included) 32 (chroma: 16) times -- so a macroblock
// row leaves the workgroup as one contiguous run of 16 full frame-buffer rows.
#include "vp8_common.hip.h"

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef GLOBAL_AS u32x4_t *g_x4p;
typedef GLOBAL_AS u32x2_t *g_x2p;

extern "C" __global__ void __launch_bounds__(256)
vp8_detile_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int extend)
{
    const DevJob &job = jobs[blockIdx.y];
    const int r = blockIdx.x, cols = g.mb_cols, rows = g.mb_rows;
    const int t = threadIdx.x;
    const unsigned char *trow = job.tile + (long)r * cols * VP8_TILE_BYTES;
    unsigned char *dst = job.dst;
    const int tile = t & 7;
    const bool luma = t < 128;
    const int pl = (t - 128) >> 6;                                  // chroma threads: 0 = U, 1 = V
    const int row = luma ? t >> 3 : ((t - 128) >> 3) & 7;           // pixel row inside the macroblock
    const int nrow = luma ? 16 : 8, border = luma ? 32 : 16;
    const long stride = luma ? g.y_stride : g.uv_stride;
    unsigned char *prow = dst + (luma ? g.y_off : (pl ? g.v_off : g.u_off)) + (long)(r * nrow + row) * stride;
    // rows of the top / bottom border this thread's row is copied to (vertical extension)
    const int vcopies = !extend ? 0 : ((r == 0 && row == 0) || (r == rows - 1 && row == nrow - 1)) ? border : 0;
    const long vstep = (r == 0 && row == 0) ? -stride : stride;
    // four iterations' worth of loads are issued before the first store: the pass is latency-bound otherwise
    for (int cb = 0; cb < cols; cb += 32) {
        u32x4_t vy[4];
        u32x2_t vc[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int c = cb + 8 * k + tile;
            if (c < cols) {
                const unsigned char *tp = trow + (long)c * VP8_TILE_BYTES;
                if (luma) vy[k] = *(const GLOBAL_AS u32x4_t *)(tp + 16 * row);
                else vc[k] = *(const GLOBAL_AS u32x2_t *)(tp + 256 + 64 * pl + 8 * row);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int c = cb + 8 * k + tile;
            if (c >= cols) continue;
            if (luma) {
                const u32x4_t v = vy[k];
                unsigned char *p = prow + c * 16;
                *(g_x4p)p = v;
                const bool lb = extend && c == 0, rb = extend && c == cols - 1;
                const unsigned int l = (v.x & 0xff) * 0x01010101u, rr = (v.w >> 24) * 0x01010101u;
                const u32x4_t lv = { l, l, l, l }, rv = { rr, rr, rr, rr };
                if (lb) { *(g_x4p)(p - 32) = lv; *(g_x4p)(p - 16) = lv; }
                if (rb) { *(g_x4p)(p + 16) = rv; *(g_x4p)(p + 32) = rv; }
                for (int b = 1; b <= vcopies; b++) {
                    unsigned char *q = p + b * vstep;
                    *(g_x4p)q = v;
                    if (lb) { *(g_x4p)(q - 32) = lv; *(g_x4p)(q - 16) = lv; }
                    if (rb) { *(g_x4p)(q + 16) = rv; *(g_x4p)(q + 32) = rv; }
                }
            } else {
                const u32x2_t v = vc[k];
                unsigned char *p = prow + c * 8;
                *(g_x2p)p = v;
                const bool lb = extend && c == 0, rb = extend && c == cols - 1;
                const unsigned int l = (v.x & 0xff) * 0x01010101u, rr = (v.y >> 24) * 0x01010101u;
                const u32x2_t lv = { l, l }, rv = { rr, rr };
                if (lb) { *(g_x2p)(p - 16) = lv; *(g_x2p)(p - 8) = lv; }
                if (rb) { *(g_x2p)(p + 8) = rv; *(g_x2p)(p + 16) = rv; }
                for (int b = 1; b <= vcopies; b++) {
                    unsigned char *q = p + b * vstep;
                    *(g_x2p)q = v;
                    if (lb) { *(g_x2p)(q - 16) = lv; *(g_x2p)(q - 8) = lv; }
                    if (rb) { *(g_x2p)(q + 8) = rv; *(g_x2p)(q + 16) = rv; }
                }
            }
        }
    }
}

// The same pass for the tiles vp8_keyframe_kernel leaves (vp8_keyframe_simt.hip): rows x (cols + 1) tiles per frame; luma rows
// 0..11 (chroma rows 0..3) of tile c hold the pixel columns 16c-4 .. 16c+11 (8c-4 .. 8c+3) -- the macroblock's WINDOW, shifted
// left by the four pixels its left-edge filter still changes --, luma rows 12..15 (chroma 4..7) the macroblock's own columns.
// Eight neighbouring threads copy the same pixel row of eight tiles: 128 (64) contiguous bytes of a frame row, for the
// window rows at an offset of -4 (the pieces of neighbouring groups complete the lines).  The first window's four pixels left of
// the frame and the last window's pixels right of it land in the border, which vp8_extend_kernel writes afterwards.
//
// This pass is meant to run BESIDE the next launch's vp8_keyframe_kernel, whose two waves per SIMD leave 16 of the SIMD's 512
// registers free: it keeps to 16 (two pointers, one piece of data, a counter; one load in flight per thread -- memory-level
// parallelism comes from the six free wave slots of every SIMD, not from unrolling), so its waves fit in that gap and the
// pass is hidden behind a kernel that is bound by the vector ALU.
typedef u32x4_t u32x4_u4 __attribute__((aligned(4)));
typedef u32x2_t u32x2_u4 __attribute__((aligned(4)));
extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(16)))
vp8_detile_kf_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g)
{
    const int cols = g.mb_cols, rows = g.mb_rows;
    const int t = threadIdx.x;
    const int tile = t & 7;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);          // waves 0, 1: luma; 2: U; 3: V
    const bool luma = wv < 2;
    const int pl = wv - 2;
    const int row = luma ? t >> 3 : ((t - 128) >> 3) & 7;           // pixel row inside the macroblock
    const bool window = luma ? row < 12 : row < 4;
    const int src = luma ? (window ? 16 * row : 192 + 16 * (row - 12)) : (window ? 256 + 32 * pl + 8 * row : 320 + 32 * pl + 8 * (row - 4));
    const int stride = luma ? g.y_stride : g.uv_stride;
    // per-thread 32-bit offsets; the 64-bit bases are uniform (scalar registers)
    const unsigned doff = (unsigned)((luma ? g.y_off : (pl ? g.v_off : g.u_off)) + row * stride - (window ? 4 : 0) + tile * (luma ? 16 : 8));
    const unsigned soff = (unsigned)(tile * VP8_TILE_BYTES + src);
    const int ntiles = (window ? cols + 1 : cols) - tile;           // tiles tile, tile + 8, ... of a macroblock row
    // a workgroup takes macroblock rows blockIdx.x, blockIdx.x + gridDim.x, ... of the launch (row r of job j: unit j * rows + r)
#pragma unroll 1
    for (int unit = blockIdx.x; unit < njobs * rows; unit += gridDim.x) {
        const int j = unit / rows, r = unit - j * rows;
        const GLOBAL_AS unsigned char *sbase = (const GLOBAL_AS unsigned char *)jobs[j].tile + (long)r * (cols + 1) * VP8_TILE_BYTES;
        GLOBAL_AS unsigned char *dbase = (GLOBAL_AS unsigned char *)jobs[j].dst + (long)(r * (luma ? 16 : 8)) * stride;
        unsigned so = soff, dO = doff;
        if (luma) {
#pragma unroll 1
            for (int left = ntiles; left > 0; left -= 8, so += 8 * VP8_TILE_BYTES, dO += 128)
                *(GLOBAL_AS u32x4_u4 *)(dbase + dO) = *(const GLOBAL_AS u32x4_t *)(sbase + so);
        } else {
#pragma unroll 1
            for (int left = ntiles; left > 0; left -= 8, so += 8 * VP8_TILE_BYTES, dO += 64)
                *(GLOBAL_AS u32x2_u4 *)(dbase + dO) = *(const GLOBAL_AS u32x2_t *)(sbase + so);
        }
    }
}
