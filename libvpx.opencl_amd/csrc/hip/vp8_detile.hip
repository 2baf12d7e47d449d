// Macroblock-window tiles -> raster frame buffer.
//
// The one-MB-row-per-lane kernels (vp8_keyframe_simt.hip) leave a frame as 384-byte tiles, one per macroblock and one more per
// macroblock row, so that every lane writes whole 64-byte half lines.  The frame buffer the rest of the world sees -- reference
// frames for motion compensation, the frames handed back through vp8hip_frame_download -- is the reference decoder's raster
// YV12 layout with its 32-pixel borders (vpx_scale/generic/yv12config.c:55-112).  This pass is pure data movement;
// vp8_extend_kernel (vp8_loopfilter.hip) adds the borders behind it.
#include "vp8_common.hip.h"

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef GLOBAL_AS u32x4_t *g_x4p;
typedef GLOBAL_AS u32x2_t *g_x2p;

// The tiles vp8_keyframe_kernel / vp8_interframe_kernel leave (vp8_keyframe_simt.hip): rows x (cols + 1) tiles per frame; luma rows
// 0..11 (chroma rows 0..3) of tile c hold the pixel columns 16c-4 .. 16c+11 (8c-4 .. 8c+3) -- the macroblock's WINDOW, shifted
// left by the four pixels its left-edge filter still changes --, luma rows 12..15 (chroma 4..7) the macroblock's own columns.
// Eight neighbouring threads copy the same pixel row of eight tiles: 128 (64) contiguous bytes of a frame row, for the
// window rows at an offset of -4 (the pieces of neighbouring groups complete the lines).  The first window's four pixels left of
// the frame and the last window's pixels right of it land in the border, which vp8_extend_kernel writes afterwards.
//
// The pass keeps to 16 registers (two pointers, one piece of data, a counter; one load in flight per thread -- memory-level
// parallelism comes from the waves, not from unrolling): through round 3 it ran after every large launch, beside the next
// launch's vp8_keyframe_kernel, in the 16 registers that kernel's two waves leave free on a SIMD; now it runs when a frame is
// asked for in raster form (vp8hip_launch.hip).
typedef u32x4_t u32x4_u4 __attribute__((aligned(4)));
typedef u32x2_t u32x2_u4 __attribute__((aligned(4)));
// `unit_of(unit, sbase, dbase)`: where macroblock row r of frame j (unit = j * rows + r) has its tiles and where the frame's raster form is
// RETILE: the other way round -- the tiles of a frame that only exists in raster form (with its borders: the first window's four
// pixels left of the frame and the last window's right of it are border pixels, which no reader of tiles looks at)
template <bool RETILE = false, class F>
__device__ __forceinline__ void detile_body(int nunits, DevGeom g, F unit_of)
{
    const int cols = g.mb_cols;
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
    // a workgroup takes macroblock rows blockIdx.x, blockIdx.x + gridDim.x, ... of the launch
#pragma unroll 1
    for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
        GLOBAL_AS unsigned char *sbase;
        GLOBAL_AS unsigned char *dbase;
        int r;
        unit_of(unit, sbase, dbase, r);
        sbase += (long)r * (cols + 1) * VP8_TILE_BYTES;
        dbase += (long)(r * (luma ? 16 : 8)) * stride;
        unsigned so = soff, dO = doff;
        if (luma) {
#pragma unroll 1
            for (int left = ntiles; left > 0; left -= 8, so += 8 * VP8_TILE_BYTES, dO += 128) {
                if constexpr (RETILE) *(GLOBAL_AS u32x4_t *)(sbase + so) = *(const GLOBAL_AS u32x4_u4 *)(dbase + dO);
                else *(GLOBAL_AS u32x4_u4 *)(dbase + dO) = *(const GLOBAL_AS u32x4_t *)(sbase + so);
            }
        } else {
#pragma unroll 1
            for (int left = ntiles; left > 0; left -= 8, so += 8 * VP8_TILE_BYTES, dO += 64) {
                if constexpr (RETILE) *(GLOBAL_AS u32x2_t *)(sbase + so) = *(const GLOBAL_AS u32x2_u4 *)(dbase + dO);
                else *(GLOBAL_AS u32x2_u4 *)(dbase + dO) = *(const GLOBAL_AS u32x2_t *)(sbase + so);
            }
        }
    }
}

extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(16)))
vp8_detile_kf_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g)
{
    const int rows = g.mb_rows;
    detile_body(njobs * rows, g, [&](int unit, GLOBAL_AS unsigned char *&sbase, GLOBAL_AS unsigned char *&dbase, int &r) {
        const int j = unit / rows;
        r = unit - j * rows;
        sbase = (GLOBAL_AS unsigned char *)jobs[j].tile;
        dbase = (GLOBAL_AS unsigned char *)jobs[j].dst;
    });
}

// Raster frame buffer -> macroblock-window tiles: a reference frame that only exists in raster form -- decoded by a small launch,
// uploaded (VP8_SET_REFERENCE) -- for a launch that reads its other references as tiles (vp8_inter_pred_tiles_kernel)
extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(16)))
vp8_retile_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g)
{
    const int rows = g.mb_rows;
    detile_body<true>(njobs * rows, g, [&](int unit, GLOBAL_AS unsigned char *&sbase, GLOBAL_AS unsigned char *&dbase, int &r) {
        const int j = unit / rows;
        r = unit - j * rows;
        sbase = (GLOBAL_AS unsigned char *)jobs[j].tile;
        dbase = (GLOBAL_AS unsigned char *)jobs[j].dst;
    });
}

// The same for `count` frames whose tiled forms stand tstride bytes apart, into raster forms dstride bytes apart -- page-locked
// HOST memory (vp8hip_frames_fetch_async): the frames leave the device through this pass, over PCIe, and their raster form never
// exists in HBM.  (The left / right border columns next to the picture receive the first window's four pixels left of the frame and
// the last window's pixels right of it: the destination is a whole frame buffer with its borders, whose contents outside the
// picture are not defined by this path.)
extern "C" __global__ void __launch_bounds__(256)
vp8_detile_run_kernel(const uint8_t *__restrict__ tiles, size_t tstride, uint8_t *__restrict__ dst, size_t dstride, int count, DevGeom g)
{
    const int rows = g.mb_rows;
    detile_body(count * rows, g, [&](int unit, GLOBAL_AS unsigned char *&sbase, GLOBAL_AS unsigned char *&dbase, int &r) {
        const int j = unit / rows;
        r = unit - j * rows;
        sbase = (GLOBAL_AS unsigned char *)(tiles + tstride * (size_t)j);
        dbase = (GLOBAL_AS unsigned char *)(dst + dstride * (size_t)j);
    });
}

// ---- frames that leave as PACKED I420 (vp8hip_frames_fetch_i420_async): w x h luma, then the two (w / 2) x ((h + 1) / 2) chroma
// planes, back to back, no borders -- what `vpxdec --i420` writes and what the MD5s are taken over.  Whole frame buffers carry 10 %
// of border over the link (1080p: 3.43 MB for 3.11 MB of picture), and the link is what a pipeline that downloads every frame is
// bound by.  w a multiple of 8: every piece below is whole dwords.
// From tiles: the thread arrangement of detile_body; a piece is stored dword by dword where it lies inside the picture.
extern "C" __global__ void __launch_bounds__(256)
vp8_pack_i420_tiles_kernel(const uint8_t *__restrict__ tiles, size_t tstride, uint8_t *__restrict__ dst, size_t dstride, int count, DevGeom g,
                           int w, int h)
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
    const int cw = w >> 1, ch = (h + 1) >> 1;
    const int pw = luma ? w : cw, ph = luma ? h : ch;               // the plane's picture
    const long poff = luma ? 0 : (long)w * h + (pl ? (long)cw * ch : 0);
    const unsigned soff = (unsigned)(tile * VP8_TILE_BYTES + src);
    const int ntiles = (window ? cols + 1 : cols) - tile;
    const int x_first = tile * (luma ? 16 : 8) - (window ? 4 : 0);
#pragma unroll 1
    for (int unit = blockIdx.x; unit < count * rows; unit += gridDim.x) {
        const int j = unit / rows, r = unit - j * rows;
        const GLOBAL_AS unsigned char *sbase = (const GLOBAL_AS unsigned char *)(tiles + tstride * (size_t)j) + (long)r * (cols + 1) * VP8_TILE_BYTES;
        const int y = r * (luma ? 16 : 8) + row;
        if (y >= ph) continue;
        GLOBAL_AS unsigned char *drow = (GLOBAL_AS unsigned char *)(dst + dstride * (size_t)j) + poff + (long)y * pw;
        unsigned so = soff;
        int x = x_first;
        if (luma) {
#pragma unroll 1
            for (int left = ntiles; left > 0; left -= 8, so += 8 * VP8_TILE_BYTES, x += 128) {
                const u32x4_t v = *(const GLOBAL_AS u32x4_t *)(sbase + so);
                if (x >= 0 && x + 16 <= pw) *(GLOBAL_AS u32x4_u4 *)(drow + x) = v;
                else {
                    if (x >= 0 && x + 4 <= pw) *(GLOBAL_AS unsigned int *)(drow + x) = v.x;
                    if (x + 4 >= 0 && x + 8 <= pw) *(GLOBAL_AS unsigned int *)(drow + x + 4) = v.y;
                    if (x + 8 >= 0 && x + 12 <= pw) *(GLOBAL_AS unsigned int *)(drow + x + 8) = v.z;
                    if (x + 12 >= 0 && x + 16 <= pw) *(GLOBAL_AS unsigned int *)(drow + x + 12) = v.w;
                }
            }
        } else {
#pragma unroll 1
            for (int left = ntiles; left > 0; left -= 8, so += 8 * VP8_TILE_BYTES, x += 64) {
                const u32x2_t v = *(const GLOBAL_AS u32x2_t *)(sbase + so);
                if (x >= 0 && x + 4 <= pw) *(GLOBAL_AS unsigned int *)(drow + x) = v.x;
                if (x + 4 >= 0 && x + 8 <= pw) *(GLOBAL_AS unsigned int *)(drow + x + 4) = v.y;
            }
        }
    }
}

// From the raster form: a workgroup per plane row, dword by dword.  frames: frame buffer 0 of the pool, fstride apart; first: the first one of the run.
extern "C" __global__ void __launch_bounds__(256)
vp8_pack_i420_raster_kernel(const uint8_t *__restrict__ frames, size_t fstride, int first, uint8_t *__restrict__ dst, size_t dstride, int count,
                            DevGeom g, int w, int h)
{
    const int cw = w >> 1, ch = (h + 1) >> 1;
    const int per_frame = h + 2 * ch;
#pragma unroll 1
    for (long unit = blockIdx.x; unit < (long)count * per_frame; unit += gridDim.x) {
        const int j = (int)(unit / per_frame);
        int y = (int)(unit - (long)j * per_frame);
        const int pl = y < h ? 0 : y < h + ch ? 1 : 2;
        if (pl) y -= h + (pl - 1) * ch;
        const int pw = pl ? cw : w;
        const GLOBAL_AS unsigned int *s = (const GLOBAL_AS unsigned int *)(frames + fstride * (size_t)(first + j) + (pl == 0 ? g.y_off : pl == 1 ? g.u_off : g.v_off) +
                                                                           (long)y * (pl ? g.uv_stride : g.y_stride));
        GLOBAL_AS unsigned int *d = (GLOBAL_AS unsigned int *)(dst + dstride * (size_t)j + (pl == 0 ? 0 : (long)w * h + (pl == 2 ? (long)cw * ch : 0)) + (long)y * pw);
        for (int i = threadIdx.x; i < pw / 4; i += 256) d[i] = s[i];
    }
}
