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
template <class F>
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
        const GLOBAL_AS unsigned char *sbase;
        GLOBAL_AS unsigned char *dbase;
        int r;
        unit_of(unit, sbase, dbase, r);
        sbase += (long)r * (cols + 1) * VP8_TILE_BYTES;
        dbase += (long)(r * (luma ? 16 : 8)) * stride;
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

extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(16)))
vp8_detile_kf_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g)
{
    const int rows = g.mb_rows;
    detile_body(njobs * rows, g, [&](int unit, const GLOBAL_AS unsigned char *&sbase, GLOBAL_AS unsigned char *&dbase, int &r) {
        const int j = unit / rows;
        r = unit - j * rows;
        sbase = (const GLOBAL_AS unsigned char *)jobs[j].tile;
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
    detile_body(count * rows, g, [&](int unit, const GLOBAL_AS unsigned char *&sbase, GLOBAL_AS unsigned char *&dbase, int &r) {
        const int j = unit / rows;
        r = unit - j * rows;
        sbase = (const GLOBAL_AS unsigned char *)(tiles + tstride * (size_t)j);
        dbase = (GLOBAL_AS unsigned char *)(dst + dstride * (size_t)j);
    });
}
