// VP8 in-loop deblocking filter + frame border extension for gfx950.  Replaces
//   vp8_loop_filter_frame / _frame_init / _update_sharpness / lf_init_lut   vp8/common/loopfilter.c:24-316
//   vp8_loop_filter_{mbv,bv,mbh,bh}_c, simple variants, edge filters         vp8/common/loopfilter_filters.c
//   vp8_yv12_extend_frame_borders                                            vpx_scale/generic/yv12extend.c:24-145
// (and the reference's own per-wavefront-level OpenCL offload, vp8/common/opencl/loopfilter_cl.c:521-707,
// which launched 2*(rows-1)+cols kernels per frame and uploaded one cl_uint per pixel).
//
// Mapping: the filter of MB(r,c) must run after MB(r,c-1), MB(r-1,c) and MB(r-1,c+1) -- the same
// wavefront as intra prediction.  One workgroup = one frame at a time (persistent over the jobs of
// a launch), one wave = one MB row, row r trails row r-1 by two MBs; progress flags live in LDS.
// The frame is filtered in place in HBM: a wave keeps a 20x20 luma / 12x12 chroma working tile in
// LDS (4 context pixels left and above), reads each MB once, and writes back the pixels it
// changed.  The 4 bottom rows a wave hands to the wave below travel through the frame itself
// (same CU, same L1/L2: workgroup-scope release/acquire).  Vertical edges use one lane per pixel
// row, horizontal edges one lane per pixel column; Y, U and V edges of the same kind share a step.
#include "vp8_common.hip.h"

#define LY_STRIDE 20
#define LC_STRIDE 12
#define LY_AT(y, x) (((y) + 4) * LY_STRIDE + (x) + 4)
#define LC_AT(y, x) (((y) + 4) * LC_STRIDE + (x) + 4)

struct __attribute__((aligned(16))) LfWaveLds {
    unsigned char tY[20 * LY_STRIDE];    // 400
    unsigned char tU[12 * LC_STRIDE];    // 144
    unsigned char tV[12 * LC_STRIDE];    // 144 -> 688
    unsigned char lvl[64];               // [seg][ref][mode] filter levels -> 752
    unsigned char pad[16];               // -> 768
};
static_assert(sizeof(LfWaveLds) % 16 == 0, "LfWaveLds alignment");

__device__ __forceinline__ int sc8(int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

// vp8_filter_mask (loopfilter_filters.c:27-40): true = filter this position
__device__ __forceinline__ bool lf_mask(int limit, int blimit, const int p[8])
{
    bool m = iabs(p[0] - p[1]) > limit;
    m |= iabs(p[1] - p[2]) > limit;
    m |= iabs(p[2] - p[3]) > limit;
    m |= iabs(p[5] - p[4]) > limit;
    m |= iabs(p[6] - p[5]) > limit;
    m |= iabs(p[7] - p[6]) > limit;
    m |= iabs(p[3] - p[4]) * 2 + iabs(p[2] - p[5]) / 2 > blimit;
    return !m;
}
// vp8_hevmask (:43-49)
__device__ __forceinline__ bool lf_hev(int thr, const int p[8])
{
    return iabs(p[2] - p[3]) > thr || iabs(p[5] - p[4]) > thr;
}
// vp8_filter (:51-95): p[2..5] = p1 p0 q0 q1
__device__ __forceinline__ void lf_inner(int p[8], bool mask, bool hev)
{
    int ps1 = p[2] - 128, ps0 = p[3] - 128, qs0 = p[4] - 128, qs1 = p[5] - 128;
    int f = sc8(ps1 - qs1);
    f = hev ? f : 0;
    f = sc8(f + 3 * (qs0 - ps0));
    f = mask ? f : 0;
    int f1 = sc8(f + 4) >> 3, f2 = sc8(f + 3) >> 3;
    p[4] = sc8(qs0 - f1) + 128;
    p[3] = sc8(ps0 + f2) + 128;
    f = (f1 + 1) >> 1;
    f = hev ? 0 : f;
    p[5] = sc8(qs1 - f) + 128;
    p[2] = sc8(ps1 + f) + 128;
}
// vp8_mbfilter (:161-214): p[1..6] = p2 p1 p0 q0 q1 q2
__device__ __forceinline__ void lf_mbedge(int p[8], bool mask, bool hev)
{
    int ps2 = p[1] - 128, ps1 = p[2] - 128, ps0 = p[3] - 128;
    int qs0 = p[4] - 128, qs1 = p[5] - 128, qs2 = p[6] - 128;
    int f = sc8(ps1 - qs1);
    f = sc8(f + 3 * (qs0 - ps0));
    f = mask ? f : 0;
    int f2 = hev ? f : 0;
    int f1 = sc8(f2 + 4) >> 3;
    f2 = sc8(f2 + 3) >> 3;
    qs0 = sc8(qs0 - f1);
    ps0 = sc8(ps0 + f2);
    f = hev ? 0 : f;
    int u = sc8((63 + f * 27) >> 7);
    p[4] = sc8(qs0 - u) + 128;
    p[3] = sc8(ps0 + u) + 128;
    u = sc8((63 + f * 18) >> 7);
    p[5] = sc8(qs1 - u) + 128;
    p[2] = sc8(ps1 + u) + 128;
    u = sc8((63 + f * 9) >> 7);
    p[6] = sc8(qs2 - u) + 128;
    p[1] = sc8(ps2 + u) + 128;
}
// vp8_simple_filter_mask + vp8_simple_filter (:292-315): p[2..5] = p1 p0 q0 q1
__device__ __forceinline__ void lf_simple(int p[8], int blimit)
{
    bool mask = iabs(p[3] - p[4]) * 2 + iabs(p[2] - p[5]) / 2 <= blimit;
    int p1 = p[2] - 128, p0 = p[3] - 128, q0 = p[4] - 128, q1 = p[5] - 128;
    int f = sc8(p1 - q1);
    f = sc8(f + 3 * (q0 - p0));
    f = mask ? f : 0;
    int f1 = sc8(f + 4) >> 3;
    p[4] = sc8(q0 - f1) + 128;
    int f2 = sc8(f + 3) >> 3;
    p[3] = sc8(p0 + f2) + 128;
}

struct LfParams { int mblim, blim, lim, hev_thr; };

// One edge, one position per lane.  `base` points at q0 of this lane's position inside the tile,
// `across` is the byte step over the edge.  kind: 0 inner normal, 1 MB-edge normal, 2 simple.
__device__ __forceinline__ void filter_position(unsigned char *base, int across, int kind, const LfParams &lp,
                                                int edge_limit)
{
    int p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = base[(i - 4) * across];
    if (kind == 2) {
        lf_simple(p, edge_limit);
        base[-across] = (unsigned char)p[3];
        base[0] = (unsigned char)p[4];
        return;
    }
    const bool m = lf_mask(lp.lim, edge_limit, p), hv = lf_hev(lp.hev_thr, p);
    if (kind == 1) {
        lf_mbedge(p, m, hv);
        base[-3 * across] = (unsigned char)p[1];
        base[2 * across] = (unsigned char)p[6];
    } else
        lf_inner(p, m, hv);
    base[-2 * across] = (unsigned char)p[2];
    base[-across] = (unsigned char)p[3];
    base[0] = (unsigned char)p[4];
    base[across] = (unsigned char)p[5];
}

// vp8_loop_filter_frame_init (loopfilter.c:117-201): level per [segment][ref_frame][mode class]
__device__ __forceinline__ void build_levels(const vp8ir_frame_hdr &h, unsigned char *lvl, int lane)
{
    const int seg = lane >> 4, ref = (lane >> 2) & 3, mode = lane & 3;
    int base = h.filter_level;
    if (h.segmentation_enabled) {
        if (h.mb_segment_abs_delta) base = h.segment_lf[seg];
        else { base += h.segment_lf[seg]; base = base < 0 ? 0 : (base > 63 ? 63 : base); }
    }
    int v;
    if (!h.mode_ref_lf_delta_enabled)
        v = base & 0xff;
    else {
        int rlev = base + h.ref_lf_deltas[ref];
        if (ref == 0) {
            if (mode == 0) v = rlev + h.mode_lf_deltas[0];
            else v = rlev;                       // only mode class 1 is ever looked up for intra
            if (mode > 1) v = 0;
        } else {
            v = mode == 0 ? 0 : rlev + h.mode_lf_deltas[mode];
        }
        v = v < 0 ? 0 : (v > 63 ? 63 : v);
    }
    lvl[lane] = (unsigned char)v;
}

// vp8_loop_filter_update_sharpness + hev threshold LUT (loopfilter.c:24-96)
__device__ __forceinline__ LfParams lf_params(int sharp, int level, int frame_type)
{
    LfParams l;
    int ilimit = level >> (sharp > 0);
    ilimit >>= (sharp > 4);
    if (sharp > 0 && ilimit > 9 - sharp) ilimit = 9 - sharp;
    if (ilimit < 1) ilimit = 1;
    l.lim = ilimit;
    l.blim = (2 * level + ilimit) & 0xff;
    l.mblim = (2 * (level + 2) + ilimit) & 0xff;
    if (level >= 40) l.hev_thr = frame_type == 0 ? 2 : 3;
    else if (level >= 20) l.hev_thr = frame_type == 0 ? 1 : 2;
    else if (level >= 15) l.hev_thr = 1;
    else l.hev_thr = 0;
    return l;
}

extern "C" __global__ void __launch_bounds__(1024)
vp8_loopfilter_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NW = blockDim.x >> 6;
    const int cols = g.mb_cols, rows = g.mb_rows;
    int *prog = (int *)smem;
    LfWaveLds *wl = (LfWaveLds *)(smem + 256) + wave;
    unsigned char *tY = wl->tY, *tU = wl->tU, *tV = wl->tV;

    if (threadIdx.x < 64) prog[threadIdx.x] = 0;
    __syncthreads();

    const int myjobs = (njobs - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total_rows = myjobs * rows;
    const int dep_wave = (wave + NW - 1) % NW;
    // mode_lf_lut (loopfilter.c:52-63) indexed by MB mode: DC,V,H,TM -> 1, B_PRED -> 0,
    // NEAREST,NEAR,NEW -> 2, ZERO -> 1, SPLIT -> 3; packed 2 bits each.
    const unsigned mode_class = (1u) | (1u << 2) | (1u << 4) | (1u << 6) | (0u << 8) | (2u << 10) | (2u << 12)
                              | (1u << 14) | (2u << 16) | (3u << 18);

    for (int R = wave, k = 0; R < total_rows; R += NW, ++k) {
        const int jj = R / rows, r = R - jj * rows;
        const DevJob &job = jobs[blockIdx.x + jj * gridDim.x];
        const vp8ir_frame_hdr &hdr = job.hdr;
        const int dep_seq = (R - 1) / NW;
        if (hdr.filter_level == 0) {             // frame not filtered at all (onyxd_if.c:576)
            wg_publish_global(&prog[wave], (k + 1) << 16, lane);
            continue;
        }
        build_levels(hdr, wl->lvl, lane);
        wave_lds_sync();
        const bool simple = hdr.filter_type != 0;
        const int sharp = hdr.sharpness_level, ftype = hdr.frame_type;
        const vp8ir_mb *mbrow = job.mbs + (long)r * cols;
        uint8_t *fY = job.dst + g.y_off + (long)r * 16 * g.y_stride;
        uint8_t *fU = job.dst + g.u_off + (long)r * 8 * g.uv_stride;
        uint8_t *fV = job.dst + g.v_off + (long)r * 8 * g.uv_stride;

        // lane roles for loads/stores of the MB body: Y lane -> (row lane>>2, dword lane&3);
        // chroma lanes 0..31 -> plane lane>>4, row (lane>>1)&7, dword lane&1
        const int by = lane >> 2, bxd = (lane & 3) * 4;
        const int cpl = lane >> 4, cy = (lane >> 1) & 7, cxd = (lane & 1) * 4;
        unsigned int nY = *(const unsigned int *)(fY + (long)by * g.y_stride + bxd);
        unsigned int nC = 0;
        if (lane < 32) nC = *(const unsigned int *)((cpl ? fV : fU) + (long)cy * g.uv_stride + cxd);

        for (int c = 0; c < cols; ++c) {
            const vp8ir_mb &mb = mbrow[c];
            const int y_mode = mb.y_mode;
            const bool skip_lf = y_mode != VP8IR_B_PRED && y_mode != VP8IR_SPLITMV && (mb.flags & VP8IR_MB_SKIP);
            const int level = wl->lvl[((mb.segment_id & 3) << 4) | ((mb.ref_frame & 3) << 2)
                                      | ((mode_class >> (2 * y_mode)) & 3)];
            const unsigned int curY = nY, curC = nC;
            if (c + 1 < cols) {                  // prefetch the next MB's (still unfiltered) pixels
                nY = *(const unsigned int *)(fY + (long)by * g.y_stride + (c + 1) * 16 + bxd);
                if (lane < 32)
                    nC = *(const unsigned int *)((cpl ? fV : fU) + (long)cy * g.uv_stride + (c + 1) * 8 + cxd);
            }
            // ---- slide the tile: previous MB's 4 right-hand columns become the left context
            wave_lds_sync();
            if (c > 0) {
                if (lane < 16)
                    *(unsigned int *)(tY + LY_AT(lane, -4)) = *(const unsigned int *)(tY + LY_AT(lane, 12));
                else if (lane < 24)
                    *(unsigned int *)(tU + LC_AT(lane - 16, -4)) = *(const unsigned int *)(tU + LC_AT(lane - 16, 4));
                else if (lane < 32)
                    *(unsigned int *)(tV + LC_AT(lane - 24, -4)) = *(const unsigned int *)(tV + LC_AT(lane - 24, 4));
            }
            wave_lds_sync();
            *(unsigned int *)(tY + LY_AT(by, bxd)) = curY;
            if (lane < 32) *(unsigned int *)((cpl ? tV : tU) + LC_AT(cy, cxd)) = curC;

            // ---- top context: 4 rows above, final-so-far values written by the wave of row r-1
            if (r > 0) {
                wg_wait_ge(&prog[dep_wave], (dep_seq << 16) + min(c + 2, cols));
                if (lane < 16) {
                    const int ty = (lane >> 2) - 4, tx = (lane & 3) * 4;
                    *(unsigned int *)(tY + LY_AT(ty, tx)) =
                        *(const unsigned int *)(fY + (long)ty * g.y_stride + c * 16 + tx);
                } else if (lane < 32) {
                    const int pl = (lane >> 3) & 1, ty = ((lane >> 1) & 3) - 4, tx = (lane & 1) * 4;
                    *(unsigned int *)((pl ? tV : tU) + LC_AT(ty, tx)) =
                        *(const unsigned int *)((pl ? fV : fU) + (long)ty * g.uv_stride + c * 8 + tx);
                }
            }

            wave_lds_sync();
            if (level) {
                const LfParams lp = lf_params(sharp, level, ftype);
                // position roles: lanes 0..15 luma position, 16..23 U, 24..31 V
                unsigned char *tile = lane < 16 ? tY : (lane < 24 ? tU : tV);
                const int stride = lane < 16 ? LY_STRIDE : LC_STRIDE;
                const int pos = lane < 16 ? lane : (lane & 7);
                const int origin = lane < 16 ? LY_AT(0, 0) : LC_AT(0, 0);
                if (!simple) {
                    if (lane < 32) {
                        // vertical edges: position = pixel row, step across = 1
                        unsigned char *rowp = tile + origin + pos * stride;
                        if (c > 0) filter_position(rowp, 1, 1, lp, lp.mblim);
                        if (!skip_lf) {
                            filter_position(rowp + 4, 1, 0, lp, lp.blim);
                            if (lane < 16) {
                                filter_position(rowp + 8, 1, 0, lp, lp.blim);
                                filter_position(rowp + 12, 1, 0, lp, lp.blim);
                            }
                        }
                    }
                    wave_lds_sync();
                    if (lane < 32) {
                        // horizontal edges: position = pixel column, step across = stride
                        unsigned char *colp = tile + origin + pos;
                        if (r > 0) filter_position(colp, stride, 1, lp, lp.mblim);
                        if (!skip_lf) {
                            filter_position(colp + 4 * stride, stride, 0, lp, lp.blim);
                            if (lane < 16) {
                                filter_position(colp + 8 * stride, stride, 0, lp, lp.blim);
                                filter_position(colp + 12 * stride, stride, 0, lp, lp.blim);
                            }
                        }
                    }
                } else if (lane < 16) {          // simple filter: luma only (loopfilter.c:284-299)
                    unsigned char *rowp = tY + LY_AT(lane, 0);
                    if (c > 0) filter_position(rowp, 1, 2, lp, lp.mblim);
                    if (!skip_lf) {
                        filter_position(rowp + 4, 1, 2, lp, lp.blim);
                        filter_position(rowp + 8, 1, 2, lp, lp.blim);
                        filter_position(rowp + 12, 1, 2, lp, lp.blim);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    unsigned char *colp = tY + LY_AT(0, lane);
                    if (r > 0) filter_position(colp, LY_STRIDE, 2, lp, lp.mblim);
                    if (!skip_lf) {
                        filter_position(colp + 4 * LY_STRIDE, LY_STRIDE, 2, lp, lp.blim);
                        filter_position(colp + 8 * LY_STRIDE, LY_STRIDE, 2, lp, lp.blim);
                        filter_position(colp + 12 * LY_STRIDE, LY_STRIDE, 2, lp, lp.blim);
                    }
                }
            }

            // ---- write back: MB body, the 4 context rows above (cols 0..15) and the 4 context
            // columns to the left (rows 0..15); the corner is never touched by this MB's filters.
            wave_lds_sync();
            *(unsigned int *)(fY + (long)by * g.y_stride + c * 16 + bxd) = *(const unsigned int *)(tY + LY_AT(by, bxd));
            if (lane < 32)
                *(unsigned int *)((cpl ? fV : fU) + (long)cy * g.uv_stride + c * 8 + cxd) =
                    *(const unsigned int *)((cpl ? tV : tU) + LC_AT(cy, cxd));
            if (r > 0) {
                if (lane < 16) {
                    const int ty = (lane >> 2) - 4, tx = (lane & 3) * 4;
                    *(unsigned int *)(fY + (long)ty * g.y_stride + c * 16 + tx) = *(const unsigned int *)(tY + LY_AT(ty, tx));
                } else if (lane < 32) {
                    const int pl = (lane >> 3) & 1, ty = ((lane >> 1) & 3) - 4, tx = (lane & 1) * 4;
                    *(unsigned int *)((pl ? fV : fU) + (long)ty * g.uv_stride + c * 8 + tx) =
                        *(const unsigned int *)((pl ? tV : tU) + LC_AT(ty, tx));
                }
            }
            if (c > 0) {
                if (lane >= 32 && lane < 48) {
                    const int yy = lane - 32;
                    *(unsigned int *)(fY + (long)yy * g.y_stride + c * 16 - 4) = *(const unsigned int *)(tY + LY_AT(yy, -4));
                } else if (lane >= 48) {
                    const int pl = (lane >> 3) & 1, yy = lane & 7;
                    *(unsigned int *)((pl ? fV : fU) + (long)yy * g.uv_stride + c * 8 - 4) =
                        *(const unsigned int *)((pl ? tV : tU) + LC_AT(yy, -4));
                }
            }
            wg_publish_global(&prog[wave], c + 1 == cols ? (k + 1) << 16 : (k << 16) + c + 1, lane);
        }
    }
}

// ---- border extension ------------------------------------------------------------------------
// vp8_yv12_extend_frame_borders (yv12extend.c:24-145): replicate the first/last pixel of every row
// 32 (luma) / 16 (chroma) times, then the first/last (already widened) row 32 / 16 times.
// grid = (rows-of-work, njobs); every thread writes one dword.
extern "C" __global__ void __launch_bounds__(256)
vp8_extend_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g)
{
    const DevJob &job = jobs[blockIdx.y];
    // plane table
    for (int plane = 0; plane < 3; ++plane) {
        const int w = plane ? g.aligned_w / 2 : g.aligned_w, h = plane ? g.aligned_h / 2 : g.aligned_h;
        const int stride = plane ? g.uv_stride : g.y_stride, border = plane ? 16 : 32;
        uint8_t *p = job.dst + (plane == 0 ? g.y_off : plane == 1 ? g.u_off : g.v_off);
        const int full_w = w + 2 * border;               // bytes per widened row
        const int dw_per_row = full_w / 4;
        // phase A (left/right of every image row) and phase B (top/bottom rows) are fused: a thread
        // owns one dword of one row of the widened plane (rows -border .. h+border-1) that lies in
        // the border, and computes its value directly from the clamped source coordinate.
        const long total = (long)(h + 2 * border) * dw_per_row;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const int row = (int)(i / dw_per_row) - border, xd = (int)(i % dw_per_row) * 4 - border;
            const bool inside_x = xd >= 0 && xd < w, inside_y = row >= 0 && row < h;
            if (inside_x && inside_y) continue;
            const int sy = row < 0 ? 0 : (row >= h ? h - 1 : row);
            unsigned int v;
            if (inside_x) v = *(const unsigned int *)(p + (long)sy * stride + xd);
            else v = p[(long)sy * stride + (xd < 0 ? 0 : w - 1)] * 0x01010101u;
            *(unsigned int *)(p + (long)row * stride + xd) = v;
        }
    }
}
