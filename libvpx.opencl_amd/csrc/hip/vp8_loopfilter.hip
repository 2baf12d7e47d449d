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
#include "vp8_block_prims.hip.h"

// Per-wave LDS tile = a RING of 8 macroblock columns: luma 20 rows (y = -4..15) x 128 bytes, chroma
// 12 rows (y = -4..7) x 64 bytes, addressed by the absolute x inside the MB row modulo the ring.
// The filters work in place in the ring; finished columns are written to the frame four MBs at a
// time as 64-byte-aligned 64-byte row segments (full memory sectors instead of 16-byte pieces).
#define RING_MBS 8
#define LY_STRIDE 128
#define LC_STRIDE 64
#define LY_AT(y, X) (((y) + 4) * LY_STRIDE + ((X) & 127))
#define LC_AT(y, X) (((y) + 4) * LC_STRIDE + ((X) & 63))

struct __attribute__((aligned(16))) LfWaveLds {
    unsigned char tY[20 * LY_STRIDE];    // 2560
    unsigned char tU[12 * LC_STRIDE];    //  768
    unsigned char tV[12 * LC_STRIDE];    //  768 -> 4096
};
static_assert(sizeof(LfWaveLds) == 4096, "LfWaveLds layout");

// All edges of one pixel line held in registers: a[0..19] = positions -4..15 across the MB
// (x for the vertical-edge pass, y for the horizontal-edge pass; chroma lines use a[0..11]).
// Order and gating exactly as vp8_loop_filter_frame (loopfilter.c:265-299): MB edge at 0 (if there
// is a neighbour), then inner edges at 4, 8, 12 (if !skip_lf); chroma has edges 0 and 4 only and
// is not touched by the simple filter.
__device__ __forceinline__ void filter_line(int a[20], bool luma, bool simple, bool mb_edge, bool inner,
                                            const LfParams &lp)
{
    if (simple) {
        if (luma) {
            if (mb_edge) filter_edge(a + 4, 2, lp, lp.mblim);
            if (inner) { filter_edge(a + 8, 2, lp, lp.blim); filter_edge(a + 12, 2, lp, lp.blim); filter_edge(a + 16, 2, lp, lp.blim); }
        }
        return;
    }
    if (mb_edge) filter_edge(a + 4, 1, lp, lp.mblim);
    if (inner) {
        filter_edge(a + 8, 0, lp, lp.blim);
        if (luma) { filter_edge(a + 12, 0, lp, lp.blim); filter_edge(a + 16, 0, lp, lp.blim); }
    }
}

// vp8_loop_filter_frame_init (loopfilter.c:117-201): level per [segment][ref_frame][mode class]
__device__ __forceinline__ int build_level(const vp8ir_frame_hdr &h, int lane)
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
    return v;
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

typedef unsigned int u32;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef GLOBAL_AS u32x4 *g_u32x4p;
typedef GLOBAL_AS const u32x4 *g_cu32x4p;
typedef GLOBAL_AS u32x2 *g_u32x2p;
typedef GLOBAL_AS const u32x2 *g_cu32x2p;

__device__ __forceinline__ void unpack4(u32 v, int *a) { a[0] = v & 0xff; a[1] = (v >> 8) & 0xff; a[2] = (v >> 16) & 0xff; a[3] = v >> 24; }
__device__ __forceinline__ u32 pack4(const int *a) { return (u32)a[0] | ((u32)a[1] << 8) | ((u32)a[2] << 16) | ((u32)a[3] << 24); }

// Two frames per wave: lanes 0..31 filter frame A, lanes 32..63 frame B of a job pair, at the same
// MB position.  One MB keeps only 32 lanes busy (16 luma + 8 U + 8 V pixel lines) and the kernel is
// bound by VALU issue (one wave instruction = 4 SIMD cycles whatever the number of active lanes),
// so sharing the instruction stream between two independent frames halves the cost per MB.
//
// XCU = true (small launches, see vp8_recon.hip): the rows of one frame pair are spread over the waves of S workgroups
// on different CUs.  The four context rows a wave hands to the wave below (luma rows 12..15, chroma rows 4..7 of
// its macroblocks, as far as it has filtered them) travel as 8-byte granules {4 pixels, launch epoch} through a
// per-row buffer in global memory (agent-scope stores and polling loads; the tag is the progress flag), and the
// upper wave leaves the rows the lower one finishes (luma 13..15, chroma 5..7) out of its own frame writes: two CUs
// must never write the same bytes.

template <bool XCU>
__device__ __forceinline__ void lf_body(const DevJob *__restrict__ jobs, int njobs, DevGeom g, u64 *gran_base, u32 epoch,
                                        int S, int *err)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NW = blockDim.x >> 6;
    const int cols = g.mb_cols, rows = g.mb_rows;
    const int half = lane >> 5, hl = lane & 31;
    int *prog = (int *)smem;
    LfWaveLds *wl = (LfWaveLds *)(smem + 256) + wave * 2 + half;
    const int xq = blockIdx.x >> 3;
    const int group = XCU ? (xq / S) * 8 + (int)(blockIdx.x & 7) : (int)blockIdx.x;
    const int gw = XCU ? (xq % S) * NW + wave : wave;
    const int TW = XCU ? S * NW : NW;
    const int GS = cols * 32;           // granules per MB row: luma 4 rows x cols*4, then U, V 4 rows x cols*2 each

    if (threadIdx.x < 64) prog[threadIdx.x] = 0;
    __syncthreads();

    const int npairs = (njobs + 1) >> 1;
    const int mypairs = XCU ? (group < npairs ? 1 : 0) : (npairs - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total_rows = mypairs * rows;
    const int dep_wave = (wave + NW - 1) % NW;
    // mode_lf_lut (loopfilter.c:52-63) indexed by MB mode: DC,V,H,TM -> 1, B_PRED -> 0,
    // NEAREST,NEAR,NEW -> 2, ZERO -> 1, SPLIT -> 3; packed 2 bits each.
    const unsigned mode_class = (1u) | (1u << 2) | (1u << 4) | (1u << 6) | (0u << 8) | (2u << 10) | (2u << 12)
                              | (1u << 14) | (2u << 16) | (3u << 18);

    // ---- lane roles inside a half: every lane owns one pixel line of the MB in both filter passes --
    // hl 0..15 a luma line (row y in the vertical-edge pass, column x in the horizontal-edge pass),
    // hl 16..23 a U line, hl 24..31 a V line.  The first four lanes of each plane additionally fetch
    // the four context rows above the MB.
    const bool luma = hl < 16;
    const int li = luma ? hl : (hl & 7);                          // line index inside its plane
    unsigned char *tile = luma ? wl->tY : ((hl & 8) ? wl->tV : wl->tU);
    const bool is_top = li < 4;
    const int top_row = li - 4;

    for (int R = gw, k = 0; R < total_rows; R += TW, ++k) {
        const int jj = R / rows, r = R - jj * rows;
        const int pair = XCU ? group : (int)blockIdx.x + jj * (int)gridDim.x;
        const bool haveB = 2 * pair + 1 < njobs;
        const DevJob &jobA = jobs[2 * pair];
        const DevJob &jobB = jobs[haveB ? 2 * pair + 1 : 2 * pair];
        const vp8ir_frame_hdr &hA = jobA.hdr, &hB = jobB.hdr;
        const int dep_seq = (R - 1) / NW;
        const bool onA = hA.filter_level != 0, onB = haveB && hB.filter_level != 0;
        if (!onA && !onB) {                      // neither frame is filtered at all (onyxd_if.c:576)
            if (!XCU) wg_publish_lds(&prog[wave], (k + 1) << 16, lane);
            continue;
        }
        g_u64p gran_mine = (g_u64p)(gran_base + ((long)(pair * 2 + half) * rows + r) * GS);
        g_u64p gran_above = gran_mine - GS;
        const bool frame_on = half ? onB : onA;
        // lane l holds lvl[seg][ref][mode class] of each frame, l = seg<<4 | ref<<2 | class
        const int lvlA = build_level(hA, lane), lvlB = build_level(hB, lane);
        const bool simple = (half ? hB.filter_type : hA.filter_type) != 0;
        const int sharp = half ? hB.sharpness_level : hA.sharpness_level;
        const int ftype = half ? vp8ir_lf_frame_type(&hB) : vp8ir_lf_frame_type(&hA);
        const vp8ir_mbx *mbs = half ? jobB.mbx : jobA.mbx;
        uint8_t *dst = half ? jobB.dst : jobA.dst;
        g_cu32p mbrow = (g_cu32p)(mbs + (long)r * cols);              // VP8IR_MBX_WORDS dwords per MB
        g_u8p fY = (g_u8p)(dst + g.y_off + (long)r * 16 * g.y_stride);
        g_u8p fU = (g_u8p)(dst + g.u_off + (long)r * 8 * g.uv_stride);
        g_u8p fV = (g_u8p)(dst + g.v_off + (long)r * 8 * g.uv_stride);
        const int pstride = luma ? g.y_stride : g.uv_stride;
        g_u8p plane = luma ? fY : ((hl & 8) ? fV : fU);
        g_u8p frow = plane + (long)li * pstride;       // this lane's pixel row in the frame
        g_u8p trow = plane + (long)top_row * pstride;  // its context row (is_top lanes)

        // One MB.  `body` = this lane's 16 (8) unfiltered pixels, prefetched; `w0`,`w1` = the first two
        // dwords of the MB descriptor of this half's frame.
        auto process = [&](const int c, const u32x4 body, const u32 w0, const u32 w1, const int level) {
            const int y_mode = w0 & 0xff;
            const bool skip_lf = y_mode != VP8IR_B_PRED && y_mode != VP8IR_SPLITMV && ((w0 >> 24) & VP8IR_MB_SKIP);
            const LfParams lp = lf_params(sharp, level, ftype);
            const bool on = frame_on && level != 0;
            const int X = luma ? c * 16 : c * 8;      // absolute x of this MB in the lane's plane

            // ---- context rows above: written by the wave of row r-1 (final-so-far values); the loads are
            // issued now and consumed after the vertical-edge pass
            u32x4 topv = { 0, 0, 0, 0 };
            g_u64p gq = nullptr;
            u64 g0 = 0, g1 = 0, g2 = 0, g3 = 0;
            if (r > 0) {
                if (XCU) {
                    // granules of MB c of the row above exist once that row is done with MB c; requested now,
                    // checked (and polled, if they were not there yet) after the vertical-edge pass
                    if (is_top) {
                        gq = gran_above + (luma ? li * cols * 4 + c * 4 : ((hl & 8) ? cols * 24 : cols * 16) + li * cols * 2 + c * 2);
                        g0 = gran_load(gq); g1 = gran_load(gq + 1);
                        if (luma) { g2 = gran_load(gq + 2); g3 = gran_load(gq + 3); }
                    }
                } else {
                    wg_wait_ge(&prog[dep_wave], (dep_seq << 16) + c + 1);
                    if (is_top) {
                        if (luma) topv = *(g_cu32x4p)(trow + c * 16);
                        else { const u32x2 t = *(g_cu32x2p)(trow + c * 8); topv.x = t.x; topv.y = t.y; }
                    }
                }
            }

            // ---- vertical edges: the whole pixel row in registers
            int a[20];
            {
                const u32 left = luma ? *(const u32 *)(tile + LY_AT(li, X - 4)) : *(const u32 *)(tile + LC_AT(li, X - 4));
                unpack4(left, a); unpack4(body.x, a + 4); unpack4(body.y, a + 8); unpack4(body.z, a + 12); unpack4(body.w, a + 16);
                if (on) filter_line(a, luma, simple, c > 0, !skip_lf, lp);
                if (luma) {
                    *(u32 *)(tile + LY_AT(li, X - 4)) = pack4(a);
                    *(u32x4 *)(tile + LY_AT(li, X)) = (u32x4){ pack4(a + 4), pack4(a + 8), pack4(a + 12), pack4(a + 16) };
                } else {
                    *(u32 *)(tile + LC_AT(li, X - 4)) = pack4(a);
                    *(u32x2 *)(tile + LC_AT(li, X)) = (u32x2){ pack4(a + 4), pack4(a + 8) };
                }
            }
            if (XCU && r > 0 && is_top) {
                topv.x = gran_wait(gq, g0, epoch, err, 2); topv.y = gran_wait(gq + 1, g1, epoch, err, 2);
                if (luma) { topv.z = gran_wait(gq + 2, g2, epoch, err, 2); topv.w = gran_wait(gq + 3, g3, epoch, err, 2); }
            }
            if (r > 0 && is_top) {
                if (luma) *(u32x4 *)(tile + LY_AT(top_row, X)) = topv;
                else *(u32x2 *)(tile + LC_AT(top_row, X)) = (u32x2){ topv.x, topv.y };
            }
            wave_lds_sync();

            // ---- horizontal edges: the whole pixel column in registers
            if (on) {
                const int stride = luma ? LY_STRIDE : LC_STRIDE;
                unsigned char *colp = tile + (luma ? LY_AT(-4, X + li) : LC_AT(-4, X + li));
                const int n = luma ? 20 : 12;
#pragma unroll
                for (int i = 0; i < 20; i++) a[i] = i < n ? colp[i * stride] : 0;
                filter_line(a, luma, simple, r > 0, !skip_lf, lp);
#pragma unroll
                for (int i = 1; i < 19; i++) if (i < n - 1) colp[i * stride] = (unsigned char)a[i];
            }
            wave_lds_sync();
        };

        // Write MBs [m0, m1) of the ring to the frame: luma rows -3..15 (0..15 on the first MB row), chroma
        // rows -3..7; 16-byte (luma) / 8-byte (chroma) pieces, one per lane and pass.  Rows 12..15 are
        // provisional (the wave below finishes them) but must be visible to it.
        auto flush = [&](const int m0, const int m1) {
            if (!frame_on) return;
            const int n = m1 - m0, y0 = r > 0 ? -3 : 0;
            // XCU: rows 13..15 (chroma 5..7) belong to the wave below, which writes their final values
            const bool keep_bottom = XCU && r < rows - 1;
            const int ny = (keep_bottom ? 13 : 16) - y0, nc = (keep_bottom ? 5 : 8) - y0;
            for (int t = hl; t < ny * n; t += 32) {
                const int y = y0 + t / n, m = m0 + t % n;
                *(g_u32x4p)(fY + (long)y * g.y_stride + m * 16) = *(const u32x4 *)(wl->tY + LY_AT(y, m * 16));
            }
            for (int t = hl; t < 2 * nc * n; t += 32) {
                const int pl = t >= nc * n, tt = pl ? t - nc * n : t;
                const int y = y0 + tt / n, m = m0 + tt % n;
                *(g_u32x2p)((pl ? fV : fU) + (long)y * g.uv_stride + m * 8) = *(const u32x2 *)((pl ? wl->tV : wl->tU) + LC_AT(y, m * 8));
            }
        };

        auto load_body = [&](int c) -> u32x4 {
            u32x4 v = { 0, 0, 0, 0 };
            if (luma) v = *(g_cu32x4p)(frow + c * 16);
            else { const u32x2 t = *(g_cu32x2p)(frow + c * 8); v.x = t.x; v.y = t.y; }
            return v;
        };
        auto load_desc = [&](int c) -> u32 { return hl < 2 ? mbrow[c * VP8IR_MBX_WORDS + hl] : 0u; };
        // descriptor fields and filter level of this half's MB (lanes 0,1 / 32,33 hold the dwords)
        auto mb_fields = [&](const u32 d, u32 &w0, u32 &w1, int &level) {
            const u32 a0 = (u32)__builtin_amdgcn_readlane((int)d, 0), a1 = (u32)__builtin_amdgcn_readlane((int)d, 1);
            const u32 b0 = (u32)__builtin_amdgcn_readlane((int)d, 32), b1 = (u32)__builtin_amdgcn_readlane((int)d, 33);
            const int la = __builtin_amdgcn_readlane(lvlA, ((a1 & 3) << 4) | (((a0 >> 16) & 3) << 2) | ((mode_class >> (2 * (a0 & 0xff))) & 3));
            const int lb = __builtin_amdgcn_readlane(lvlB, ((b1 & 3) << 4) | (((b0 >> 16) & 3) << 2) | ((mode_class >> (2 * (b0 & 0xff))) & 3));
            w0 = half ? b0 : a0; w1 = half ? b1 : a1; level = half ? lb : la;
        };

        // After MB c: once MB c's vertical-edge pass has run, MBs < c are final as far as this wave is
        // concerned.  Frame writes happen in groups that end on 64-byte sector boundaries of the frame
        // rows (x = 32 mod 64, i.e. MB index = 2 mod 4).  Publishing is one group late: by the time the
        // next group is written the previous group's stores have long been acknowledged, so the
        // vmcnt(0) in wg_publish_global costs (almost) nothing.
        // XCU: hand MB m's bottom context rows (as filtered so far: final as far as this wave is concerned) to the wave
        // below, one granule per lane: 16 luma (rows 12..15 x 4) + 8 U + 8 V (rows 4..7 x 2)
        auto emit = [&](const int m) {
            if (hl < 16) {
                const int y = 12 + (hl >> 2), i = hl & 3;
                gran_store(gran_mine + (hl >> 2) * cols * 4 + m * 4 + i, *(const u32 *)(wl->tY + LY_AT(y, m * 16 + i * 4)), epoch);
            } else {
                const int pl = (hl >> 3) & 1, yy = (hl >> 1) & 3, i = hl & 1;
                gran_store(gran_mine + (pl ? cols * 24 : cols * 16) + yy * cols * 2 + m * 2 + i,
                           *(const u32 *)((pl ? wl->tV : wl->tU) + LC_AT(4 + yy, m * 8 + i * 4)), epoch);
            }
        };
        int flushed = 0;
        auto after_mb = [&](const int c) {
            const bool last = c == cols - 1;
            if (XCU && r < rows - 1) {
                if (c > 0) emit(c - 1);
                if (last) emit(c);
            }
            if (last || (c & 3) == 2) {
                if (!XCU) wg_publish_global(&prog[wave], (k << 16) + flushed, lane);
                flush(flushed, last ? cols : c);
                flushed = last ? cols : c;
            }
        };

        // ---- software pipeline, unrolled by two so that no loaded register is ever copied
        u32x4 bodyA = load_body(0), bodyB = { 0, 0, 0, 0 };
        u32 dA = load_desc(0), dB = cols > 1 ? load_desc(1) : 0u;
        for (int c = 0; c < cols; c += 2) {
            {
                u32 w0, w1; int level;
                mb_fields(dA, w0, w1, level);
                if (c + 1 < cols) bodyB = load_body(c + 1);
                if (c + 2 < cols) dA = load_desc(c + 2);
                process(c, bodyA, w0, w1, level);
                after_mb(c);
            }
            if (c + 1 < cols) {
                u32 w0, w1; int level;
                mb_fields(dB, w0, w1, level);
                if (c + 2 < cols) bodyA = load_body(c + 2);
                if (c + 3 < cols) dB = load_desc(c + 3);
                process(c + 1, bodyB, w0, w1, level);
                after_mb(c + 1);
            }
        }
        if (!XCU) wg_publish_global(&prog[wave], (k + 1) << 16, lane);
    }
}

extern "C" __global__ void __launch_bounds__(1024)
vp8_loopfilter_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g)
{
    lf_body<false>(jobs, njobs, g, nullptr, 0u, 1, nullptr);
}

// grid = 8 * S * ceil(npairs / 8) workgroups of four or eight waves; gran: npairs * 2 * rows * cols * 32 granules
extern "C" __global__ void __launch_bounds__(512)
vp8_loopfilter_xcu_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, unsigned long long *gran, unsigned int epoch,
                          int S, int *err)
{
    lf_body<true>(jobs, njobs, g, gran, epoch, S, err);
}

// ---- border extension ------------------------------------------------------------------------
// vp8_yv12_extend_frame_borders (yv12extend.c:24-145): replicate the first/last pixel of every row
// 32 (luma) / 16 (chroma) times, then the first/last (already widened) row 32 / 16 times.
// grid = (chunks, njobs).  Only border dwords are enumerated: per plane first the 2 * border rows above and below the
// image (whole widened rows: contiguous stores), then the left and right pieces of the h image rows; every value comes
// straight from the clamped source coordinate, so there is no ordering between the two parts.
extern "C" __global__ void __launch_bounds__(256)
vp8_extend_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g)
{
    const DevJob &job = jobs[blockIdx.y];
    for (int plane = 0; plane < 3; ++plane) {
        const int w = plane ? g.aligned_w / 2 : g.aligned_w, h = plane ? g.aligned_h / 2 : g.aligned_h;
        const int stride = plane ? g.uv_stride : g.y_stride, border = plane ? 16 : 32;
        g_u8p p = (g_u8p)(job.dst + (plane == 0 ? g.y_off : plane == 1 ? g.u_off : g.v_off));
        const int row_dw = (w + 2 * border) / 4, side_dw = border / 4;
        const int n_tb = 2 * border * row_dw;               // dwords above + below
        const int total = n_tb + h * 2 * side_dw;           // + left and right of the image rows
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
            int row, xd;
            if (i < n_tb) {
                const int rr = i / row_dw;
                xd = (i - rr * row_dw) * 4 - border;
                row = rr < border ? rr - border : h + (rr - border);
            } else {
                const int j = i - n_tb, rr = j / (2 * side_dw), k = j - rr * 2 * side_dw;
                row = rr;
                xd = k < side_dw ? k * 4 - border : w + (k - side_dw) * 4;
            }
            const int sy = row < 0 ? 0 : (row >= h ? h - 1 : row);
            unsigned int v;
            if (xd >= 0 && xd < w) v = *(g_cu32p)(p + (long)sy * stride + xd);
            else v = p[(long)sy * stride + (xd < 0 ? 0 : w - 1)] * 0x01010101u;
            *(g_u32p)(p + (long)row * stride + xd) = v;
        }
    }
}
