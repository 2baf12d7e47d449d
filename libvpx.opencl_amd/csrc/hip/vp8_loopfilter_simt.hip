// VP8 in-loop deblocking filter, "one macroblock row per LANE" formulation for gfx950.
//
// Same job as vp8_loopfilter.hip: vp8_loop_filter_frame (vp8/common/loopfilter.c:203-316) with the
// filters of vp8/common/loopfilter_filters.c (normal: vp8_loop_filter_c / vp8_mbloop_filter_c behind
// vp8_loop_filter_{mbv,bv,mbh,bh}_c; simple: vp8_loop_filter_simple_*), level / limit derivation of
// loopfilter.c:24-201.  Organised like vp8_recon_simt.hip, for the same reason (the path is bound by
// VALU issue, not by memory):
//
//   * lane p of a wave owns macroblock rows p, p+G, p+2G, ... of a strand of frames and filters one
//     whole macroblock per step, two macroblocks behind lane p-1; the raster order of the reference
//     (MB (r-1,c+1) has touched the three pixel columns left of it before MB (r,c) filters its top edge)
//     holds by construction;
//   * the filter arithmetic runs on TWO pixel lines per instruction as packed 16-bit lanes (v_pk_*):
//     rows (y, y+1) for the vertical edges, columns (x, x+1) for the horizontal ones; the signed-char
//     saturations of the reference become packed min/max;
//   * the macroblock (plus the four columns left of it and the four rows above it) sits in a per-lane,
//     lane-interleaved LDS tile, so both passes are short rolled loops over conflict-free ds_read_b32;
//   * pixels another macroblock will still modify are not written early: the four right-hand columns
//     wait in registers for the next macroblock's left edge, the four bottom rows travel to the lane
//     below by DPP wave shift and are written by it.  The first lane of a strand reads them back from
//     the frame (L2-coherent loads); the last lane of a strand and the last row of a frame write them.
//     Every frame byte is written once, as aligned 16-byte (luma) / 8-byte (chroma) row pieces.
#include "vp8_common.hip.h"
#include <stddef.h>

#include "vp8_simt_prims.hip.h"

// grid = waves (one wave per block); lgG, P, nstrands as in vp8_recon_simt_kernel.  Works in place on the
// jobs' macroblock-tiled scratch frames (DevJob::tile, see VP8_TILE_BYTES): a macroblock is three 128-byte
// lines -- luma rows 0..7, luma rows 8..15, U+V -- and every line is written exactly once, whole, by the lane
// that knows its final content:
//   line 0 of MB (r,c)    by its own lane, one step later (after MB (r,c+1) revisited its last 4 columns);
//   line 1 and the chroma line by the lane below (which filters their last three rows), or by the own lane
//   when nobody is below (last row of the frame) or the lane below reads them back from memory (the first
//   lane of a strand follows the last one).
//
// PLANES selects what a wave filters: LF_BOTH (one wave does a macroblock's luma, then U, then V), or LF_LUMA / LF_CHROMA
// for the split launch, where a luma kernel and a chroma kernel run side by side.  The two halves share nothing but the
// macroblock descriptors -- separate lines of the tiled scratch frame, separate planes of the frame buffer -- and a luma
// wave and a chroma wave together fit one SIMD (registers and LDS), which a pair of whole-macroblock waves does not: the
// SIMD then has two instruction streams to issue from instead of one that stalls on every LDS and memory round trip.
enum { LF_BOTH = 0, LF_LUMA = 1, LF_CHROMA = 2 };
template <int PLANES>
__device__ __forceinline__ void lf_simt_body(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, int raster)
{
    constexpr bool DO_Y = PLANES != LF_CHROMA, DO_C = PLANES != LF_LUMA;
    __shared__ u32 tile[(DO_Y ? 100 : 36) * 64];   // luma: 20 rows x 5 dwords; chroma (reuses it): 12 rows x 3 dwords
    const int lane = threadIdx.x;
    const int G = 1 << lgG;
    const int pos = lane & (G - 1);
    const int spw = 64 >> lgG;
    const int strand = blockIdx.x * spw + (lane >> lgG);
    const int cols = g.mb_cols, rows = g.mb_rows;
    const int myjobs = strand < njobs ? (njobs - strand + nstrands - 1) / nstrands : 0;
    const int Vmax = myjobs * rows;
    const int wavejobs = (njobs - (int)blockIdx.x * spw + nstrands - 1) / nstrands;
    const int T = ((wavejobs * rows + G - 1) >> lgG) * P + 2 * (G - 1);
    const long rowbytes = (long)cols * VP8_TILE_BYTES;
    u32 *const TL = tile + lane;
    v2u one = mku(1);
    asm volatile("" : "+v"(one));            // see nz_clear

    // ---- per-lane row state
    g_cu32p mbp = nullptr;
    u32 nx_w0 = 0, nx_w1 = 0;                   // descriptor words 0, 1 of the macroblock after the current one, fetched a step ahead
    g_u8p trow = nullptr;                        // tile (r, 0)
    g_u8p rasY = nullptr, rasU = nullptr, rasV = nullptr;   // raster == 1: pixel (0,0) of MB row r in the frame buffer
    const DevJob *job = jobs;
    int r = 0;
    bool lf_on = false, simple = false;
    // the previous macroblock of the row: its first 12 (chroma: 4) pixel columns, final, and its last 4,
    // which the current macroblock's left edge may still change
    u32 pbY[16][3], sY[16], pbU[8], sU[8], pbV[8], sV[8];
    // what the lane below asks for: luma rows 8..15 and the chroma rows of the macroblock finished two steps ago
    u32 hY[8][4], hU[8][2], hV[8][2];
    // raster output: rows of an even macroblock wait one step for their right-hand neighbour, so that a frame row is
    // written in 32-byte pieces (two 16-byte stores back to back) instead of 16-byte ones a step apart
    u32x4 holdA[8], holdB[8];
    u32x2 holdU[8], holdV[8];
    if constexpr (DO_Y) {
#pragma unroll
        for (int y = 0; y < 16; y++) { pbY[y][0] = pbY[y][1] = pbY[y][2] = sY[y] = 0; }
#pragma unroll
        for (int y = 0; y < 8; y++) hY[y][0] = hY[y][1] = hY[y][2] = hY[y][3] = 0;
    }
    if constexpr (DO_C) {
#pragma unroll
        for (int y = 0; y < 8; y++) { pbU[y] = sU[y] = pbV[y] = sV[y] = 0; hU[y][0] = hU[y][1] = hV[y][0] = hV[y][1] = 0; }
    }

    int c = -2 * pos, V = pos;
    STAMP_DECL
#pragma unroll 1
    for (int t = 0; t < T; ++t, ++c) {
        STAMP(0)
        if (c == P) { c = 0; V += G; }
        // rows 8..15 (luma) / all rows (chroma) of the macroblock above, from the lane above
        u32 tY[8][4], tU[8][2], tV[8][2];
#pragma unroll
        for (int y = 0; y < 8; y++) {
            if constexpr (DO_Y) {
#pragma unroll
                for (int i = 0; i < 4; i++) tY[y][i] = from_lane_above(hY[y][i]);
            }
            if constexpr (DO_C) {
                tU[y][0] = from_lane_above(hU[y][0]); tU[y][1] = from_lane_above(hU[y][1]);
                tV[y][0] = from_lane_above(hV[y][0]); tV[y][1] = from_lane_above(hV[y][1]);
            }
        }

        // What the lane below will fetch at the start of the next step: the macroblock held from the previous
        // step.  Its last four columns are still provisional if this step filters a left edge against them
        // (they are fixed up below, after the vertical-edge pass); otherwise -- end of a row, idle step --
        // they are final as they stand.
#pragma unroll
        for (int y = 0; y < 8; y++) {
            if constexpr (DO_Y) { hY[y][0] = pbY[8 + y][0]; hY[y][1] = pbY[8 + y][1]; hY[y][2] = pbY[8 + y][2]; hY[y][3] = sY[8 + y]; }
            if constexpr (DO_C) { hU[y][0] = pbU[y]; hU[y][1] = sU[y]; hV[y][0] = pbV[y]; hV[y][1] = sV[y]; }
        }

        STAMP(1)
        const bool act = c >= 0 && c < cols && V < Vmax;
        if (act) {
            if (c == 0) {
                const int j = V / rows;
                r = V - j * rows;
                job = jobs + (strand + j * nstrands);
                const vp8ir_frame_hdr &h = job->hdr;
                lf_on = h.filter_level != 0;
                simple = h.filter_type == 1;
                mbp = (g_cu32p)(job->mbs + (long)r * cols);
                nx_w0 = mbp[0]; nx_w1 = mbp[1];
                trow = (g_u8p)(job->tile + (long)r * rowbytes);
                rasY = (g_u8p)(job->dst + g.y_off + (long)r * 16 * g.y_stride);
                rasU = (g_u8p)(job->dst + g.u_off + (long)r * 8 * g.uv_stride);
                rasV = (g_u8p)(job->dst + g.v_off + (long)r * 8 * g.uv_stride);
            }
            // (the descriptor was fetched a step ago: the tile loads below go out at once instead of behind a memory round trip)
            const u32 w0 = nx_w0, w1 = nx_w1;
            nx_w0 = mbp[16]; nx_w1 = mbp[17];     // (past the end of a row: the next row's first macroblock, or padding -- unused)
            if (lf_on || raster) {      // raster output: an unfiltered frame is still carried from the scratch to its frame buffer
            const vp8ir_frame_hdr &h = job->hdr;
            const int y_mode = w0 & 0xff, ref_frame = (w0 >> 16) & 0xff;
            const u32 flags = w0 >> 24;
            const int level = mb_level(h, w1 & 3, ref_frame & 3, y_mode);
            const Lim L = mb_limits(h.sharpness_level, level, vp8ir_lf_frame_type(&h), one);
            const bool on = lf_on && level != 0;
            const bool skip_lf = y_mode != VP8IR_B_PRED && y_mode != VP8IR_SPLITMV && (flags & VP8IR_MB_SKIP);
            const bool mbv = on && c > 0, inner = on && !skip_lf, mbh = on && r > 0;
#ifdef VP8_LF_NOFILTER    // measurement aid: data movement only
            const bool any_normal = false, any_simple = false;
#else
            const bool any_normal = __builtin_amdgcn_ballot_w64(on && !simple) != 0;
            const bool any_simple = __builtin_amdgcn_ballot_w64(on && simple) != 0;
#endif
            auto gate = [](bool b) { return mku(b ? 0xffff : 0); };
            const Gates gvY = { gate(mbv && !simple), gate(inner && !simple), gate(mbv && simple), gate(inner && simple), any_normal, any_simple };
            const Gates ghY = { gate(mbh && !simple), gate(inner && !simple), gate(mbh && simple), gate(inner && simple), any_normal, any_simple };
            // the simple filter leaves chroma alone (loopfilter.c:283-299)
            const Gates gvC = { gvY.mb, gvY.inner, mku(0), mku(0), any_normal, false };
            const Gates ghC = { ghY.mb, ghY.inner, mku(0), mku(0), any_normal, false };
            const bool last_col = c == cols - 1;
            // lines another lane would otherwise finish are written here when nobody below takes them over
            const bool write_bottom = pos == G - 1 || r == rows - 1;
            const bool readback = r > 0 && pos == 0;

            g_u8p tp = trow + (long)c * VP8_TILE_BYTES;           // this macroblock's tile
            // Where finished lines go.  raster == 0: back into the tiled scratch frame (vp8_detile_kernel converts later).
            // raster == 1: FINAL lines straight into the raster frame buffer; lines that are only handed to the first
            // lane of the strand (write_bottom on a row that is not the frame's last) still travel through the scratch.
            const bool ras = raster != 0, ras_bottom = ras && r == rows - 1;
            const int ysY = ras ? g.y_stride : 16, ysC = ras ? g.uv_stride : 8;
            const int ybY = ras_bottom ? g.y_stride : 16, ybC = ras_bottom ? g.uv_stride : 8;
            g_u8p o_left_lo = ras ? rasY + (c - 1) * 16 : tp - VP8_TILE_BYTES;                                   // MB c-1 rows 0..7
            g_u8p o_left_hi = ras_bottom ? rasY + (c - 1) * 16 + 8 * g.y_stride : tp - VP8_TILE_BYTES + 128;      // MB c-1 rows 8..15
            g_u8p o_own_lo = ras ? rasY + c * 16 : tp, o_own_hi = ras_bottom ? rasY + c * 16 + 8 * g.y_stride : tp + 128;
            g_u8p o_above = ras ? rasY - 8 * g.y_stride + c * 16 : tp - rowbytes + 128;                           // MB (r-1, c) rows 8..15
            u32x4 inY[16], inU[4], inV[4];
            if constexpr (DO_Y) {
#pragma unroll
                for (int y = 0; y < 16; y++) inY[y] = *(g_cu32x4p)(tp + 16 * y);
            }
            if constexpr (DO_C) {
#pragma unroll
                for (int y = 0; y < 4; y++) { inU[y] = *(g_cu32x4p)(tp + 256 + 16 * y); inV[y] = *(g_cu32x4p)(tp + 320 + 16 * y); }
            }
            if (readback) {
                const unsigned char *ta = (const unsigned char *)tp - rowbytes;
#pragma unroll
                for (int y = 0; y < 8; y++) {
                    if constexpr (DO_Y) {
                        const unsigned long long a = load_l2_64(ta + 128 + 16 * y), b = load_l2_64(ta + 128 + 16 * y + 8);
                        tY[y][0] = (u32)a; tY[y][1] = (u32)(a >> 32); tY[y][2] = (u32)b; tY[y][3] = (u32)(b >> 32);
                    }
                    if constexpr (DO_C) {
                        const unsigned long long u = load_l2_64(ta + 256 + 8 * y), v = load_l2_64(ta + 320 + 8 * y);
                        tU[y][0] = (u32)u; tU[y][1] = (u32)(u >> 32); tV[y][0] = (u32)v; tV[y][1] = (u32)(v >> 32);
                    }
                }
            }

            STAMP(2)
            // =============================== luma ===============================
            if constexpr (DO_Y) {
#pragma unroll
            for (int y = 0; y < 16; y++) {
                u32 *row = TL + (4 + y) * 5 * 64;
                row[0] = sY[y] ^ VP8_LF_BIAS; row[64] = inY[y].x ^ VP8_LF_BIAS; row[128] = inY[y].y ^ VP8_LF_BIAS; row[192] = inY[y].z ^ VP8_LF_BIAS; row[256] = inY[y].w ^ VP8_LF_BIAS;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                u32 *row = TL + j * 5 * 64;
#pragma unroll
                for (int i = 0; i < 4; i++) row[(1 + i) * 64] = tY[4 + j][i] ^ VP8_LF_BIAS;
            }
            STAMP(3)
            filter_plane<4, 16>(TL, gvY, ghY, L);
            STAMP(4)
            // ---- the macroblock to the left is final now: its 12 held columns + the 4 just revisited
            if (c > 0) {
#pragma unroll
                for (int y = 0; y < 16; y++) {
                    const u32 s = TL[(4 + y) * 5 * 64] ^ VP8_LF_BIAS;
                    if (y < 8) {
                        const u32x4 v = { pbY[y][0], pbY[y][1], pbY[y][2], s };
                        if (!ras) *(g_u32x4p)(o_left_lo + ysY * y) = v;
                        else if ((c - 1) & 1) { *(g_u32x4p)(o_left_lo + ysY * y - 16) = holdA[y]; *(g_u32x4p)(o_left_lo + ysY * y) = v; }
                        else holdA[y] = v;
                    }
                    else if (write_bottom) *(g_u32x4p)(o_left_hi + ybY * (y - 8)) = (u32x4){ pbY[y][0], pbY[y][1], pbY[y][2], s };
                    if (y >= 8) hY[y - 8][3] = s;
                }
            }
            // ---- line 1 of the macroblock above: rows 8..12 as received, rows 13..15 filtered
            if (r > 0) {
                const bool pair_hold = ras && !(c & 1) && !last_col, pair_flush = ras && (c & 1);
#pragma unroll
                for (int y = 0; y < 8; y++) {
                    u32x4 v;
                    if (y < 5) v = (u32x4){ tY[y][0], tY[y][1], tY[y][2], tY[y][3] };
                    else { const u32 *row = TL + (y - 4) * 5 * 64; v = (u32x4){ row[64] ^ VP8_LF_BIAS, row[128] ^ VP8_LF_BIAS, row[192] ^ VP8_LF_BIAS, row[256] ^ VP8_LF_BIAS }; }
                    if (pair_hold) holdB[y] = v;
                    else {
                        if (pair_flush) *(g_u32x4p)(o_above + ysY * y - 16) = holdB[y];
                        *(g_u32x4p)(o_above + ysY * y) = v;
                    }
                }
            }
            // ---- this macroblock: hold it, or finish it at the end of the row
#pragma unroll
            for (int y = 0; y < 16; y++) {
                const u32 *row = TL + (4 + y) * 5 * 64;
                const u32 d0 = row[64] ^ VP8_LF_BIAS, d1 = row[128] ^ VP8_LF_BIAS, d2 = row[192] ^ VP8_LF_BIAS, d3 = row[256] ^ VP8_LF_BIAS;
                pbY[y][0] = d0; pbY[y][1] = d1; pbY[y][2] = d2; sY[y] = d3;
                if (last_col && y < 8) {
                    if (ras && (c & 1)) *(g_u32x4p)(o_own_lo + ysY * y - 16) = holdA[y];      // its even left neighbour was waiting
                    *(g_u32x4p)(o_own_lo + ysY * y) = (u32x4){ d0, d1, d2, d3 };
                }
                if (last_col && y >= 8 && write_bottom) *(g_u32x4p)(o_own_hi + ybY * (y - 8)) = (u32x4){ d0, d1, d2, d3 };
            }
            }
            STAMP(5)
            // =============================== chroma ===============================
            if constexpr (DO_C) {
#pragma unroll
            for (int pl = 0; pl < 2; pl++) {
                g_u8p tc = tp + (pl ? 320 : 256);
                g_u8p rasC = pl ? rasV : rasU;
                g_u8p oc_left = ras_bottom ? rasC + (c - 1) * 8 : tc - VP8_TILE_BYTES, oc_own = ras_bottom ? rasC + c * 8 : tc;
                g_u8p oc_above = ras ? rasC - 8 * g.uv_stride + c * 8 : tc - rowbytes;
                u32 (&pb)[8] = pl ? pbV : pbU;
                u32 (&sC)[8] = pl ? sV : sU;
                u32 (&tC)[8][2] = pl ? tV : tU;
                u32 (&hC)[8][2] = pl ? hV : hU;
                const u32x4 (&in)[4] = pl ? inV : inU;
#pragma unroll
                for (int y = 0; y < 8; y++) {
                    u32 *row = TL + (4 + y) * 3 * 64;
                    row[0] = sC[y] ^ VP8_LF_BIAS;
                    row[64] = ((y & 1) ? in[y >> 1].z : in[y >> 1].x) ^ VP8_LF_BIAS; row[128] = ((y & 1) ? in[y >> 1].w : in[y >> 1].y) ^ VP8_LF_BIAS;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    u32 *row = TL + j * 3 * 64;
                    row[64] = tC[4 + j][0] ^ VP8_LF_BIAS; row[128] = tC[4 + j][1] ^ VP8_LF_BIAS;
                }
                STAMP(6)
                filter_plane<2, 8>(TL, gvC, ghC, L);
                STAMP(7)
                if (c > 0) {
#pragma unroll
                    for (int y = 0; y < 8; y++) {
                        const u32 s = TL[(4 + y) * 3 * 64] ^ VP8_LF_BIAS;
                        if (write_bottom) *(g_u32x2p)(oc_left + ybC * y) = (u32x2){ pb[y], s };
                        hC[y][1] = s;
                    }
                }
                if (r > 0) {
                    const bool pair_hold = ras && !(c & 1) && !last_col, pair_flush = ras && (c & 1);
                    u32x2 (&hold)[8] = pl ? holdV : holdU;
#pragma unroll
                    for (int y = 0; y < 8; y++) {
                        u32x2 v;
                        if (y < 5) v = (u32x2){ tC[y][0], tC[y][1] };
                        else { const u32 *row = TL + (y - 4) * 3 * 64; v = (u32x2){ row[64] ^ VP8_LF_BIAS, row[128] ^ VP8_LF_BIAS }; }
                        if (pair_hold) hold[y] = v;
                        else {
                            if (pair_flush) *(g_u32x2p)(oc_above + ysC * y - 8) = hold[y];
                            *(g_u32x2p)(oc_above + ysC * y) = v;
                        }
                    }
                }
#pragma unroll
                for (int y = 0; y < 8; y++) {
                    const u32 *row = TL + (4 + y) * 3 * 64;
                    const u32 d0 = row[64] ^ VP8_LF_BIAS, d1 = row[128] ^ VP8_LF_BIAS;
                    pb[y] = d0; sC[y] = d1;
                    if (last_col && write_bottom) *(g_u32x2p)(oc_own + ybC * y) = (u32x2){ d0, d1 };
                }
                STAMP(8)
            }
            }
            }
            mbp += 16;
        }
    }
    STAMP_FLUSH(vp8_stamps_lf)
}

// (Round 2 also built lf_simt_body<LF_BOTH>, one wave for all three planes: 402 registers, never faster than the pair; gone.)
extern "C" __global__ void __launch_bounds__(64)
vp8_loopfilter_simt_luma_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, int raster)
{
    lf_simt_body<LF_LUMA>(jobs, njobs, g, lgG, P, nstrands, raster);
}
extern "C" __global__ void __launch_bounds__(64)
vp8_loopfilter_simt_chroma_kernel(const DevJob *__restrict__ jobs, int njobs, DevGeom g, int lgG, int P, int nstrands, int raster)
{
    lf_simt_body<LF_CHROMA>(jobs, njobs, g, lgG, P, nstrands, raster);
}
